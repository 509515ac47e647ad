"""fp64 re-run rate on MIXED read and haplotype lengths (config 3 shape) with noisy reads -- tuning aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

eng = capi.Engine(0)
for sub in (0.08, 0.15):
    b = synth.make_batch(seed=9, n_regions=300, reads=40, haps=25, read_len=(50, 250), hap_len=(100, 500), sub_rate=sub, qual=30)
    rb = eng.upload(b)
    for _ in range(3):
        rb.sweep()
    ms, ms32 = rb.sweep_timed(5)
    raw, l10, st = rb.download()
    n = int(st.n_rescued)
    import numpy as np
    resc = raw < np.float32(1e-28)
    rl = b.read_len[b.pair_read][resc].astype(np.int64)
    hl = b.hap_len[b.pair_hap][resc].astype(np.int64)
    rcells = int((rl * hl).sum())
    print(f"sub_rate={sub}: pairs={b.n_pairs} rescued={n} ({100.0*n/b.n_pairs:.1f}%)  pass {ms:.3f} ms (fp32 {ms32:.3f}, fp64 {ms-ms32:.3f})  "
          f"fp64 rate {rcells/max(ms-ms32,1e-9)/1e6:.0f} GCUPS", flush=True)
    rb.release()
