"""Throughput when many pairs underflow fp32 and are re-run in fp64 (noisy uniform-quality reads) -- tuning aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

eng = capi.Engine(0)
for sub, q in ((0.01, "illumina"), (0.08, 30), (0.15, 30)):
    b = synth.make_batch(seed=9, n_regions=300, reads=40, haps=25, read_len=150, hap_len=300, sub_rate=sub, qual=q)
    rb = eng.upload(b)
    for _ in range(3):
        rb.sweep()
    ms, ms32 = rb.sweep_timed(5)
    raw, l10, st = rb.download()
    n = int(st.n_rescued)
    print(f"sub_rate={sub} qual={q}: pairs={b.n_pairs} rescued={n} ({100.0*n/b.n_pairs:.1f}%)  pass {ms:.3f} ms (fp32 {ms32:.3f} ms, "
          f"fp64 {ms-ms32:.3f} ms)  {b.cells()/ms/1e6:.0f} GCUPS overall; fp64 rate {n*45000/max(ms-ms32,1e-9)/1e6:.0f} GCUPS", flush=True)
    rb.release()
