"""fp32 team kernel on batches of mid-length reads (300-2000 bp) -- tuning aid: how much the zero-superstep shortcut saves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

eng = capi.Engine(0)
for R, H, n_regions in ((400, 600, 400), (1000, 1200, 100), (2000, 2400, 30)):
    b = synth.make_batch(seed=4, n_regions=n_regions, reads=20, haps=10, read_len=R, hap_len=H)
    rb = eng.upload(b)
    for _ in range(2):
        rb.sweep()
    ms, ms32 = rb.sweep_timed(3)
    raw, l10, st = rb.download()
    print(f"R={R} H={H} pairs={b.n_pairs}: pass {ms:.3f} ms (fp32 {ms32:.3f})  rescued {st.n_rescued}  {b.cells()/ms/1e6:.0f} GCUPS", flush=True)
    rb.release()
