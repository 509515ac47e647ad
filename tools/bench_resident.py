"""Resident fp32 sweep rate of a BASELINE config (tuning aid, GPU box): bench_resident.py c3 [scale].  Kernel time by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
eng = capi.Engine(0)
b = synth.make_config(name, scale=scale)
rb = eng.upload(b)
for _ in range(3):
    rb.sweep()
ms, ms32 = rb.sweep_timed(8)
print(f"{name} x{scale} pairs={b.n_pairs} fp32 {ms32:8.3f} ms  {b.cells()/ms32/1e6:8.1f} GCUPS  (lib {os.path.basename(os.environ.get('ROVACA_AMD_HIP_LIB', 'default'))}, ug={os.environ.get('PHMM_NO_UNIFORM_GOP')})", flush=True)
rb.release()
