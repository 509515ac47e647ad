"""Resident fp32 sweep rate on one shape (tuning aid, GPU box): bench_sweep.py R H [regions] [reads haps].  Kernel time by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

R, H = int(sys.argv[1]), int(sys.argv[2])
reads, haps = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (40, 25)
n_regions = int(sys.argv[3]) if len(sys.argv) > 3 else max(8, int(4.5e10 / (reads * haps * R * H)))
eng = capi.Engine(0)
b = synth.make_batch(seed=7, n_regions=n_regions, reads=reads, haps=haps, read_len=R, hap_len=H)
rb = eng.upload(b)
for _ in range(4):
    rb.sweep()
ms, ms32 = rb.sweep_timed(10)
print(f"R={R} H={H} pairs={b.n_pairs} fp32 {ms32:8.3f} ms  {b.cells()/ms32/1e6:8.1f} GCUPS  (lib {os.path.basename(os.environ.get('ROVACA_AMD_HIP_LIB', 'default'))}, PHMM_TAB={os.environ.get('PHMM_TAB')})", flush=True)
rb.release()
