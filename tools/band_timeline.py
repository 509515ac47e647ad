"""Per-stripe time line of the fp64 band kernel on long pairs (PHMM_BAND_TIMELINE, a tuning aid of the library): a few pairs of one
shape, each on its team of four waves -- when every stripe began, had its first non-skipped superstep behind it, ended, and how long
its wave waited for the stripe above.  usage: python tools/band_timeline.py [n_pairs=25] [R=10000] [H=12000] [out=gpurun_out/band_timeline.txt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 25
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
H = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
out = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/band_timeline.txt"
os.environ["PHMM_ENABLE_TEST_HOOKS"] = "1"
from rovaca_amd import capi, synth

b = synth.make_batch(seed=5, n_regions=max(1, n_pairs // 25), reads=5, haps=5, read_len=R, hap_len=H)
eng = capi.Engine(0)
rb = eng.upload(b)
for _ in range(2):
    rb.sweep()
ms, ms32 = rb.sweep_timed(3)
print(f"{b.n_pairs} pairs of {R} x {H}: pass {ms:.3f} ms = fp32 {ms32:.3f} + fp64 {ms - ms32:.3f}")
os.environ["PHMM_BAND_TIMELINE"] = out
rb.sweep()
eng.synchronize()
del os.environ["PHMM_BAND_TIMELINE"]
rows = [l.split() for l in open(out) if not l.startswith("#")]
team0 = [r for r in rows if r[0] == "0"]
print("team 0: stripe begin first_done end c_first out_first out_last waited_us run")
for r in team0:
    print("  ".join(f"{x:>9s}" for x in r[1:]))
