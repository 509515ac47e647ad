"""Does the device run slower right after it has been idle?  ("Config 3 one-shot".)

A resident batch of `pairs` pairs of config 2's or 3's shape is swept (HIP events around the sweep, `phmm_sweep_timed`) back to back,
and then with the host sleeping `gap` ms after the device has drained, for a list of gaps.  Sizes from a one-shot call's chunk
(100 000 pairs, ~1 ms) to the whole batch.  Prints the median and the minimum of the sweeps' device time per case."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PHMM_ENABLE_TEST_HOOKS", "1")
from rovaca_amd import capi, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
full = synth.make_config(cfg)
eng = capi.Engine(0)
print(f"# {cfg}: sweep of a resident batch, device ms by HIP events; gap = host sleep after the device drained")
print("# pairs gap_ms median_ms min_ms max_ms GCUPS_at_median")
for n in (100_000, 250_000, full.n_pairs):
    b = full if n == full.n_pairs else full.subset(range(n))
    res = eng.upload(b)
    for _ in range(5):
        res.sweep_timed(1)
    for gap in (0.0, 0.2, 0.5, 1.0, 2.0, 5.0, 20.0):
        ms = []
        for _ in range(15):
            eng.synchronize()
            if gap:
                t_end = time.perf_counter() + gap * 1e-3
                while time.perf_counter() < t_end:
                    pass
            ms.append(res.sweep_timed(1)[0])
        med = statistics.median(ms)
        print(f"{n} {gap} {med:.4f} {min(ms):.4f} {max(ms):.4f} {b.cells() / med / 1e6:.0f}", flush=True)
    res.release()

# the one-shot call (host buffers in, log10 out) after the same pauses: wall time of the call
import numpy as np
out = (np.zeros(full.n_pairs, np.float32), np.zeros(full.n_pairs, np.float64))
for _ in range(3):
    eng.compute(full, out=out)
print("# one-shot phmm_compute of the whole batch: gap_ms median_wall_ms min_ms max_ms GCUPS_at_median")
for gap in (0.0, 0.5, 1.0, 2.0, 5.0, 20.0):
    ms = []
    for _ in range(9):
        if gap:
            t_end = time.perf_counter() + gap * 1e-3
            while time.perf_counter() < t_end:
                pass
        t0 = time.perf_counter()
        eng.compute(full, out=out)
        ms.append((time.perf_counter() - t0) * 1e3)
    med = statistics.median(ms)
    print(f"{gap} {med:.3f} {min(ms):.3f} {max(ms):.3f} {full.cells() / med / 1e6:.0f}", flush=True)
