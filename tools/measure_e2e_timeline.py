"""Host time line of the chunked one-shot call (phmm_compute on BASELINE config 2) -- PHMM_PHASE_TIMING=1 makes the library print
when each chunk was packed, enqueued and collected; this script runs a few warm calls, then prints the median wall time of ten."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

b = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "c2")
eng = capi.Engine(0)
out = (np.zeros(b.n_pairs, np.float32), np.zeros(b.n_pairs, np.float64))
os.environ.pop("PHMM_PHASE_TIMING", None)
for _ in range(3):
    eng.compute(b, out=out)
walls = []
for _ in range(10):
    t0 = time.perf_counter()
    _, _, st = eng.compute(b, out=out)
    walls.append(time.perf_counter() - t0)
print(f"median wall {np.median(walls) * 1e3:.3f} ms (min {min(walls) * 1e3:.3f}) -> {b.cells() / np.median(walls) / 1e9:.0f} GCUPS; "
      f"device span of the last call {st.kernel_ms:.3f} ms", flush=True)
os.environ["PHMM_PHASE_TIMING"] = "1"
for _ in range(2):
    sys.stderr.write("---- call ----\n")
    eng.compute(b, out=out)
