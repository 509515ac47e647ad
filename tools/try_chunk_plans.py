"""One-shot phmm_compute (host buffers in, log10 out) under different chunk-size plans (PHMM_CHUNK_WEIGHTS) -- tuning aid, GPU box.
usage: python tools/try_chunk_plans.py [config] [plan ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
plans = sys.argv[2:] or ["1,2,3,2,1", "1,2,1", "1,3,2", "2,3,1", "1,1", "3,5,4,2", "1,2,3,2,1"]
b = synth.make_config(cfg)
eng = capi.Engine(0)
eng.compute(b.subset(range(1000)))
out = (np.zeros(b.n_pairs, np.float32), np.zeros(b.n_pairs, np.float64))
for pl in plans:
    os.environ["PHMM_ENABLE_TEST_HOOKS"] = "1"
    os.environ["PHMM_CHUNK_WEIGHTS"] = pl
    for _ in range(3):
        eng.compute(b, out=out)
    w = []
    for _ in range(12):
        t0 = time.perf_counter()
        _, l10, st = eng.compute(b, out=out)
        w.append(time.perf_counter() - t0)
    print(f"{cfg} {pl:24s} median {np.median(w)*1e3:7.3f} ms  min {min(w)*1e3:7.3f}  -> {b.cells()/np.median(w)/1e9:7.1f} GCUPS  kernel_ms {st.kernel_ms:.2f}", flush=True)
