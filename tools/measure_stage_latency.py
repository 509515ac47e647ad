"""Latency of one region batch through the pinned stage (phmm_stage_submit + phmm_stage_wait) -- what a dispatcher of the drop-in
adapter does per submission.  Tuning aid: wall time against the device time of set-up + sweep + re-run (HIP events)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

eng = capi.Engine(0)
st = capi.Stage(eng, 128, 8192, 2048, 4 << 20, 128 * 1024)
for n_regions in (1, 4, 8, 16, 30, 60, 100):
    b = synth.make_batch(seed=3, n_regions=n_regions, reads=48, haps=9, read_len=150, hap_len=300)
    mapq = np.full(b.n_reads, 60, np.uint8)
    for model in (0, 3):
        n = st.fill(b, capi.regions_of(b), mapq)
        tot, ker = [], []
        for _ in range(60):
            t0 = time.perf_counter()
            st.submit(*n, 18, model)
            s = st.wait()
            tot.append(time.perf_counter() - t0)
            ker.append(s.kernel_ms)
        tot, ker = np.array(tot[10:]) * 1e6, np.array(ker[10:]) * 1e3
        print(f"regions={n_regions:4d} pairs={b.n_pairs:6d} pcr_model={model}: wall {np.median(tot):7.1f} us (min {tot.min():7.1f})  "
              f"device {np.median(ker):7.1f} us  -> {b.cells() / np.median(tot) / 1e3:7.1f} GCUPS", flush=True)
st.close()
eng.close()
