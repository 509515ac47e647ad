"""Latency of the one-shot C ABI call on region-sized batches (what the cross-region batcher submits) -- tuning aid.
Prints host wall time of phmm_compute (stats.total_ms) against the device time of the sweep (stats.kernel_ms)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

eng = capi.Engine(0)
for n_regions in (1, 4, 10, 30, 100):
    b = synth.make_batch(seed=3, n_regions=n_regions, reads=48, haps=9, read_len=150, hap_len=300)
    for prep in (False, True):
        kw = dict(prep=(np.full(b.n_reads, 60, np.uint8), 18, 3)) if prep else {}
        tot, ker = [], []
        for _ in range(60):
            raw, l10, st = eng.compute(b, **kw)
            tot.append(st.total_ms)
            ker.append(st.kernel_ms)
        if os.environ.get('PHMM_PHASE_TIMING'):
            sys.stderr.flush()
        tot, ker = np.array(tot[10:]), np.array(ker[10:])
        print(f"regions={n_regions:4d} pairs={b.n_pairs:6d} prep={int(prep)}: total {np.median(tot)*1e3:7.1f} us (min {tot.min()*1e3:7.1f})  "
              f"kernel {np.median(ker)*1e3:7.1f} us  -> {b.cells()/np.median(tot)/1e6:7.1f} GCUPS", flush=True)
