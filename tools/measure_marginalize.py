"""where does allele marginalisation belong?  Host routine (rovaca_amd/host, what the drop-in adapter uses)
against the device form behind the C ABI (phmm_marginalize: upload, kernel, download), for a region-sized matrix and for
batch-scale ones.  Prints microseconds per call; the numbers go into DESIGN.md."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, hostapi

eng = capi.Engine(0)
rng = np.random.default_rng(1)
for n_old, n_ev, n_new in ((25, 40, 3), (64, 400, 4), (128, 5000, 6), (64, 200000, 5), (128, 1000000, 6)):
    old = -rng.random((n_old, n_ev)) * 30.0
    m = rng.integers(-1, n_new, n_old)
    reps = 2000 if n_ev <= 400 else 200 if n_ev <= 5000 else 10
    hostapi.marginal_likelihoods_direct(old, m, n_new)
    t0 = time.perf_counter()
    for _ in range(reps):
        h = hostapi.marginal_likelihoods_direct(old, m, n_new)
    host_us = (time.perf_counter() - t0) / reps * 1e6
    eng.marginalize(old, m, n_new)
    t0 = time.perf_counter()
    kern = 0.0
    for _ in range(reps):
        d, ms = eng.marginalize(old, m, n_new)
        kern += ms
    dev_us = (time.perf_counter() - t0) / reps * 1e6
    assert np.array_equal(h.view(np.uint64), d.view(np.uint64))
    byt = 8 * n_ev * (n_old + n_new)
    print(f"{n_old:4d} old alleles x {n_ev:8d} reads -> {n_new}: host {host_us:10.1f} us   device call {dev_us:10.1f} us "
          f"(kernel {kern / reps * 1e3:8.1f} us, {byt / (kern / reps * 1e-3) / 1e9 if kern else 0:7.1f} GB/s of {byt / 1e6:.2f} MB)")
eng.close()
