"""Latency of swhip_align on region-sized batches (tuning aid): host wall time of the C call and device time of the launches."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import swapi

rng = np.random.default_rng(5)
acgt = np.frombuffer(b"ACGT", np.uint8)
al = swapi.Aligner(0)
for n_regions in (1, 4, 10, 30):
    refs, seqs, pairs = [], [], []
    for region in range(n_regions):
        hap = acgt[rng.integers(0, 4, 300)]
        refs.append(hap)
        for r in range(48):
            a = int(rng.integers(0, 150))
            rd = hap[a:a + 150].copy()
            rd[rng.random(150) < 0.01] = ord("C")
            seqs.append(rd)
            pairs.append((region, len(seqs) - 1))
    tot, ker = [], []
    for _ in range(40):
        out, cig, st = al.align(refs, seqs, pairs)
        tot.append(st.total_ms)
        ker.append(st.kernel_ms)
    print(f"regions={n_regions:3d} alignments={len(pairs):5d}: call {np.median(tot[5:])*1e3:7.1f} us  kernel {np.median(ker[5:])*1e3:7.1f} us", flush=True)
