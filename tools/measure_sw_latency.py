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

# the assembler's call site (hc_assemble_cigar_cacl.c:417-428): every haplotype path of a region against the padded reference
for n_haps, L in ((16, 400), (64, 500), (128, 600)):
    ref = acgt[rng.integers(0, 4, L)]
    seqs = []
    for h in range(n_haps):
        s = ref.copy()
        for _ in range(h % 5):
            s[int(rng.integers(0, L))] = acgt[rng.integers(0, 4)]
        if h % 3 == 1:
            k = int(rng.integers(10, L - 10))
            s = np.concatenate([s[:k], s[k + 3:]])
        seqs.append(s)
    pairs = [(0, h) for h in range(n_haps)]
    tot, ker = [], []
    for _ in range(30):
        out, cig, st = al.align([ref], seqs, pairs, swapi.PARAMS_ASSEMBLE)
        tot.append(st.total_ms)
        ker.append(st.kernel_ms)
    print(f"assembly shape: {n_haps:3d} haplotypes x reference {L}: call {np.median(tot[5:])*1e3:7.1f} us  kernel {np.median(ker[5:])*1e3:7.1f} us  "
          f"({st.cells/np.median(ker[5:])/1e6:.0f} GCUPS)", flush=True)
