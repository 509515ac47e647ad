"""Smith-Waterman kernel time by shape (tuning aid, GPU box): n alignments of reads of a given length against haplotypes of a given
length, a few shapes, kernel time by the call's HIP events.  usage: python tools/sw_shapes.py [n=48000]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import swapi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
rng = np.random.default_rng(3)
acgt = np.frombuffer(b"ACGT", np.uint8)
al = swapi.Aligner(0)
for rl, hl in ((60, 300), (100, 300), (150, 400), (250, 600), (150, 1000)):
    n_h = max(1, n // 12)
    refs = [acgt[rng.integers(0, 4, hl)] for _ in range(n_h)]
    seqs, pairs = [], []
    for i in range(n):
        h = refs[i % n_h]
        a = int(rng.integers(0, hl - rl + 1))
        r = h[a:a + rl].copy()
        r[rng.random(rl) < 0.01] = ord("C")
        seqs.append(r)
        pairs.append((i % n_h, i))
    pb = swapi.PackedBatch(refs, seqs, pairs)
    for _ in range(2):
        al.align_packed(pb)
    ks = []
    for _ in range(6):
        _, _, st = al.align_packed(pb)
        ks.append(st.kernel_ms)
    print(f"{n} x ({rl} x {hl}): kernel {np.median(ks):.3f} ms -> {st.cells / np.median(ks) / 1e6:.0f} GCUPS ({st.n_launches} launch(es))", flush=True)
