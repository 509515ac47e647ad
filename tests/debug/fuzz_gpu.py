"""Randomised parity fuzz on the GPU (not collected by pytest): random batches of random shapes through both C ABIs against the
CPU restatements.  usage: python tests/debug/fuzz_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402
from rovaca_amd import capi, swapi  # noqa: E402
from rovaca_amd.batch import Batch  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
acgt = np.frombuffer(b"ACGTN", np.uint8)
eng, al = capi.Engine(0), swapi.Aligner(0)
t_end = time.time() + budget
n_ph = n_sw = it = 0
while time.time() < t_end:
    it += 1
    # ---- PairHMM ----
    mode = int(rng.integers(0, 4))
    rmax = [40, 260, 700, 2600][mode]
    hmax = [80, 520, 1500, 5000][mode]
    n_reads, n_haps = int(rng.integers(1, [40, 30, 10, 4][mode])), int(rng.integers(1, [12, 10, 5, 3][mode]))
    haps = [acgt[rng.integers(0, 5 if rng.random() < 0.2 else 4, int(rng.integers(1, hmax)))] for _ in range(n_haps)]
    reads = []
    general = rng.random() < 0.3
    for _ in range(n_reads):
        R = int(rng.integers(1, rmax))
        hap = haps[int(rng.integers(0, n_haps))]
        if len(hap) >= R and rng.random() < 0.8:
            a = int(rng.integers(0, len(hap) - R + 1))
            seq = hap[a:a + R].copy()
            seq[rng.random(R) < rng.choice([0.0, 0.01, 0.1])] = ord("G")
        else:
            seq = acgt[rng.integers(0, 4, R)]
        q = rng.integers(2, 60, R).astype(np.uint8)
        if general:
            gi, gd, gc = (rng.integers(5, 50, R).astype(np.uint8) for _ in range(3))
        else:
            gi = gd = rng.choice(np.array([45, 40, 35], np.uint8), R)
            gc = np.full(R, 10, np.uint8)
        reads.append((seq, q, gi, gd, gc))
    pairs = [(int(rng.integers(0, n_reads)), int(rng.integers(0, n_haps))) for _ in range(int(rng.integers(1, 200)))]
    b = Batch.from_lists(reads, haps, pairs)
    raw, l10, st = eng.compute(b)
    oraw, ol10, oresc = oracle.batch(b, n_threads=8)
    if not np.array_equal(raw.view(np.uint32), oraw.view(np.uint32)):
        bad = np.nonzero(raw.view(np.uint32) != oraw.view(np.uint32))[0]
        print("PAIRHMM fp32 MISMATCH iter", it, "pairs", bad[:5], [(len(reads[pairs[k][0]][0]), len(haps[pairs[k][1]])) for k in bad[:5]])
        sys.exit(1)
    fin = np.isfinite(ol10)
    if not (np.array_equal(np.isfinite(l10), fin) and np.array_equal(l10[~oresc], ol10[~oresc]) and
            (not (oresc & fin).any() or np.abs(l10[oresc & fin] - ol10[oresc & fin]).max() < 1e-9)):
        print("PAIRHMM log10 MISMATCH iter", it, "general", general, "mode", mode)
        for k in range(len(pairs)):
            same = (l10[k] == ol10[k]) or (np.isnan(l10[k]) and np.isnan(ol10[k]))
            if not same:
                print("  pair", k, "R", len(reads[pairs[k][0]][0]), "H", len(haps[pairs[k][1]]), "rescued", bool(oresc[k]), "raw", raw[k], oraw[k],
                      "gpu", l10[k], "oracle", ol10[k], "delta", l10[k] - ol10[k])
        sys.exit(1)
    n_ph += len(pairs)
    # ---- Smith-Waterman ----
    n_ref = int(rng.integers(1, 6))
    lim = int(rng.choice([40, 300, 700, 2047]))
    refs = [acgt[rng.integers(0, 4, int(rng.integers(1, lim + 1)))] for _ in range(n_ref)]
    seqs, sp = [], []
    for _ in range(int(rng.integers(1, 40))):
        ref = refs[int(rng.integers(0, n_ref))]
        n = int(rng.integers(1, min(len(ref), lim) + 1))
        if rng.random() < 0.8:
            a = int(rng.integers(0, len(ref) - n + 1))
            s = ref[a:a + n].copy()
            s[rng.random(n) < rng.choice([0.0, 0.02, 0.2])] = ord("T")
            if rng.random() < 0.4 and n > 2:
                k = int(rng.integers(1, n - 1))
                s = np.concatenate([s[:k], acgt[rng.integers(0, 4, int(rng.integers(1, 12)))], s[k:]])[:2047]
        else:
            s = acgt[rng.integers(0, 4, int(rng.integers(1, lim + 1)))]
        seqs.append(s)
        sp.append((int(rng.integers(0, n_ref)), len(seqs) - 1))
    params = swapi.PARAMS_ASSEMBLE if rng.random() < 0.5 else swapi.PARAMS_GENOTYPE
    out, cig, _ = al.align(refs, seqs, sp, params)
    for p, (r, s) in enumerate(sp):
        o = oracle.sw_align(refs[r], seqs[s], params)
        got = (int(out[p]["score"]), int(out[p]["max_i"]), int(out[p]["max_j"]), int(out[p]["alignment_offset"]))
        if got != (o["score"], o["max_i"], o["max_j"], o["alignment_offset"]) or not np.array_equal(swapi.cigar_of(out[p], cig), o["cigar"]):
            print("SW MISMATCH iter", it, "pair", p, len(refs[r]), len(seqs[s]), got, o["score"], o["max_i"], o["max_j"], o["alignment_offset"],
                  oracle.cigar_string(swapi.cigar_of(out[p], cig)), oracle.cigar_string(o["cigar"]))
            sys.exit(1)
    n_sw += len(sp)
    if it % 20 == 0:
        print(f"iter {it}: {n_ph} pairhmm pairs, {n_sw} alignments ok", flush=True)
print(f"fuzz ok: {it} iterations, {n_ph} PairHMM pairs, {n_sw} Smith-Waterman alignments, seed {seed}")
