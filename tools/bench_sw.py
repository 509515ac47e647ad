"""Throughput of the MI355X Smith-Waterman (include/smithwaterman_hip.h) on region-shaped work -- every read of a region
against its best haplotype, the genotyping call site (alignment_utils.cpp:572-579) -- with the reference's own sw_avx.c on
one host thread beside it (oracle/_ref/libsw_ref.so).  GCUPS = sum(len1 * len2) / time / 1e9.  Parity of the whole batch is
checked against the CPU restatement on a sample."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import swapi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--regions", type=int, default=1000)
ap.add_argument("--reads", type=int, default=48)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--hap-len", type=int, default=400)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--no-cpu", action="store_true")
ap.add_argument("--no-check", action="store_true", help="tuning builds that break parity on purpose")
args = ap.parse_args()

rng = np.random.default_rng(5)
acgt = np.frombuffer(b"ACGT", np.uint8)
refs, seqs, pairs = [], [], []
for region in range(args.regions):
    base = acgt[rng.integers(0, 4, args.hap_len)]
    haps = []
    for h in range(4):
        hap = base.copy()
        for _ in range(h):
            hap[int(rng.integers(0, len(hap)))] = acgt[rng.integers(0, 4)]
        haps.append(hap)
    h0 = len(refs)
    refs += haps
    for r in range(args.reads):
        hap = haps[r % 4]
        a = int(rng.integers(0, len(hap) - args.read_len + 1))
        read = hap[a:a + args.read_len].copy()
        read[rng.random(len(read)) < 0.01] = ord("C")
        if r % 6 == 0:
            k = int(rng.integers(1, len(read) - 1))
            read = np.concatenate([read[:k], acgt[rng.integers(0, 4, 3)], read[k:]])
        seqs.append(read)
        pairs.append((h0 + r % 4, len(seqs) - 1))

al = swapi.Aligner(0)
out, cig, st = al.align(refs, seqs, pairs)  # warm-up: buffers
best_call, best_kernel = 1e30, 1e30
for _ in range(args.iters):
    out, cig, st = al.align(refs, seqs, pairs)
    best_call = min(best_call, st.total_ms * 1e-3)  # the C call: binning, one upload, launches, downloads
    best_kernel = min(best_kernel, st.kernel_ms * 1e-3)
cells = int(st.cells)
res = {"alignments": len(pairs), "cells": cells, "gpu_kernel_ms": round(best_kernel * 1e3, 3),
       "gpu_kernel_gcups": round(cells / best_kernel / 1e9, 2), "gpu_call_ms": round(best_call * 1e3, 3),
       "gpu_call_gcups_pcie_inclusive": round(cells / best_call / 1e9, 2)}

import oracle  # noqa: E402  (checker + CPU baseline only)
if not args.no_check:
    for p in rng.choice(len(pairs), 200, replace=False):
        o = oracle.sw_align(refs[pairs[p][0]], seqs[pairs[p][1]])
        assert (int(out[p]["score"]), int(out[p]["alignment_offset"])) == (o["score"], o["alignment_offset"])
        assert np.array_equal(swapi.cigar_of(out[p], cig), o["cigar"])
    res["parity_sample"] = "200 alignments bit-identical to the CPU restatement"
if not args.no_cpu and oracle.sw_ref() is not None:
    rb, ro, rl = swapi._tables(refs)
    sb, so, sl = swapi._tables(seqs)
    pr = np.ascontiguousarray([p[0] for p in pairs], np.uint32)
    ps = np.ascontiguousarray([p[1] for p in pairs], np.uint32)
    n = min(len(pairs), 20000)
    chk = C.c_int64()
    sec = oracle.sw_ref().ref_sw_batch(0, rb, ro, rl, sb, so, sl, pr[:n].copy(), ps[:n].copy(), n, 1, C.byref(chk))
    ccells = int(sum(int(rl[pr[k]]) * int(sl[ps[k]]) for k in range(n)))
    res["cpu_reference_1_thread_gcups"] = round(ccells / sec / 1e9, 3)
    res["cpu_sample"] = "%d alignments through the reference's sw_avx.c (function chosen by sw_avx_init), %.2f s" % (n, sec)
print(json.dumps(res))
