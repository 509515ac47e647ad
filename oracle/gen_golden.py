"""Generate tests/golden/pairhmm_golden.npz from the REFERENCE's own kernels (oracle/_ref).

Run in the build container only (needs /root/reference to have been compiled by ``make -C oracle``):

    python -m oracle.gen_golden

The fixture is data only: inputs (flat batch arrays) and the values the reference kernels returned for them.
  raw_avx512_s   compute_fp_avx512_s            (core/avx512_s.cpp:259)    per pair, raw fp32 sum (x 2^120)
  raw_16x1       compute_full_prob_avx512_float_1xn (rovaca/avx_512_float.cc:29), region cases only
  raw_avx2_s     compute_fp_avx2_s              (core/avx2_s.cpp:234)
  log10_ref      (double)(log10f(raw_avx512_s) - 36.1236000061f)           (pairhmm_internal.cpp:583), NaN where
                 raw < 1e-28f (the reference's fp64 kernels are defective, SURVEY.md 8c -- no reference value exists)
  ph2pr_f32, mm_f32, ph2pr_f64, mm_f64   the reference's Context<float>/Context<double> tables (core/context.h)
  d_avx512_d / d_avx2_d   what the reference's fp64 kernels return on a few pairs (documents the defect)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from rovaca_amd import synth  # noqa: E402
from rovaca_amd.batch import Batch  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pairhmm_golden.npz")


def hand_cases(rng):
    """Edge cases SURVEY.md 8(c) lists: R not a multiple of 8/16/64, R < 16, N in read and hap, quals at
    6/18/40/MQ-capped, PCR-capped gap-opens, exact-match and 20 %-error reads, per-base varying i/d/c."""
    reads, haps, pairs = [], [], []
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def rnd(n):
        return acgt[rng.integers(0, 4, n)]

    def add(read, hap):
        reads.append(read)
        haps.append(hap)
        pairs.append((len(reads) - 1, len(haps) - 1))

    for R, H in [(1, 1), (1, 7), (2, 3), (3, 23), (7, 7), (15, 40), (16, 16), (17, 33), (31, 64), (32, 65), (33, 31),
                 (63, 100), (64, 64), (65, 129), (100, 150), (127, 128), (128, 300), (129, 257), (150, 300), (151, 299),
                 (250, 500), (40, 20), (10, 200)]:
        hap = rnd(H)
        st = rng.integers(0, max(1, H - R + 1))
        seq = np.resize(hap[st:st + R], R).copy() if H >= 1 else rnd(R)
        if R > H:
            seq = rnd(R)
        q = rng.choice(np.array([6, 18, 25, 37, 40, 60], np.uint8), R)
        i = rng.choice(np.array([45, 40, 39, 33], np.uint8), R)
        add((seq, q, i, i.copy(), np.full(R, 10, np.uint8)), hap)
    # N in read and hap, and a non-ACGTN symbol (never matches, pairhmm_internal.cpp:34-41)
    hap = rnd(80)
    hap[[5, 40, 41]] = ord("N")
    seq = hap[10:70].copy()
    seq[[0, 30]] = ord("N")
    seq[12] = ord("A") if seq[12] != ord("A") else ord("C")
    add((seq, np.full(60, 30, np.uint8), np.full(60, 45, np.uint8), np.full(60, 45, np.uint8), np.full(60, 10, np.uint8)), hap)
    # distinct i / d / c per base, quals with the high bit set (masked & 127, avx512_s.cpp:90-93)
    hap = rnd(120)
    seq = hap[20:90].copy()
    add((seq, (rng.integers(6, 60, 70) | 128).astype(np.uint8), rng.integers(10, 46, 70).astype(np.uint8),
         rng.integers(10, 46, 70).astype(np.uint8), rng.integers(5, 30, 70).astype(np.uint8)), hap)
    # 20 % error read (lands near / below 1e-28 with low quals)
    for qv in (6, 10, 20, 30):
        hap = rnd(300)
        seq = hap[50:200].copy()
        e = rng.random(150) < 0.2
        seq[e] = acgt[rng.integers(0, 4, int(e.sum()))]
        add((seq, np.full(150, qv, np.uint8), np.full(150, 40, np.uint8), np.full(150, 40, np.uint8), np.full(150, 10, np.uint8)), hap)
    # unrelated read / hap with high quals: deep underflow -> rescue
    for R, H in [(100, 150), (150, 300), (60, 90)]:
        add((rnd(R), np.full(R, 40, np.uint8), np.full(R, 45, np.uint8), np.full(R, 45, np.uint8), np.full(R, 10, np.uint8)), rnd(H))
    return Batch.from_lists(reads, haps, pairs)


def main():
    R = oracle.ref()
    if R is None:
        raise SystemExit("oracle/_ref/libphmm_ref.so missing (run `make -C oracle` where /root/reference exists)")
    rng = np.random.default_rng(20251121)
    out = {}

    def run_pairs(tag, b, regions=False):
        raw512, _ = oracle.ref_pairs_avx512_s(b)
        raw2 = np.array([R.ref_pair_avx2_s(*b.read(int(r)), int(b.read_len[r]), b.hap(int(h)), int(b.hap_len[h]))
                         for r, h in zip(b.pair_read, b.pair_hap)], np.float32)
        for k in ("read_bases", "read_qual", "read_ins", "read_del", "read_gcp", "read_off", "read_len", "hap_bases",
                  "hap_off", "hap_len", "pair_read", "pair_hap"):
            out[f"{tag}_{k}"] = getattr(b, k)
        out[f"{tag}_raw_avx512_s"] = raw512
        out[f"{tag}_raw_avx2_s"] = raw2
        with np.errstate(divide="ignore", invalid="ignore"):
            l10 = (np.log10(raw512.astype(np.float32)).astype(np.float32) - np.float32(36.1236000061)).astype(np.float64)
        # log10f via numpy float32 log10 is not guaranteed to be glibc's log10f: take it from libm through the oracle lib
        L = oracle.lib()
        l10 = np.array([L.phmm_oracle_finalize_f32(float(x)) if x >= np.float32(1e-28) else np.nan for x in raw512])
        out[f"{tag}_log10_ref"] = l10
        if regions:
            raw16, _ = oracle.ref_regions_16x1(b)
            out[f"{tag}_raw_16x1"] = raw16
            for k in ("region_read", "region_hap", "region_pair"):
                out[f"{tag}_{k}"] = getattr(b, k)
        print(f"{tag}: {b.n_pairs} pairs, rescued-range {int((raw512 < np.float32(1e-28)).sum())}")
        return raw512

    hb = hand_cases(rng)
    run_pairs("edge", hb)
    # a few values of the reference's (defective) fp64 kernels, to document SURVEY.md 8c finding 1
    n = min(12, hb.n_pairs)
    out["edge_d_avx512_d"] = np.array([R.ref_pair_avx512_d(*hb.read(int(hb.pair_read[p])), int(hb.read_len[hb.pair_read[p]]),
                                                             hb.hap(int(hb.pair_hap[p])), int(hb.hap_len[hb.pair_hap[p]]))
                                       for p in range(n)])
    out["edge_d_avx2_d"] = np.array([R.ref_pair_avx2_d(*hb.read(int(hb.pair_read[p])), int(hb.read_len[hb.pair_read[p]]),
                                                         hb.hap(int(hb.pair_hap[p])), int(hb.hap_len[hb.pair_hap[p]]))
                                     for p in range(n)])
    # region-shaped samples of the BASELINE shapes (small: a few hundred pairs each)
    run_pairs("c1", synth.make_batch(seed=101, n_regions=1, reads=20, haps=8, read_len=100, hap_len=150), regions=True)
    run_pairs("c2", synth.make_batch(seed=102, n_regions=1, reads=20, haps=8, read_len=150, hap_len=300), regions=True)
    run_pairs("c3", synth.make_batch(seed=103, n_regions=6, reads=19, haps=5, read_len=(50, 250), hap_len=(100, 500),
                                     hap_n_rate=0.002), regions=True)
    # noisy uniform-quality reads: a good share lands below 1e-28 (rescue range)
    run_pairs("noisy", synth.make_batch(seed=104, n_regions=2, reads=16, haps=4, read_len=(100, 250), hap_len=(260, 500),
                                        sub_rate=0.15, qual=30))
    t = oracle.ref_tables()
    out.update({k: v for k, v in t.items()})
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
