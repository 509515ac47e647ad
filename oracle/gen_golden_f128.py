"""Generate tests/golden/rescue_f128.npz: pairs whose fp32 sum underflows (the fp64 "rescue" path) with their log10 likelihood
from the independent __float128 statement of the recurrence (oracle/phmm_oracle_f128.c).

    python -m oracle.gen_golden_f128          (a minute or two on 8 cores: the long pairs are tens of millions of quad-precision cells)

Why: the reference's fp64 kernels are defective, so nothing reference-made pins the rescue; this fixture does, from
arithmetic that shares no code with the double restatement or the device kernels.  The fixture is data only: inputs (a flat batch)
and values.  Transition / emission probabilities come from the reference's own Context<double> tables when oracle/_ref is there
(they equal the oracle's, tests/test_oracle_golden.py), so the quad values answer "what should the reference's fp64 path return".
  log10_f128      log10 of the likelihood in 113-bit arithmetic (no 2^1020 scale needed), rounded to double
  below_denormal  1 where likelihood * 2^1020 < 2^-1075: a double at the reference's scale is exactly 0 there (-> -inf)
  group           0 rescued pairs of the committed reference-made goldens, 1 c2 shape with 15 % errors, 2 c3 shape with 15 % errors,
                  3 c5 shape (reads 2 000-6 000 x haplotypes <= 9 000), 4 unrelated reads (some underflow even fp64)
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from rovaca_amd import synth  # noqa: E402
from rovaca_amd.batch import Batch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "rescue_f128.npz")


def _rescued_subset(b, limit, rng):
    _, _, resc = oracle.batch(b, n_threads=8)
    idx = np.nonzero(resc)[0]
    if len(idx) > limit:
        idx = np.sort(rng.choice(idx, limit, replace=False))
    return b.subset(idx)


def _compact(b):
    """Keep only the reads and haplotypes the pair list refers to."""
    ru, ri = np.unique(b.pair_read, return_inverse=True)
    hu, hi = np.unique(b.pair_hap, return_inverse=True)
    reads = [b.read(int(r)) for r in ru]
    haps = [b.hap(int(h)) for h in hu]
    return Batch.from_lists(reads, haps, list(zip(ri.tolist(), hi.tolist())))


def main():
    rng = np.random.default_rng(128)
    parts, groups = [], []
    # 0: the rescued pairs of the reference-made goldens (their fp32 sums are the reference's own: log10_ref is NaN there)
    z = np.load(os.path.join(ROOT, "tests", "golden", "pairhmm_golden.npz"))
    tags = sorted({k.rsplit("_read_bases", 1)[0] for k in z.files if k.endswith("_read_bases")})
    for tag in tags:
        g = {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")}
        b = Batch(g["read_bases"], g["read_qual"], g["read_ins"], g["read_del"], g["read_gcp"], g["read_off"], g["read_len"],
                  g["hap_bases"], g["hap_off"], g["hap_len"], g["pair_read"], g["pair_hap"])
        idx = np.nonzero(g["raw_avx512_s"] < np.float32(1e-28))[0]
        if len(idx):
            parts.append(_compact(b.subset(idx[:60])))
            groups.append(0)
    # 1, 2: BASELINE shapes with 15 % substitutions
    parts.append(_compact(_rescued_subset(synth.make_batch(seed=71, n_regions=3, reads=40, haps=25, read_len=150, hap_len=300, sub_rate=0.15), 360, rng)))
    groups.append(1)
    parts.append(_compact(_rescued_subset(synth.make_batch(seed=72, n_regions=4, reads=20, haps=8, read_len=(50, 250), hap_len=(100, 500), sub_rate=0.15,
                                                           gap_open=45, qual=30), 120, rng)))
    groups.append(2)
    # 3: long reads (config 5's shape, cut at 6 000 x 9 000 so that the quad-precision sweep stays within a minute per pair)
    long_b = synth.make_batch(seed=73, n_regions=10, reads=2, haps=2, read_len=(2000, 6000), hap_len=(6000, 9000))
    parts.append(_compact(long_b.subset(np.arange(0, long_b.n_pairs, 2))))  # 20 pairs
    groups.append(3)
    # 4: reads unrelated to their haplotype: short ones stay finite in fp64, long ones leave even its range
    acgt = np.frombuffer(b"ACGT", np.uint8)
    reads, haps = [], []
    for R, H in [(150, 300), (250, 500), (600, 900), (2500, 3000), (3000, 4000), (4000, 4200)]:
        seq = acgt[rng.integers(0, 4, R)]
        reads.append((seq, np.full(R, 37, np.uint8), np.full(R, 40, np.uint8), np.full(R, 40, np.uint8), np.full(R, 10, np.uint8)))
        haps.append(acgt[rng.integers(0, 4, H)])
    parts.append(Batch.from_lists(reads, haps, [(i, i) for i in range(len(reads))]))
    groups.append(4)

    b = Batch.concat(parts)
    group = np.concatenate([np.full(p.n_pairs, g, np.uint8) for p, g in zip(parts, groups)])
    tabs = oracle.ref_tables() if oracle.ref() is not None else oracle.tables()
    tabs = {k: np.ascontiguousarray(v) for k, v in tabs.items()}
    order = np.argsort(-(b.read_len[b.pair_read].astype(np.int64) * b.hap_len[b.pair_hap]))  # longest first: balances the pool
    out, below = np.zeros(b.n_pairs), np.zeros(b.n_pairs, np.uint8)

    def one(p):
        v, bd = oracle.pair_f128_log10(b, int(p), tabs)
        out[p], below[p] = v, bd
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        list(ex.map(one, order))
    np.savez_compressed(OUT, read_bases=b.read_bases, read_qual=b.read_qual, read_ins=b.read_ins, read_del=b.read_del, read_gcp=b.read_gcp,
                        read_off=b.read_off, read_len=b.read_len, hap_bases=b.hap_bases, hap_off=b.hap_off, hap_len=b.hap_len,
                        pair_read=b.pair_read, pair_hap=b.pair_hap, log10_f128=out, below_denormal=below, group=group,
                        tables_from=np.array("reference Context<double> (oracle/_ref)" if oracle.ref() is not None else "oracle/phmm_oracle.c"))
    print(f"{OUT}: {b.n_pairs} pairs ({np.bincount(group).tolist()} per group), {int(below.sum())} below the fp64 range, "
          f"log10 from {np.nanmin(out[np.isfinite(out)]):.1f} to {np.nanmax(out[np.isfinite(out)]):.1f}")


if __name__ == "__main__":
    main()
