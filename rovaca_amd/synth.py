"""Deterministic synthetic (read, haplotype) batches of the BASELINE.json shapes.

A batch is a list of *regions*, as the caller of the path sees them (haplotypecaller_engine.cpp:250): each region
has a few haplotypes (a random base sequence plus SNP variants of it) and reads sampled from them (a window of one
haplotype with 1 % substitutions and 0.1 % N).  The pair list is the full reads x haplotypes cross product of each
region.  Base qualities are Illumina-like {2: 1 %, 11: 4 %, 25: 15 %, 37: 80 %} after the reference's quality
modification (q < 18 -> 6, MQ 60: pairhmm_internal.cpp:434-439); gap-open 40 (45 capped by the CONSERVATIVE PCR
model, adapter.h:15), gap-continuation 10 (read_record.cpp:205-210).
"""
import numpy as np

from .batch import Batch

_ACGT = np.frombuffer(b"ACGT", np.uint8)
_QV = np.array([6, 6, 25, 37], np.uint8)  # 2 and 11 are below the threshold 18 -> MIN_QUALITY 6
_QP = np.array([0.01, 0.04, 0.15, 0.80])

CONFIGS = {
    # name: (seed, n_regions, reads/region, haps/region, read_len, hap_len)
    "c1": dict(seed=1, n_regions=1, reads=40, haps=25, read_len=100, hap_len=150),  # 1 000 pairs, 100x150
    "c2": dict(seed=2, n_regions=1000, reads=40, haps=25, read_len=150, hap_len=300),  # 1 000 000 pairs, 150x300
    "c3": dict(seed=3, n_regions=1000, reads=40, haps=25, read_len=(50, 250), hap_len=(100, 500)),  # mixed lengths
    # WGS-scale stream: 1e8 pairs over 8 GPUs = 12.5 M pairs per GPU with c3's length distribution, consumed in chunks
    "c4": dict(seed=4, n_regions=12500, reads=40, haps=25, read_len=(50, 250), hap_len=(100, 500)),
    "c5": dict(seed=5, n_regions=40, reads=5, haps=5, read_len=(2000, 10000), hap_len=(2000, 12000)),  # long reads
}


def _rand_len(rng, spec, n, lo_floor=None):
    if isinstance(spec, (tuple, list)):
        lo, hi = spec
        v = rng.integers(lo, hi + 1, size=n)
    else:
        v = np.full(n, int(spec))
    if lo_floor is not None:
        v = np.maximum(v, lo_floor)
    return v.astype(np.int64)


def make_batch(seed=2, n_regions=1000, reads=40, haps=25, read_len=150, hap_len=300, sub_rate=0.01, n_rate=0.001,
               snps_per_hap=3, gap_open=40, gcp=10, qual="illumina", hap_n_rate=0.0):
    """Build a region-structured Batch.  ``qual``: "illumina" or an int for a uniform quality."""
    rng = np.random.default_rng(seed)
    rb, rq, hb = [], [], []
    rlen_all, hlen_all = [], []
    region_read, region_hap, region_pair = [0], [0], [0]
    pr, ph = [], []
    n_r = n_h = n_p = 0
    for _ in range(n_regions):
        H = int(_rand_len(rng, hap_len, 1)[0])
        base = _ACGT[rng.integers(0, 4, size=H)]
        hmat = np.repeat(base[None, :], haps, axis=0)
        for k in range(1, haps):  # hap 0 is the unmodified base sequence
            pos = rng.integers(0, H, size=snps_per_hap)
            hmat[k, pos] = _ACGT[(np.searchsorted(_ACGT, hmat[k, pos]) + rng.integers(1, 4, size=snps_per_hap)) % 4]
        if hap_n_rate > 0:
            hmat[rng.random(hmat.shape) < hap_n_rate] = ord("N")
        rl = np.minimum(_rand_len(rng, read_len, reads), H)
        src = rng.integers(0, haps, size=reads)
        start = (rng.random(reads) * (H - rl + 1)).astype(np.int64)
        for k in range(reads):
            R = int(rl[k])
            seq = hmat[src[k], start[k]:start[k] + R].copy()
            sub = rng.random(R) < sub_rate
            ns = int(sub.sum())
            if ns:
                cur = np.searchsorted(_ACGT, np.where(seq[sub] == ord("N"), ord("A"), seq[sub]))
                seq[sub] = _ACGT[(cur + rng.integers(1, 4, size=ns)) % 4]
            seq[rng.random(R) < n_rate] = ord("N")
            rb.append(seq)
            if qual == "illumina":
                rq.append(_QV[rng.choice(4, size=R, p=_QP)])
            else:
                rq.append(np.full(R, int(qual), np.uint8))
        hb.append(hmat.ravel())
        rlen_all.append(rl)
        hlen_all.append(np.full(haps, H, np.int64))
        a, b = np.meshgrid(np.arange(reads) + n_r, np.arange(haps) + n_h, indexing="ij")
        pr.append(a.ravel())
        ph.append(b.ravel())
        n_r += reads
        n_h += haps
        n_p += reads * haps
        region_read.append(n_r)
        region_hap.append(n_h)
        region_pair.append(n_p)
    rlen = np.concatenate(rlen_all).astype(np.uint32)
    hlen = np.concatenate(hlen_all).astype(np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen, dtype=np.int64)[:-1]]).astype(np.uint32)
    hoff = np.concatenate([[0], np.cumsum(hlen, dtype=np.int64)[:-1]]).astype(np.uint32)
    read_bases = np.concatenate(rb)
    n_bytes = read_bases.shape[0]
    b = Batch(read_bases, np.concatenate(rq), np.full(n_bytes, gap_open, np.uint8), np.full(n_bytes, gap_open, np.uint8),
              np.full(n_bytes, gcp, np.uint8), roff, rlen, np.concatenate(hb), hoff, hlen, np.concatenate(pr),
              np.concatenate(ph), region_read=np.array(region_read), region_hap=np.array(region_hap),
              region_pair=np.array(region_pair),
              meta=dict(seed=seed, n_regions=n_regions, reads=reads, haps=haps, read_len=read_len, hap_len=hap_len))
    return b


def make_config(name, scale=1.0, **over):
    """One of BASELINE.json's configs ("c1", "c2", "c3", "c5"); ``scale`` multiplies the region count."""
    cfg = dict(CONFIGS[name])
    cfg["n_regions"] = max(1, int(round(cfg["n_regions"] * scale)))
    cfg.update(over)
    b = make_batch(**cfg)
    b.meta["config"] = name
    return b
