"""Flat (read table, haplotype table, pair list) batch -- the numpy view of ``phmm_batch`` in include/pairhmm_hip.h.

Per pair the semantics are those of the reference's ``::TestCase`` (core/common.h:13-24): read bases as ASCII,
``qual/ins/del/gcp`` as raw phred bytes (not +33), haplotype bases as ASCII.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np


def _u8(x):
    return np.ascontiguousarray(x, dtype=np.uint8)


def _u32(x):
    return np.ascontiguousarray(x, dtype=np.uint32)


@dataclass
class Batch:
    read_bases: np.ndarray
    read_qual: np.ndarray
    read_ins: np.ndarray
    read_del: np.ndarray
    read_gcp: np.ndarray
    read_off: np.ndarray
    read_len: np.ndarray
    hap_bases: np.ndarray
    hap_off: np.ndarray
    hap_len: np.ndarray
    pair_read: np.ndarray
    pair_hap: np.ndarray
    # optional region structure (region r owns reads [region_read[r], region_read[r+1]) etc.; its pairs are the
    # full cross product in (read, hap) order starting at region_pair[r])
    region_read: Optional[np.ndarray] = None
    region_hap: Optional[np.ndarray] = None
    region_pair: Optional[np.ndarray] = None
    meta: dict = field(default_factory=dict)

    def __post_init__(self):
        for k in ("read_bases", "read_qual", "read_ins", "read_del", "read_gcp", "hap_bases"):
            setattr(self, k, _u8(getattr(self, k)))
        for k in ("read_off", "read_len", "hap_off", "hap_len", "pair_read", "pair_hap"):
            setattr(self, k, _u32(getattr(self, k)))
        if self.region_read is not None:
            self.region_read = _u32(self.region_read)
            self.region_hap = _u32(self.region_hap)
            self.region_pair = np.ascontiguousarray(self.region_pair, dtype=np.uint64)

    @property
    def n_reads(self):
        return int(self.read_len.shape[0])

    @property
    def n_haps(self):
        return int(self.hap_len.shape[0])

    @property
    def n_pairs(self):
        return int(self.pair_read.shape[0])

    def cells(self):
        """Total DP cells = sum over pairs of R*H (the numerator of GCUPS)."""
        return int((self.read_len[self.pair_read].astype(np.int64) * self.hap_len[self.pair_hap].astype(np.int64)).sum())

    def algorithmic_bytes(self):
        """3R + H + 4 per pair (read bases, base quals, gap-open quals; hap bases; fp32 result)."""
        r = self.read_len[self.pair_read].astype(np.int64)
        h = self.hap_len[self.pair_hap].astype(np.int64)
        return int((3 * r + h + 4).sum())

    def read(self, k):
        o, n = int(self.read_off[k]), int(self.read_len[k])
        s = slice(o, o + n)
        return self.read_bases[s], self.read_qual[s], self.read_ins[s], self.read_del[s], self.read_gcp[s]

    def hap(self, k):
        o, n = int(self.hap_off[k]), int(self.hap_len[k])
        return self.hap_bases[o:o + n]

    def subset(self, idx):
        """Same tables, a subset of the pair list."""
        idx = np.asarray(idx)
        return Batch(self.read_bases, self.read_qual, self.read_ins, self.read_del, self.read_gcp, self.read_off,
                     self.read_len, self.hap_bases, self.hap_off, self.hap_len, self.pair_read[idx], self.pair_hap[idx],
                     meta=dict(self.meta))

    @staticmethod
    def concat(parts):
        """One batch out of several: tables laid end to end, offsets and pair indices rebased (region structure kept when
        every part has one)."""
        parts = list(parts)
        if len(parts) == 1:
            return parts[0]
        rbytes = np.cumsum([0] + [p.read_bases.shape[0] for p in parts])
        hbytes = np.cumsum([0] + [p.hap_bases.shape[0] for p in parts])
        nr = np.cumsum([0] + [p.n_reads for p in parts])
        nh = np.cumsum([0] + [p.n_haps for p in parts])
        npairs = np.cumsum([0] + [p.n_pairs for p in parts])
        assert rbytes[-1] < 2 ** 32 and hbytes[-1] < 2 ** 32
        cat = lambda k: np.concatenate([getattr(p, k) for p in parts])
        shifted = lambda k, base: np.concatenate([getattr(p, k).astype(np.int64) + int(base[i]) for i, p in enumerate(parts)])
        regions = all(p.region_read is not None for p in parts)
        tail = lambda k, base: np.concatenate([[0]] + [getattr(p, k)[1:].astype(np.int64) + int(base[i]) for i, p in enumerate(parts)])
        return Batch(cat("read_bases"), cat("read_qual"), cat("read_ins"), cat("read_del"), cat("read_gcp"),
                     shifted("read_off", rbytes), cat("read_len"), cat("hap_bases"), shifted("hap_off", hbytes), cat("hap_len"),
                     shifted("pair_read", nr), shifted("pair_hap", nh),
                     region_read=tail("region_read", nr) if regions else None, region_hap=tail("region_hap", nh) if regions else None,
                     region_pair=tail("region_pair", npairs) if regions else None, meta=dict(parts[0].meta, concatenated=len(parts)))

    @staticmethod
    def from_lists(reads, haps, pairs=None):
        """reads: list of (bases, qual, ins, del, gcp) byte-likes; haps: list of byte-likes; pairs: list of
        (read_idx, hap_idx) or None for the full cross product in (read, hap) order."""
        def cat(parts):
            return np.concatenate([np.frombuffer(bytes(p), np.uint8) for p in parts]) if parts else np.zeros(0, np.uint8)
        rl = np.array([len(bytes(r[0])) for r in reads], np.uint32)
        hl = np.array([len(bytes(h)) for h in haps], np.uint32)
        ro = np.concatenate([[0], np.cumsum(rl)[:-1]]).astype(np.uint32) if len(rl) else np.zeros(0, np.uint32)
        ho = np.concatenate([[0], np.cumsum(hl)[:-1]]).astype(np.uint32) if len(hl) else np.zeros(0, np.uint32)
        if pairs is None:
            pr, ph = np.meshgrid(np.arange(len(reads)), np.arange(len(haps)), indexing="ij")
            pr, ph = pr.ravel(), ph.ravel()
        else:
            pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
            pr, ph = pairs[:, 0], pairs[:, 1]
        return Batch(cat([r[0] for r in reads]), cat([r[1] for r in reads]), cat([r[2] for r in reads]),
                     cat([r[3] for r in reads]), cat([r[4] for r in reads]), ro, rl, cat(haps), ho, hl, pr, ph)
