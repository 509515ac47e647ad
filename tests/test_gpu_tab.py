"""The table sweep (pairhmm_systolic_tab.h: the prior looked up in an LDS table per read) against the oracle and against the bit-plane
sweep it replaces -- the same operations on the same operands, so the same bits -- on the cases that are its own: haplotypes with N or
other symbols (the workgroup falls back to the bit-plane sweep), workgroups with more reads than table slots (rounds), pair lists
that jump between reads, haplotypes at every byte alignment, every launch form (one class, ragged ranges, uniform budget, stages)."""
import numpy as np
import pytest

import oracle
from rovaca_amd import capi, synth
from rovaca_amd.batch import Batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    e = capi.Engine(0)
    yield e
    e.close()


def _bits(x):
    return np.ascontiguousarray(x).view(np.uint32)


def _both_ways(engine, b, monkeypatch):
    """default (table sweep where it applies) and PHMM_NO_TAB (bit-plane sweep): identical fp32 sums and log10 values"""
    raw, l10, st = engine.compute(b)
    monkeypatch.setenv("PHMM_NO_TAB", "1")
    raw0, l100, st0 = engine.compute(b)
    monkeypatch.delenv("PHMM_NO_TAB")
    assert np.array_equal(_bits(raw), _bits(raw0))
    assert np.array_equal(l10, l100)
    assert int(st.n_rescued) == int(st0.n_rescued)
    return raw, l10, st


def _against_oracle(b, raw, l10):
    oraw, ol10, oresc = oracle.batch(b, n_threads=8)
    bad = np.nonzero(_bits(raw) != _bits(oraw))[0]
    assert bad.size == 0, (bad[:8], raw[bad[:8]], oraw[bad[:8]])
    assert np.array_equal(l10[~oresc], ol10[~oresc])


@pytest.mark.parametrize("gop", ["uniform", "per-base"])
def test_haplotypes_with_n_and_other_symbols_fall_back(engine, gop, monkeypatch):
    if gop == "per-base":
        monkeypatch.setenv("PHMM_NO_UNIFORM_GOP", "1")
    # 2 % N in every haplotype of half the regions, a few bytes outside the alphabet in some: workgroups with and without such pairs
    b = synth.make_batch(seed=31, n_regions=60, reads=24, haps=7, read_len=(100, 160), hap_len=(180, 330), hap_n_rate=0.0)
    hb = b.hap_bases.copy()
    rng = np.random.default_rng(3)
    for h in range(0, b.hap_len.shape[0], 2):
        o, n = int(b.hap_off[h]), int(b.hap_len[h])
        hit = rng.random(n) < 0.02
        hb[o:o + n][hit] = ord("N")
        if h % 6 == 0:
            hb[o + int(rng.integers(0, n))] = ord("x")
    b2 = Batch(b.read_bases, b.read_qual, b.read_ins, b.read_del, b.read_gcp, b.read_off, b.read_len, hb, b.hap_off, b.hap_len, b.pair_read,
               b.pair_hap)
    raw, l10, _ = _both_ways(engine, b2, monkeypatch)
    _against_oracle(b2, raw, l10)


@pytest.mark.parametrize("slots", [1, 2, 3, 5])
def test_workgroups_with_more_reads_than_table_slots_sweep_in_rounds(engine, slots, monkeypatch):
    monkeypatch.setenv("PHMM_TAB_SLOTS", str(slots))
    # 3 haplotypes per read: a workgroup's 16 pairs span 6-7 reads
    b = synth.make_batch(seed=32, n_regions=40, reads=30, haps=3, read_len=(60, 150), hap_len=(150, 300))
    raw, l10, _ = _both_ways(engine, b, monkeypatch)
    _against_oracle(b, raw, l10)


def test_pair_list_that_jumps_between_reads(engine, monkeypatch):
    # the pairs of a region in random order: neighbours in the list rarely share a read (16 tables per workgroup would be needed)
    b = synth.make_batch(seed=33, n_regions=30, reads=20, haps=8, read_len=150, hap_len=(250, 320))
    rng = np.random.default_rng(4)
    perm = rng.permutation(b.pair_read.shape[0])
    b2 = Batch(b.read_bases, b.read_qual, b.read_ins, b.read_del, b.read_gcp, b.read_off, b.read_len, b.hap_bases, b.hap_off, b.hap_len,
               b.pair_read[perm].copy(), b.pair_hap[perm].copy())
    raw, l10, _ = _both_ways(engine, b2, monkeypatch)
    _against_oracle(b2, raw, l10)


@pytest.mark.parametrize("plan", ["uniform-budget", "ranges", "per-class"])
@pytest.mark.parametrize("gop", ["uniform", "per-base"])
def test_every_launch_form_and_rows_per_lane(engine, plan, gop, monkeypatch):
    if gop == "per-base":
        monkeypatch.setenv("PHMM_NO_UNIFORM_GOP", "1")
    if plan == "ranges":
        monkeypatch.setenv("PHMM_RAGGED_UNIFORM_BELOW", "0")
    elif plan == "per-class":
        monkeypatch.setenv("PHMM_NO_RAGGED", "1")
        monkeypatch.setenv("PHMM_NO_BIN_MERGE", "1")
    monkeypatch.setenv("PHMM_FORCE_LANES", "16")
    # reads of 1..256 bases (every K), haplotypes of any length and byte alignment, regions of 2..9 haplotypes
    rng = np.random.default_rng(7)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    rb, rq, hb, rl, hl, pr, ph = [], [], [], [], [], [], []
    n_r = n_h = 0
    for g in range(90):
        nh = int(rng.integers(2, 10))
        H0 = int(rng.integers(20, 520))
        base = acgt[rng.integers(0, 4, H0 + 8)]
        for k in range(nh):
            H = max(1, H0 + int(rng.integers(-6, 7)))
            hap = base[:H].copy()
            hap[rng.random(H) < 0.01] = ord("G")
            hb.append(hap)
            hl.append(H)
        nr = int(rng.integers(3, 12))
        for k in range(nr):
            R = int(rng.integers(1, 257))
            seq = base[:R].copy() if R <= base.shape[0] else acgt[rng.integers(0, 4, R)]
            seq[rng.random(R) < 0.02] = ord("T")
            if rng.random() < 0.1:
                seq[int(rng.integers(0, R))] = ord("N")
            rb.append(seq)
            rq.append(rng.choice(np.array([6, 20, 30, 40], np.uint8), R))
            rl.append(R)
            for j in range(nh):
                pr.append(n_r + k)
                ph.append(n_h + j)
        n_r += nr
        n_h += nh
    rlen, hlen = np.array(rl, np.uint32), np.array(hl, np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen)[:-1]]).astype(np.uint32)
    hoff = np.concatenate([[0], np.cumsum(hlen)[:-1]]).astype(np.uint32)
    bases = np.concatenate(rb)
    n = bases.shape[0]
    b = Batch(bases, np.concatenate(rq), np.full(n, 40, np.uint8), np.full(n, 40, np.uint8), np.full(n, 10, np.uint8), roff, rlen,
              np.concatenate(hb), hoff, hlen, np.array(pr, np.uint32), np.array(ph, np.uint32))
    raw, l10, _ = _both_ways(engine, b, monkeypatch)
    _against_oracle(b, raw, l10)


def test_resident_batch_and_sample_of_config2(engine, monkeypatch):
    b = synth.make_config("c2", scale=0.05)
    rb = engine.upload(b)
    rb.sweep()
    raw, l10, _ = rb.download()
    monkeypatch.setenv("PHMM_NO_TAB", "1")
    rb.sweep()
    raw0, l100, _ = rb.download()
    rb.release()
    assert np.array_equal(_bits(raw), _bits(raw0)) and np.array_equal(l10, l100)
    sample = np.random.default_rng(1).choice(b.n_pairs, 2000, replace=False)
    oraw, _, _ = oracle.batch(b.subset(sample), n_threads=8)
    assert np.array_equal(_bits(raw[sample]), _bits(oraw))


def test_long_haplotypes_leave_no_room_for_tables_and_take_the_bit_plane_kernels(engine, monkeypatch):
    # 16 haplotypes of 1 500-2 500 columns are 50-80 KB of entries: no table fits next to them at four workgroups per CU, so these
    # classes keep the bit-plane kernels (pairhmm_sweep.cpp: tab_slots_of); shorter haplotypes in the same batch take the table sweep
    monkeypatch.setenv("PHMM_FORCE_LANES", "16")
    parts = [synth.make_batch(seed=41, n_regions=6, reads=20, haps=6, read_len=(60, 150), hap_len=(1500, 2500)),
             synth.make_batch(seed=42, n_regions=6, reads=20, haps=6, read_len=(60, 150), hap_len=(200, 400))]
    for b in parts + [Batch.concat(parts)]:
        raw, l10, _ = _both_ways(engine, b, monkeypatch)
        _against_oracle(b, raw, l10)


def _long_pair(rng, R, H, n_rate=0.0, other=False):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, H)]
    a0 = int(rng.integers(0, H - R + 1))
    seq = hap[a0:a0 + R].copy()
    bad = rng.random(R) < 0.12
    seq[bad] = acgt[rng.integers(0, 4, int(bad.sum()))]
    if n_rate:
        hap = hap.copy()
        hap[rng.random(H) < n_rate] = ord("N")
        seq[rng.random(R) < n_rate] = ord("N")
    if other:
        hap[int(rng.integers(0, H))] = ord("x")
        seq[int(rng.integers(0, R))] = ord("y")
    q = rng.choice(np.array([8, 20, 30, 37], np.uint8), R)
    g = np.full(R, 40, np.uint8)
    return (seq, q, g, g, np.full(R, 10, np.uint8)), hap


@pytest.mark.parametrize("order", ["0", "1"])
def test_fp64_band_kernel_table_form_is_the_plane_form(engine, order, monkeypatch):
    # long pairs that underflow fp32: the fp64 band kernel (four waves per pair) in its table form -- six symbols, so N and other bytes
    # in haplotype and read need no other path -- against the bit-plane form (PHMM_NO_BAND_TAB) and, for the smaller pairs, the oracle;
    # both orders of passes (fp32 first: the kernel walks a rescue list; fp64 first: every pair)
    monkeypatch.setenv("PHMM_F64_FIRST", order)
    rng = np.random.default_rng(61)
    reads, haps = [], []
    for R, H, nr, oth in ((2100, 2600, 0.0, False), (2500, 3000, 0.01, False), (2300, 4100, 0.0, True), (3000, 3001, 0.02, True), (2050, 5000, 0.0, False)):
        rd, hp = _long_pair(rng, R, H, nr, oth)
        reads.append(rd)
        haps.append(hp)
    b = Batch.from_lists(reads, haps, [(0, 0), (1, 1), (2, 2), (3, 3), (4, 4), (0, 2), (1, 0), (3, 1)])
    raw, l10, st = engine.compute(b)
    monkeypatch.setenv("PHMM_NO_BAND_TAB", "1")
    raw0, l100, st0 = engine.compute(b)
    monkeypatch.delenv("PHMM_NO_BAND_TAB")
    assert np.array_equal(l10, l100, equal_nan=True) and int(st.n_rescued) == int(st0.n_rescued) > 0
    assert np.array_equal(_bits(raw), _bits(raw0))
    small = np.array([0, 1, 6])
    oraw, ol10, oresc = oracle.batch(b.subset(small), n_threads=3)
    fin = np.isfinite(ol10)
    assert np.array_equal(np.isfinite(l10[small]), fin) and (not fin.any() or np.abs(l10[small][fin] - ol10[fin]).max() <= 1e-9)

