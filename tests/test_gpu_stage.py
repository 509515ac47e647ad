"""Region-descriptor batches (ABI 5): phmm_compute_regions and the pinned stage, on the GPU, against the pair-list path (itself
pinned to the oracle by test_gpu_parity.py) and against the oracle directly."""
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


def _raw(seed, **kw):
    """A synthetic batch with RAW BAM-style qualities and mapping qualities (what the adapter hands over)."""
    rng = np.random.default_rng(seed)
    b = synth.make_batch(seed=seed, **kw)
    b.read_qual = rng.choice(np.array([2, 11, 25, 37, 41], np.uint8), b.read_bases.shape[0], p=[0.02, 0.05, 0.15, 0.7, 0.08])
    mapq = rng.choice(np.array([60, 60, 60, 29, 12], np.uint8), b.n_reads)
    return b, mapq


@pytest.mark.parametrize("model", [0, 3, 1])
def test_compute_regions_equals_the_pair_list_path(engine, model):
    # mixed lengths: every single-pass class, reads of 257..512 (one wave per pair) and striped reads
    b, mapq = _raw(5, n_regions=7, reads=24, haps=6, read_len=(30, 700), hap_len=(700, 1100))
    reg = capi.regions_of(b)
    raw0, l0, st0 = engine.compute(b, prep=(mapq, 18, model))
    raw1, l1, st1 = engine.compute_regions(b, reg, prep=(mapq, 18, model))
    assert np.array_equal(raw0.view(np.uint32), raw1.view(np.uint32))
    assert np.array_equal(l0, l1, equal_nan=True)
    assert st1.cells == st0.cells == b.cells() and st1.n_rescued == st0.n_rescued


def test_compute_regions_against_the_oracle_with_noisy_reads(engine):
    # 15 % substitutions: a good share of the pairs underflow fp32 and are re-run in fp64 from the stage's queue
    b, mapq = _raw(6, n_regions=4, reads=30, haps=5, read_len=(80, 200), hap_len=(200, 350), sub_rate=0.15)
    reg = capi.regions_of(b)
    raw, l10, st = engine.compute_regions(b, reg, prep=(mapq, 18, 3))
    prepared = oracle.prepare_batch(b, mapq, 18, 3)
    oraw, ol10, oresc = oracle.batch(prepared, n_threads=4)
    assert st.n_rescued == int(oresc.sum()) > 0
    assert np.array_equal(raw.view(np.uint32), oraw.view(np.uint32))
    assert np.array_equal(l10[~oresc], ol10[~oresc])
    assert np.abs(l10[oresc] - ol10[oresc]).max() < 1e-9


def test_stage_written_in_place_with_gaps_and_shared_haplotypes(engine):
    # what the adapter's workers do: reads at 4-byte-aligned slots with a count in front, regions written in any order,
    # two regions sharing their haplotypes; several submissions through one stage
    b, mapq = _raw(7, n_regions=3, reads=10, haps=4, read_len=(60, 151), hap_len=(250, 330))
    st = capi.Stage(engine, 16, 256, 64, 1 << 17, 4096)
    for round_ in range(3):
        at = 4 + 64 * round_  # (the used part of the arena moves between submissions)
        for r in range(b.n_reads):
            o, n = int(b.read_off[r]), int(b.read_len[r])
            at = (at + 3) & ~3
            st.bytes[at:at + 4] = 0xEE  # the Bases count a real decode_base leaves in front of the bases
            off = at + 4
            st.bytes[off:off + n] = b.read_bases[o:o + n]
            st.bytes[off + n:off + 2 * n] = b.read_qual[o:o + n]
            st.read_off[r], st.read_qual_off[r], st.read_len[r], st.read_mapq[r] = off, off + n, n, mapq[r]
            at = off + 2 * n
        for h in range(b.n_haps):
            o, n = int(b.hap_off[h]), int(b.hap_len[h])
            st.bytes[at:at + n] = b.hap_bases[o:o + n]
            st.hap_off[h], st.hap_len[h] = at, n
            at += n
        reg = capi.regions_of(b).copy()
        if round_ == 1:
            reg[1, 2:] = reg[0, 2:]  # region 1 evaluated against region 0's haplotypes
        st.regions[:len(reg)] = reg
        st.submit(len(reg), b.n_reads, b.n_haps, 18, 3)
        st.wait()
        n_pairs = int((reg[:, 1].astype(np.int64) * reg[:, 3]).sum())
        raw, l10, _ = st.log10(0, n_pairs)
        pr = np.concatenate([np.repeat(np.arange(g[0], g[0] + g[1]), g[3]) for g in reg])
        ph = np.concatenate([np.tile(np.arange(g[2], g[2] + g[3]), g[1]) for g in reg])
        as_pairs = Batch(b.read_bases, b.read_qual, b.read_ins, b.read_del, b.read_gcp, b.read_off, b.read_len, b.hap_bases, b.hap_off,
                         b.hap_len, pr, ph)
        want_raw, want_l10, _ = engine.compute(as_pairs, prep=(mapq, 18, 3))
        assert np.array_equal(raw.view(np.uint32), want_raw.view(np.uint32))
        assert np.array_equal(l10, want_l10)
        assert [st.pair_first(g) for g in range(len(reg))] == list(np.concatenate([[0], np.cumsum(reg[:, 1] * reg[:, 3])[:-1]]))
    st.close()


def test_stage_rejects_what_does_not_fit_or_overlaps(engine):
    b, mapq = _raw(8, n_regions=2, reads=4, haps=2, read_len=50, hap_len=90)
    st = capi.Stage(engine, 4, 16, 8, 8192, 64)
    n = st.fill(b, capi.regions_of(b), mapq)
    st.regions[1, 0] = 2  # region 1's reads overlap region 0's
    with pytest.raises(capi.PhmmError):
        st.submit(*n)
    st.fill(b, capi.regions_of(b), mapq)
    st.read_len[3] = 0
    with pytest.raises(capi.PhmmError):
        st.submit(*n)
    st.fill(b, capi.regions_of(b), mapq)
    with pytest.raises(capi.PhmmError):
        st.submit(5, 8, 4)  # more regions than the stage holds
    st.submit(*n)  # and the stage is still usable
    st.wait()
    raw, l10, _ = st.log10(0, b.n_pairs)
    want_raw, want_l10, _ = engine.compute(b, prep=(mapq, 18, 3))
    assert np.array_equal(raw.view(np.uint32), want_raw.view(np.uint32)) and np.array_equal(l10, want_l10)
    # ... and a submission that fails AFTER one that succeeded does not leave the earlier batch's numbers readable as its own
    st.read_len[3] = 0
    with pytest.raises(capi.PhmmError):
        st.submit(*n)
    st.wait()
    with pytest.raises(capi.PhmmError, match="did not succeed"):
        st.log10(0, b.n_pairs)
    assert st.pair_first(1) == 0
    st.fill(b, capi.regions_of(b), mapq)
    st.submit(*n)
    st.close()  # with the submission still in flight: destroy waits for it


@pytest.mark.parametrize("model", [3, 1, 2, 0])
def test_stage_read_preparation_on_tricky_reads(engine, model):
    # tandem repeats, homopolymer runs, N, lengths 1..259 (one region per read): the stage's set-up launch does H3 + H4 with the
    # bit-mask form of the repeat search; the result must be that of the batch prepared by the oracle's own routines
    from test_host_logic import _reads
    rng = np.random.default_rng(12)
    seqs = _reads(rng, 120)
    seqs += [np.frombuffer(b"A" * 200 + b"C", np.uint8).copy(), np.frombuffer(b"ACGTACGT" * 30 + b"T", np.uint8).copy(),
             np.frombuffer(b"G" + b"AC" * 120, np.uint8).copy()]  # runs beyond 64 and 160 positions
    acgt = np.frombuffer(b"ACGT", np.uint8)
    reads, haps = [], []
    mapq = rng.choice(np.array([60, 60, 29, 12, 0], np.uint8), len(seqs))
    for sq in seqs:
        n = len(sq)
        q = rng.choice(np.array([2, 11, 17, 18, 25, 37, 41], np.uint8), n)
        reads.append((sq, q, q, q, q))
        H = n + int(rng.integers(5, 80))
        hap = acgt[rng.integers(0, 4, H)]
        st = int(rng.integers(0, H - n + 1))
        hap[st:st + n] = np.where(sq == ord("N"), ord("A"), sq)
        haps.append(hap)
    raw = Batch.from_lists(reads, haps, [(i, i) for i in range(len(seqs))])
    reg = np.array([[i, 1, i, 1] for i in range(len(seqs))], np.uint32)
    got_raw, got_l10, _ = engine.compute_regions(raw, reg, prep=(mapq, 18, model))
    want_raw, want_l10, _ = oracle.batch(oracle.prepare_batch(raw, mapq, 18, model), n_threads=4)[:3]
    assert np.array_equal(got_raw.view(np.uint32), want_raw.view(np.uint32))


def test_stage_with_reads_beyond_the_set_up_launch_lds_copy(engine):
    # reads above 4 096 bases: the PCR model's repeat search reads global memory instead of the LDS copy; striped kernels; fp64 re-run
    b, mapq = _raw(9, n_regions=2, reads=3, haps=2, read_len=(4200, 5000), hap_len=(5200, 5600))
    rng = np.random.default_rng(9)
    o, n = int(b.read_off[1]), int(b.read_len[1])
    b.read_bases[o + 100:o + 400] = np.frombuffer(b"ACG", np.uint8)[np.arange(300) % 3]  # a long tandem repeat inside one read
    raw, l10, st = engine.compute_regions(b, capi.regions_of(b), prep=(mapq, 18, 3))
    oraw, ol10, oresc = oracle.batch(oracle.prepare_batch(b, mapq, 18, 3), n_threads=8)
    assert np.array_equal(raw.view(np.uint32), oraw.view(np.uint32)) and int(st.n_rescued) == int(oresc.sum())
    fin = np.isfinite(ol10)
    assert np.array_equal(np.isfinite(l10), fin) and np.abs(l10[fin] - ol10[fin]).max() < 1e-9
