"""Parity of the HIP path (through the C ABI) against the reference: golden vectors produced by the reference's own
AVX-512 kernels, and the oracle on fresh seeded inputs.  fp32 sums are compared BIT FOR BIT; the stated tolerance of
BASELINE.json (max |delta log10| <= 1e-5) is therefore met with delta = 0 on accepted pairs.  Rescued pairs (fp32 sum
< 1e-28) are compared with the scalar-double oracle (the reference's fp64 kernels are defective)."""
import numpy as np
import pytest

import oracle
from conftest import GOLDEN_TAGS, golden_batch
from rovaca_amd import capi, synth
from rovaca_amd.batch import Batch

pytestmark = pytest.mark.gpu

TOL_LOG10 = 1e-5  # BASELINE.json north_star
TOL_RESCUE = 1e-9  # fp64 vs scalar double oracle, same operation order

# How the rows-per-lane classes of a mixed-length batch reach the device -- the same sweep of the same pair either way, so the same bits:
# "ragged": the classes share launches, every workgroup picks its K (pairhmm_kernels_ragged.hip; the default for Rovaca-case batches);
# "ride-along": one launch per class, small classes handed to the next class up (PHMM_NO_RAGGED; the default until round 3, and still
# the plan of batches with general per-base gap penalties); "per-class": one launch per class, nothing merged.
# "ragged-ranges": the ragged form of launches of 2^18 pairs or more -- one kernel per range of K that shares a register budget -- forced
# onto small batches (the default below that size is the uniform-budget kernel: one launch, big-K classes as 32 lanes per pair).
LAUNCH_PLANS = ("ragged", "ragged-ranges", "ride-along", "per-class")


def _set_launch_plan(monkeypatch, plan):
    if plan == "ragged-ranges":
        monkeypatch.setenv("PHMM_RAGGED_UNIFORM_BELOW", "0")
    elif plan != "ragged":
        monkeypatch.setenv("PHMM_NO_RAGGED", "1")
    if plan == "per-class":
        monkeypatch.setenv("PHMM_NO_BIN_MERGE", "1")


@pytest.fixture(scope="module")
def engine():
    e = capi.Engine(0)
    yield e
    e.close()


def _check_against_oracle(engine, b, expect_bit_exact=True):
    raw, l10, st = engine.compute(b)
    oraw, ol10, oresc = oracle.batch(b, n_threads=8)
    if expect_bit_exact:
        bad = np.nonzero(raw.view(np.uint32) != oraw.view(np.uint32))[0]
        assert bad.size == 0, (bad[:8], raw[bad[:8]], oraw[bad[:8]])
    acc = ~oresc
    assert np.array_equal(l10[acc], ol10[acc])
    if oresc.any():
        fin = np.isfinite(ol10[oresc])  # hopeless pairs underflow even fp64 (scale 2^1020): -inf in the oracle and here
        assert np.array_equal(np.isfinite(l10[oresc]), fin) and np.array_equal(l10[oresc][~fin], ol10[oresc][~fin])
        if fin.any():
            d = np.abs(l10[oresc][fin] - ol10[oresc][fin])
            assert d.max() <= TOL_RESCUE, d.max()
    assert int(st.n_rescued) == int(oresc.sum())
    assert int(st.cells) == b.cells()
    return raw, l10


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_golden_vectors_bit_exact(engine, golden, tag):
    b = golden_batch(golden, tag)
    raw, l10, st = engine.compute(b)
    ref = golden[f"{tag}_raw_avx512_s"]
    bad = np.nonzero(raw.view(np.uint32) != ref.view(np.uint32))[0]
    assert bad.size == 0, (bad[:8], raw[bad[:8]], ref[bad[:8]])
    ok = ~np.isnan(golden[f"{tag}_log10_ref"])
    assert np.abs(l10[ok] - golden[f"{tag}_log10_ref"][ok]).max() <= TOL_LOG10
    assert np.array_equal(l10[ok], golden[f"{tag}_log10_ref"][ok])
    assert int(st.n_rescued) == int((~ok).sum())


def test_config1_1k_pairs_100x150(engine):
    _check_against_oracle(engine, synth.make_config("c1"))


def test_config2_sample_150x300(engine):
    _check_against_oracle(engine, synth.make_config("c2", scale=0.02))


@pytest.mark.parametrize("launches", LAUNCH_PLANS)
@pytest.mark.parametrize("lanes", [16, 64])
def test_config3_sample_mixed_lengths(engine, lanes, launches, monkeypatch):
    _set_launch_plan(monkeypatch, launches)
    monkeypatch.setenv("PHMM_FORCE_LANES", str(lanes))
    _check_against_oracle(engine, synth.make_config("c3", scale=0.02))


@pytest.mark.parametrize("launches", LAUNCH_PLANS)
@pytest.mark.parametrize("lanes", [16, 64])
def test_every_rows_per_lane_bin(engine, lanes, launches, monkeypatch):
    _set_launch_plan(monkeypatch, launches)
    # read lengths 1..256 hit every K = ceil(R/16) instantiation, including exact multiples of 16, in both shapes of the
    # single-pass kernel: four pairs per wave (16 lanes each) and one pair per wave (64 lanes, K = ceil(R/64))
    monkeypatch.setenv("PHMM_FORCE_LANES", str(lanes))
    rng = np.random.default_rng(5)
    reads, haps, pairs = [], [], []
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for R in list(range(1, 40)) + [47, 48, 49, 63, 64, 65, 95, 96, 97, 111, 112, 113, 127, 128, 129, 143, 144, 145, 159,
                                     160, 161, 175, 176, 177, 191, 192, 193, 207, 208, 209, 223, 224, 225, 239, 240, 241,
                                     255, 256]:
        H = int(rng.integers(max(1, R // 2), 2 * R + 20))
        hap = acgt[rng.integers(0, 4, H)]
        seq = acgt[rng.integers(0, 4, R)]
        n = min(R, H)
        seq[:n] = hap[:n]
        seq[rng.random(R) < 0.03] = ord("C")
        q = rng.choice(np.array([6, 20, 30, 40], np.uint8), R)
        g = rng.choice(np.array([45, 40, 38], np.uint8), R)
        reads.append((seq, q, g, g, np.full(R, 10, np.uint8)))
        haps.append(hap)
        pairs.append((len(reads) - 1, len(haps) - 1))
    _check_against_oracle(engine, Batch.from_lists(reads, haps, pairs))


@pytest.mark.parametrize("launches", LAUNCH_PLANS)
@pytest.mark.parametrize("lanes", [16, 64])
def test_general_gap_parameters(engine, lanes, launches, monkeypatch):
    _set_launch_plan(monkeypatch, launches)
    # per-base ins / del / gcp all different: exercises the non-uniform-gcp kernel variant
    monkeypatch.setenv("PHMM_FORCE_LANES", str(lanes))
    rng = np.random.default_rng(6)
    b = synth.make_batch(seed=61, n_regions=4, reads=12, haps=4, read_len=(20, 250), hap_len=(40, 400))
    n = b.read_bases.shape[0]
    b.read_ins = rng.integers(10, 46, n).astype(np.uint8)
    b.read_del = rng.integers(10, 46, n).astype(np.uint8)
    b.read_gcp = rng.integers(5, 30, n).astype(np.uint8)
    b.read_qual = (rng.integers(6, 60, n) | (rng.random(n) < 0.1) * 128).astype(np.uint8)
    _check_against_oracle(engine, b)


@pytest.mark.parametrize("rows", [0, 4, 5, 6])
def test_rescue_path_fp64(engine, rows, monkeypatch):
    # the fp64 re-run is instantiated for 4, 5 and 6 rows per lane (0: let the launch choose by padding)
    if rows:
        monkeypatch.setenv("PHMM_FORCE_F64_ROWS", str(rows))
    b = synth.make_batch(seed=62, n_regions=3, reads=16, haps=4, read_len=(100, 250), hap_len=(260, 500), sub_rate=0.15,
                         qual=30)
    raw, l10 = _check_against_oracle(engine, b)
    assert (raw < np.float32(1e-28)).sum() > 10


def test_rescue_queue_per_length_class(engine, monkeypatch):
    # large batches re-run every single-pass length class from its own queue; forced here on a small mixed-length batch
    monkeypatch.setenv("PHMM_FORCE_QUEUE_PER_CLASS", "1")
    b = synth.make_batch(seed=64, n_regions=6, reads=20, haps=6, read_len=(20, 250), hap_len=(260, 500), sub_rate=0.15, qual=30)
    raw, l10 = _check_against_oracle(engine, b)
    assert (raw < np.float32(1e-28)).sum() > 50


def test_cross_product_mode_and_N_symbols(engine):
    b = synth.make_batch(seed=63, n_regions=1, reads=23, haps=7, read_len=(30, 120), hap_len=200, n_rate=0.02,
                         hap_n_rate=0.01)
    r1, l1, _ = engine.compute(b)
    r2, l2, _ = engine.compute(b, cross=True)
    assert np.array_equal(r1.view(np.uint32), r2.view(np.uint32)) and np.array_equal(l1, l2)
    _check_against_oracle(engine, b)


def test_empty_and_invalid_inputs(engine):
    b = synth.make_batch(seed=64, n_regions=1, reads=3, haps=2, read_len=20, hap_len=30)
    empty = b.subset(np.zeros(0, np.int64))
    raw, l10, st = engine.compute(empty)
    assert raw.shape == (0,) and l10.shape == (0,)
    bad = b.subset(np.arange(b.n_pairs))
    bad.pair_hap = bad.pair_hap.copy()
    bad.pair_hap[0] = 99
    with pytest.raises(capi.PhmmError, match="outside the tables"):
        engine.compute(bad)
    z = b.subset(np.arange(b.n_pairs))
    z.read_len = z.read_len.copy()
    z.read_len[1] = 0
    with pytest.raises(capi.PhmmError, match="length 0"):
        engine.compute(z)


def test_resident_sweeps_are_idempotent(engine):
    b = synth.make_config("c2", scale=0.01)
    rb = engine.upload(b)
    rb.sweep()
    r1, l1, _ = rb.download()
    ms, ms32 = rb.sweep_timed(3)
    r2, l2, _ = rb.download()
    rb.release()
    assert ms > 0 and ms32 > 0
    assert np.array_equal(r1.view(np.uint32), r2.view(np.uint32)) and np.array_equal(l1, l2)


def test_full_size_properties_config2(engine):
    """BASELINE config 2 at full size (1M pairs, 150x300): size-independent properties instead of the scalar oracle.
    (a) every region's hap 0 is the unmodified base sequence: reads sampled from it score at least as well against it
        as the oracle says on a sample; (b) likelihoods are invariant to the order of the pair list; (c) a seeded 2 %
        sample is bit-exact against the oracle."""
    b = synth.make_config("c2")
    assert b.n_pairs == 1_000_000
    raw, l10, st = engine.compute(b)
    assert int(st.cells) == 150 * 300 * 1_000_000
    assert np.all(np.isfinite(l10)) and l10.max() <= 0.0
    rng = np.random.default_rng(7)
    perm = rng.permutation(b.n_pairs)
    raw_p, l10_p, _ = engine.compute(b.subset(perm))
    assert np.array_equal(raw_p.view(np.uint32), raw[perm].view(np.uint32))
    idx = np.sort(rng.choice(b.n_pairs, 20000, replace=False))
    oraw, ol10, _ = oracle.batch(b.subset(idx), n_threads=16)
    assert np.array_equal(raw[idx].view(np.uint32), oraw.view(np.uint32))
    assert np.array_equal(l10[idx], ol10)


def test_long_reads_striped_kernel(engine):
    # reads above 256 bases take the striped kernel (stripes of 128 rows chained through a carry row in global memory)
    rng = np.random.default_rng(8)
    reads, haps, pairs = [], [], []
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for R, H in [(257, 300), (258, 257), (300, 700), (383, 384), (384, 385), (385, 1000), (512, 512), (640, 129),
                 (1000, 1500), (1537, 2000), (2000, 2100), (129, 5000), (40, 4097)]:
        hap = acgt[rng.integers(0, 4, H)]
        st = int(rng.integers(0, max(1, H - R + 1)))
        seq = np.resize(hap[st:st + R], R).copy()
        seq[rng.random(R) < 0.02] = ord("G")
        q = rng.choice(np.array([6, 25, 37], np.uint8), R, p=[0.05, 0.15, 0.8])
        g = np.full(R, 40, np.uint8)
        reads.append((seq, q, g, g, np.full(R, 10, np.uint8)))
        haps.append(hap)
        pairs.append((len(reads) - 1, len(haps) - 1))
    pairs += [(0, 3), (5, 2), (9, 9), (3, 8)]  # mixed stripe counts inside one wave
    _check_against_oracle(engine, Batch.from_lists(reads, haps, pairs))


def test_config5_sample_long_reads(engine):
    # BASELINE config 5 shape at reduced count: reads 2-10 kbp x haps up to 12 kbp.  The scalar oracle needs ~1 s per
    # 1e8 cells, so the sample is 6 pairs; fixed pre-scaling underflows fp32 on such reads -> most pairs are rescued.
    b = synth.make_batch(seed=5, n_regions=2, reads=3, haps=1, read_len=(2000, 6000), hap_len=(6000, 9000))
    raw, l10, st = engine.compute(b)
    oraw, ol10, oresc = oracle.batch(b, n_threads=6)
    assert np.array_equal(raw.view(np.uint32), oraw.view(np.uint32))
    fin = np.isfinite(ol10)
    assert np.array_equal(l10[fin], ol10[fin])  # fp32-accepted identical, fp64-rescued identical (same op order)
    assert np.array_equal(np.isfinite(l10), fin)
    assert int(st.n_rescued) == int(oresc.sum())


def test_device_side_read_preparation_matches_host_logic(engine):
    # quality modification (H3) and PCR indel model (H4) on the device.  Integer work: the prepared
    # read table must equal the oracle's byte for byte, and the likelihoods must equal those of the host-prepared batch.
    from test_host_logic import _reads
    rng = np.random.default_rng(12)
    seqs = _reads(rng, 120)  # tandem repeats, homopolymer runs, N, random; lengths 1..259
    acgt = np.frombuffer(b"ACGT", np.uint8)
    reads, haps, pairs = [], [], []
    O = oracle.lib()
    mapq = rng.choice(np.array([60, 60, 29, 12, 0], np.uint8), len(seqs))
    raw_q = []
    for i, sq in enumerate(seqs):
        n = len(sq)
        q = rng.choice(np.array([2, 11, 17, 18, 25, 37, 41], np.uint8), n)
        raw_q.append(q)
        hq = q.copy()
        O.phmm_oracle_modify_quals(hq, n, int(mapq[i]), 18)
        g = np.full(n, 45, np.uint8)
        O.phmm_oracle_apply_pcr_error_model(sq, g, n, 3)
        reads.append((sq, hq, g, g, np.full(n, 10, np.uint8)))
        H = n + int(rng.integers(5, 80))
        hap = acgt[rng.integers(0, 4, H)]
        st = int(rng.integers(0, H - n + 1))
        hap[st:st + n] = np.where(sq == ord("N"), ord("A"), sq)
        haps.append(hap)
        pairs.append((i, i))
    host = Batch.from_lists(reads, haps, pairs)
    for model in (3, 1, 0):
        if model != 3:  # re-derive the expected gap-opens for the other models
            gops = []
            for sq in seqs:
                g = np.full(len(sq), 45, np.uint8)
                O.phmm_oracle_apply_pcr_error_model(sq, g, len(sq), model)
                gops.append(g)
            host.read_ins = np.concatenate(gops)
            host.read_del = host.read_ins.copy()
        raw = Batch(host.read_bases, np.concatenate(raw_q), host.read_ins * 0, host.read_del * 0, host.read_gcp * 0,
                    host.read_off, host.read_len, host.hap_bases, host.hap_off, host.hap_len, host.pair_read, host.pair_hap)
        rb = engine.upload(raw, prep=(mapq, 18, model))
        q, i, d, c = rb.read_table()
        assert np.array_equal(q, host.read_qual) and np.array_equal(i, host.read_ins) and np.array_equal(d, host.read_del)
        assert np.array_equal(c, host.read_gcp)
        rb.sweep()
        r_dev, l_dev, _ = rb.download()
        rb.release()
        r_host, l_host, _ = engine.compute(host)
        assert np.array_equal(r_dev.view(np.uint32), r_host.view(np.uint32)) and np.array_equal(l_dev, l_host, equal_nan=True)


@pytest.mark.parametrize("lanes", [16, 64])
@pytest.mark.parametrize("waves", [1, 4, 8])
def test_team_kernel_every_width(engine, waves, lanes, monkeypatch):
    # the multi-wavefront-per-pair pipeline must give the same bits whatever the number of waves per team and whether a
    # wave holds four pairs (16 lanes each) or one (64 lanes); mixed read lengths inside a team exercise the late-join of
    # pairs with fewer stripes, and the fp64 re-run takes the same path
    monkeypatch.setenv("PHMM_FORCE_TEAM_WAVES", str(waves))
    monkeypatch.setenv("PHMM_FORCE_TEAM_LANES", str(lanes))
    rng = np.random.default_rng(20 + waves)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    reads, haps, pairs = [], [], []
    for R, H in [(1500, 2500), (900, 2500), (300, 2400), (1301, 2300), (2049, 2200), (700, 2100), (1100, 2050)]:
        hap = acgt[rng.integers(0, 4, H)]
        st = int(rng.integers(0, H - R + 1))
        seq = hap[st:st + R].copy()
        seq[rng.random(R) < 0.01] = ord("T")
        q = rng.choice(np.array([6, 25, 37], np.uint8), R, p=[0.05, 0.15, 0.8])
        g = np.full(R, 40, np.uint8)
        reads.append((seq, q, g, g, np.full(R, 10, np.uint8)))
        haps.append(hap)
        pairs.append((len(reads) - 1, len(haps) - 1))
    pairs += [(0, 1), (2, 0), (4, 6)]
    _check_against_oracle(engine, Batch.from_lists(reads, haps, pairs))


@pytest.mark.parametrize("lanes", [16, 64])
@pytest.mark.parametrize("waves", [1, 4])
def test_striped_rescue_list_in_processing_order_with_holes(engine, waves, lanes, monkeypatch):
    # The rescue queue of a striped size range is a list in processing order: the fp32 launch writes an underflowed pair's id at the
    # pair's own position, the fp64 launch walks every position and skips the holes (KernelArgs::sparse_order; PHMM_RESCUE_APPEND=1 is
    # the appended list of rounds 1-3).  Long reads of which only some underflow -- clean ones among noisy ones, in no particular order
    # -- must give the oracle's bits in both forms, for one wave per team and for the band kernel, four pairs per wave and one,
    # and a batch of long reads none of which underflows must launch a re-run that finds nothing.
    monkeypatch.setenv("PHMM_FORCE_TEAM_WAVES", str(waves))
    monkeypatch.setenv("PHMM_FORCE_TEAM_LANES", str(lanes))
    rng = np.random.default_rng(77 + waves + lanes)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    reads, haps, pairs = [], [], []
    for i in range(22):
        R, H = int(rng.integers(520, 1400)), int(rng.integers(1500, 2600))
        hap = acgt[rng.integers(0, 4, H)]
        st = int(rng.integers(0, H - R + 1))
        seq = hap[st:st + R].copy()
        noisy = i % 3 != 1
        if noisy:
            sub = rng.random(R) < 0.04
            seq[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        q = np.full(R, 37, np.uint8) if not noisy else rng.choice(np.array([11, 25, 37], np.uint8), R, p=[0.1, 0.2, 0.7])
        g = np.full(R, 45, np.uint8)
        reads.append((seq, q, g, g, np.full(R, 10, np.uint8)))
        haps.append(hap)
        pairs.append((i, i))
    pairs += [(1, 0), (4, 4), (7, 6), (10, 10), (13, 2)]
    b = Batch.from_lists(reads, haps, pairs)
    raw, l10 = _check_against_oracle(engine, b)
    resc = raw < np.float32(1e-28)
    assert 5 < int(resc.sum()) < b.n_pairs - 5  # holes and entries both
    monkeypatch.setenv("PHMM_RESCUE_APPEND", "1")
    araw, al10, ast = engine.compute(b)
    monkeypatch.delenv("PHMM_RESCUE_APPEND")
    assert np.array_equal(araw.view(np.uint32), raw.view(np.uint32)) and np.array_equal(al10, l10)
    clean = np.flatnonzero(~resc)
    craw, cl10, cst = engine.compute(b.subset(clean))
    assert int(cst.n_rescued) == 0 and np.array_equal(cl10, l10[clean])


def _long_reads_around_the_fp32_threshold(seed, n_per_step=2):
    # reads of 600-1300 bases with exactly m substitutions at Q37, m = 6 ... 26: about 4.2 decades per substitution, so the fp32 sums
    # (scale 2^120 = 1.3e36, accepted from 1e-28) run from comfortably accepted through the threshold to flushed to zero
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    reads, haps, pairs = [], [], []
    for m in range(6, 27):
        for _ in range(n_per_step):
            R, H = int(rng.integers(600, 1300)), int(rng.integers(1400, 2400))
            hap = acgt[rng.integers(0, 4, H)]
            st = int(rng.integers(0, H - R + 1))
            seq = hap[st:st + R].copy()
            at = rng.choice(R, m, replace=False)
            seq[at] = acgt[(np.searchsorted(acgt, seq[at]) + rng.integers(1, 4, m)) % 4]
            g = np.full(R, 45, np.uint8)
            reads.append((seq, np.full(R, 37, np.uint8), g, g, np.full(R, 10, np.uint8)))
            haps.append(hap)
            pairs.append((len(reads) - 1, len(haps) - 1))
    order = rng.permutation(len(pairs))
    return Batch.from_lists(reads, haps, [pairs[i] for i in order])


@pytest.mark.parametrize("lanes", [16, 64])
@pytest.mark.parametrize("waves", [1, 4])
def test_fp64_first_decides_every_pair_as_fp32_first_does(engine, waves, lanes, monkeypatch):
    # A range of long pairs that has been underflowing is swept fp64 FIRST (every pair), and the fp32 kernel then runs only for the
    # pairs whose fp64 sum says fp32 might not underflow (pairhmm_sweep.cpp, "fp64 first"); PHMM_F64_FIRST=1 / 0 force the order.  Pairs
    # on both sides of the fp32 threshold and close to it: the log10 likelihoods, the set of re-run pairs and the fp32 sums of the
    # accepted pairs are bit for bit those of the fp32-first order and the oracle's; the fp32 sum of a pair that provably underflows is
    # reported as 0.
    monkeypatch.setenv("PHMM_FORCE_TEAM_WAVES", str(waves))
    monkeypatch.setenv("PHMM_FORCE_TEAM_LANES", str(lanes))
    b = _long_reads_around_the_fp32_threshold(300 + waves + lanes)
    monkeypatch.setenv("PHMM_F64_FIRST", "0")
    raw, l10 = _check_against_oracle(engine, b)
    resc = raw < np.float32(1e-28)
    assert 8 < int(resc.sum()) < b.n_pairs - 8
    assert ((raw > 0) & resc).any()  # the margin: underflowed, but not to zero
    monkeypatch.setenv("PHMM_F64_FIRST", "1")
    for _ in range(2):
        fraw, fl10, fst = engine.compute(b)
        assert np.array_equal(fl10, l10)
        assert np.array_equal(fraw[~resc].view(np.uint32), raw[~resc].view(np.uint32))
        assert (fraw[resc] < np.float32(1e-28)).all() and int(fst.n_rescued) == int(resc.sum())
        assert ((fraw[resc] == raw[resc]) | (fraw[resc] == 0)).all() and (fraw[resc] == 0).any()
    # resident: the two orders in turn on one uploaded batch
    rb = engine.upload(b)
    try:
        for first in ("1", "0", "1"):
            monkeypatch.setenv("PHMM_F64_FIRST", first)
            rb.sweep()
            rraw, rl10, rst = rb.download()
            assert np.array_equal(rl10, l10) and int(rst.n_rescued) == int(resc.sum())
            assert np.array_equal(rraw[~resc].view(np.uint32), raw[~resc].view(np.uint32))
    finally:
        rb.release()


def test_fp64_first_is_chosen_by_what_the_last_pass_saw(monkeypatch):
    # the default (no hook): a batch's first pass is fp32 first; when at least three quarters of a striped range underflowed, the next
    # pass over it runs fp64 first (the fp32 phase of the pass all but disappears), and a new batch in the same context starts from
    # what the context has seen; a range of clean long reads stays fp32 first.  Same answers throughout.
    monkeypatch.delenv("PHMM_F64_FIRST", raising=False)
    eng = capi.Engine(0)
    try:
        noisy = synth.make_batch(seed=95, n_regions=3, reads=6, haps=4, read_len=(1500, 3000), hap_len=(3200, 4200), sub_rate=0.02)
        rb = eng.upload(noisy)
        ms1, ms32_1 = rb.sweep_timed(1)
        raw1, l10_1, st1 = rb.download()
        assert int(st1.n_rescued) == noisy.n_pairs
        ms2, ms32_2 = rb.sweep_timed(1)
        raw2, l10_2, st2 = rb.download()
        assert np.array_equal(l10_1, l10_2) and int(st2.n_rescued) == noisy.n_pairs and (raw2 == 0).all()
        assert ms32_2 < 0.5 * ms32_1, (ms32_1, ms32_2)
        rb.release()
        # a new batch of the same kind: fp64 first from its first pass
        noisy2 = synth.make_batch(seed=96, n_regions=3, reads=6, haps=4, read_len=(1500, 3000), hap_len=(3200, 4200), sub_rate=0.02)
        rb = eng.upload(noisy2)
        ms3, ms32_3 = rb.sweep_timed(1)
        raw3, l10_3, st3 = rb.download()
        rb.release()
        assert ms32_3 < 0.5 * ms32_1, (ms32_1, ms32_3)
        monkeypatch.setenv("PHMM_F64_FIRST", "0")
        _, ref3, _ = eng.compute(noisy2)
        monkeypatch.delenv("PHMM_F64_FIRST", raising=False)
        assert np.array_equal(l10_3, ref3)
        # clean long reads: the guess from the context's history is wrong once (fp64 first, every pair sent on to the fp32 kernel), the
        # next pass is fp32 first again; the answers never change
        clean = synth.make_batch(seed=97, n_regions=3, reads=6, haps=4, read_len=(1500, 3000), hap_len=(3200, 4200), sub_rate=0.0, qual=40)
        rb = eng.upload(clean)
        outs = []
        for _ in range(3):
            rb.sweep()
            outs.append(rb.download())
        rb.release()
        assert all(int(o[2].n_rescued) == 0 for o in outs)
        assert all(np.array_equal(o[1], outs[0][1]) and np.array_equal(o[0].view(np.uint32), outs[0][0].view(np.uint32)) for o in outs)
        oraw, ol10, oresc = oracle.batch(clean, n_threads=8)
        assert not oresc.any() and np.array_equal(outs[0][0].view(np.uint32), oraw.view(np.uint32)) and np.array_equal(outs[0][1], ol10)
    finally:
        eng.close()


@pytest.mark.parametrize("wide_single_pass", [True, False])
def test_reads_of_257_to_512_bases(engine, wide_single_pass, monkeypatch):
    # still one pass with a whole wave per pair (K = ceil(R/64) up to 8), or the team kernel when that is switched off
    if not wide_single_pass:
        monkeypatch.setenv("PHMM_NO_WIDE_SINGLE_PASS", "1")
    rng = np.random.default_rng(93)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    reads, haps, pairs = [], [], []
    for R in (257, 300, 320, 321, 384, 385, 400, 448, 449, 500, 512):
        H = int(rng.integers(R, 2 * R + 100))
        hap = acgt[rng.integers(0, 4, H)]
        a = int(rng.integers(0, H - R + 1))
        seq = hap[a:a + R].copy()
        seq[rng.random(R) < 0.01] = ord("A")
        q = rng.choice(np.array([6, 25, 37], np.uint8), R, p=[0.05, 0.15, 0.8])
        g = np.full(R, 40, np.uint8)
        reads.append((seq, q, g, g, np.full(R, 10, np.uint8)))
        haps.append(hap)
        pairs.append((len(reads) - 1, len(haps) - 1))
    pairs += [(0, 3), (5, 10), (10, 10)]
    _check_against_oracle(engine, Batch.from_lists(reads, haps, pairs))


@pytest.mark.parametrize("skip_zero_supersteps", [True, False])
def test_long_noisy_reads_with_and_without_the_zero_superstep_shortcut(engine, skip_zero_supersteps, monkeypatch):
    # long reads spend most of their sweep on cells that are exactly zero; the team kernel skips supersteps whose inputs and
    # state are all zero.  Same bits either way, and the same as the oracle.
    if not skip_zero_supersteps:
        monkeypatch.setenv("PHMM_NO_EARLY_EXIT", "1")
    b = synth.make_batch(seed=91, n_regions=2, reads=4, haps=3, read_len=(1500, 4000), hap_len=(1800, 4500), sub_rate=0.02)
    raw, l10 = _check_against_oracle(engine, b)
    assert (raw == 0).all()  # fp32 underflows for good on reads this long; the fp64 re-run decides


def test_team_pipeline_with_a_haplotype_shorter_than_the_pipeline(engine):
    # found by tests/debug/fuzz_gpu.py: the team shape (waves per pair) is chosen for the longest haplotype of the launch; a
    # pair of the same launch whose haplotype is shorter than lag * waves * 32 columns must still see its carry written
    # before it is read (the kernel pads its stripes with idle supersteps)
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    haps = [acgt[rng.integers(0, 4, n)] for n in (4800, 314, 90, 700)]
    reads = []
    for R in (1745, 2500, 900, 400):
        seq = acgt[rng.integers(0, 4, R)]
        q = rng.integers(6, 41, R).astype(np.uint8)
        g = np.full(R, 40, np.uint8)
        reads.append((seq, q, g, g, np.full(R, 10, np.uint8)))
    pairs = [(r, h) for r in range(4) for h in range(4)] * 3
    _check_against_oracle(engine, Batch.from_lists(reads, haps, pairs))


def test_symbols_outside_ACGTN_never_match(engine):
    # pairhmm_internal.cpp:27-41: lower case, IUPAC codes, '-' ... map to code 0 and match nothing, not even themselves
    rng = np.random.default_rng(31)
    b = synth.make_batch(seed=31, n_regions=2, reads=12, haps=4, read_len=(30, 180), hap_len=(200, 320), n_rate=0.01,
                         hap_n_rate=0.01)
    odd = np.frombuffer(b"acgtnRYKM-*.", np.uint8)
    rb = b.read_bases.copy()
    hb = b.hap_bases.copy()
    rb[rng.random(rb.shape[0]) < 0.03] = rng.choice(odd, int((rng.random(rb.shape[0]) < 0.03).sum()) or 1)[0]
    idx = np.nonzero(rng.random(rb.shape[0]) < 0.03)[0]
    rb[idx] = rng.choice(odd, idx.size)
    idx = np.nonzero(rng.random(hb.shape[0]) < 0.02)[0]
    hb[idx] = rng.choice(odd, idx.size)
    b.read_bases, b.hap_bases = rb, hb
    _check_against_oracle(engine, b)


@pytest.mark.parametrize("waves", [4, 8])
def test_band_kernel_pair_joining_late_beside_a_stripe_that_began_mid_row(engine, waves, monkeypatch):
    # found by tests/debug/fuzz_gpu.py (seed 417): four pairs per wave with different stripe counts.  When a pair with fewer
    # stripes begins (its boundary row is non-zero everywhere, so the wave sweeps the whole row), the wave's other pairs read the
    # carry of a stripe above that may have begun its sweep in the middle of the row -- the blocks before that point were never
    # written and must read as zero.  Reads that sit at the far end of a long haplotype put the band there; noisy reads make the
    # true result underflow to exactly zero, so any stale value shows.
    monkeypatch.setenv("PHMM_FORCE_TEAM_WAVES", str(waves))
    monkeypatch.setenv("PHMM_FORCE_TEAM_LANES", "16")
    rng = np.random.default_rng(417)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    haps = [acgt[rng.integers(0, 4, n)] for n in (3603, 3500, 2900, 3603)]
    reads = []
    for R, h, at_end, noise in ((1500, 0, True, 0.0), (796, 0, True, 0.3), (300, 1, False, 0.0), (1990, 3, True, 0.01), (520, 2, True, 0.3),
                                (1400, 1, False, 0.02), (260, 3, True, 0.0), (1100, 2, True, 0.3)):
        hap = haps[h]
        a = len(hap) - R if at_end else int(rng.integers(0, len(hap) - R + 1))
        seq = hap[a:a + R].copy()
        bad = rng.random(R) < noise
        seq[bad] = acgt[rng.integers(0, 4, int(bad.sum()))]
        q = rng.integers(6, 41, R).astype(np.uint8)
        g = np.full(R, 40, np.uint8)
        reads.append((seq, q, g, g, np.full(R, 10, np.uint8)))
    pairs = [(r, h) for r in range(len(reads)) for h in range(len(haps))]
    _check_against_oracle(engine, Batch.from_lists(reads, haps, pairs))


@pytest.mark.parametrize("launches", LAUNCH_PLANS)
@pytest.mark.parametrize("lanes", [16, 64])
def test_uniform_gap_open_kernel_is_the_general_one(engine, lanes, launches, monkeypatch):
    _set_launch_plan(monkeypatch, launches)
    # batches with one gap-open quality throughout (PCR-free data, the synthetic configs) take the variant that shares p_mm / p_mx
    # and reuses M * p_mx for the Y of the next step: same bits as the Rovaca-case kernel (PHMM_NO_UNIFORM_GOP=1) and as the oracle,
    # for every rows-per-lane value, with padding rows, N and foreign symbols; a batch with ONE deviating base must not take it
    monkeypatch.setenv("PHMM_FORCE_LANES", str(lanes))
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGTN", np.uint8)
    reads, haps, pairs = [], [], []
    for R in list(range(1, 36)) + [47, 48, 49, 63, 64, 65, 100, 127, 128, 129, 150, 151, 160, 161, 175, 176, 177, 200, 239, 240, 241, 255, 256]:
        H = int(rng.integers(max(R, 20), R + 300))
        hap = acgt[rng.integers(0, 5 if R % 7 == 0 else 4, H)]
        a0 = int(rng.integers(0, H - R + 1))
        seq = hap[a0:a0 + R].copy()
        seq[rng.random(R) < 0.03] = acgt[rng.integers(0, 5)]
        q = rng.choice(np.array([6, 18, 25, 37, 41], np.uint8), R)
        g = np.full(R, 45, np.uint8)
        reads.append((seq, q, g, g.copy(), np.full(R, 10, np.uint8)))
        haps.append(hap)
        pairs.append((len(reads) - 1, len(haps) - 1))
    pairs += [(i, (i * 7 + 3) % len(haps)) for i in range(len(reads))]
    b = Batch.from_lists(reads, haps, pairs)
    raw, l10 = _check_against_oracle(engine, b)
    monkeypatch.setenv("PHMM_NO_UNIFORM_GOP", "1")
    raw2, l10_2, _ = engine.compute(b)
    monkeypatch.delenv("PHMM_NO_UNIFORM_GOP")
    assert np.array_equal(raw.view(np.uint32), raw2.view(np.uint32)) and np.array_equal(l10, l10_2, equal_nan=True)
    b.read_ins[len(b.read_ins) // 2] = 39  # one base with another gap-open (del aliases ins here: both arrays must change)
    b.read_del[len(b.read_del) // 2] = 39
    _check_against_oracle(engine, b)
    # device-side read preparation without a PCR model leaves one gap-open too
    mapq = np.full(b.n_reads, 60, np.uint8)
    praw, _, _ = engine.compute(b, prep=(mapq, 0, 0))
    b.read_ins[:] = 45
    b.read_del[:] = 45
    hraw, _, _ = engine.compute(b)
    assert np.array_equal(praw.view(np.uint32), hraw.view(np.uint32))


def test_xcd_local_workgroup_order_changes_no_result(engine, monkeypatch):
    # big single-pass launches hand runs of 64 consecutive workgroups to one XCD (KernelArgs::xcd_run_log2): the grid is rounded up to
    # whole rounds of 8 runs and every hardware workgroup takes another slice of the processing order.  Forced onto launches of every
    # size here -- far fewer workgroups than one round, just below / at / above a round, several rounds, mixed lengths -- the sums must be
    # those of the plain order and of the oracle, and every pair must be written exactly once
    monkeypatch.setenv("PHMM_FORCE_LANES", "16")
    for n_regions, reads, haps in ((1, 3, 2), (8, 32, 32), (9, 32, 29), (40, 40, 25)):
        b = synth.make_batch(seed=300 + n_regions, n_regions=n_regions, reads=reads, haps=haps, read_len=(60, 160), hap_len=(150, 330))
        monkeypatch.setenv("PHMM_XCD_LOCAL_MIN_PAIRS", "1")
        raw, l10, _ = engine.compute(b)
        monkeypatch.delenv("PHMM_XCD_LOCAL_MIN_PAIRS")
        monkeypatch.setenv("PHMM_NO_XCD_LOCAL_ORDER", "1")
        raw0, l10_0, _ = engine.compute(b)
        monkeypatch.delenv("PHMM_NO_XCD_LOCAL_ORDER")
        assert np.array_equal(raw.view(np.uint32), raw0.view(np.uint32)) and np.array_equal(l10, l10_0, equal_nan=True)
        k = np.random.default_rng(n_regions).choice(b.n_pairs, min(b.n_pairs, 300), replace=False)
        want = oracle.batch(b.subset(k))[0]
        assert np.array_equal(raw[k].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("lanes", [16, 64])
@pytest.mark.parametrize("pcr_model", [0, 3])
def test_ragged_launches_are_the_per_class_launches(engine, lanes, pcr_model, monkeypatch):
    # One launch for many rows-per-lane classes (pairhmm_kernels_ragged.hip) against one launch per class (PHMM_NO_RAGGED), same batch:
    # every class 1..16, classes with a single pair, a class that fills exactly one workgroup and one that spills into a second, rescued
    # pairs in several classes (noisy reads), each class with its own rescue queue (PHMM_FORCE_QUEUE_PER_CLASS) or all in one, the
    # XCD-local workgroup order forced onto the ragged grid, both kernel families (one gap-open quality: PCR model NONE; per-base
    # gap-opens: CONSERVATIVE) through device-side read preparation.  Identical bits, identical rescue counts, fewer launches.
    monkeypatch.setenv("PHMM_FORCE_LANES", str(lanes))
    rng = np.random.default_rng(90 + pcr_model)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    haps = [acgt[rng.integers(0, 4, int(n))] for n in rng.integers(260, 520, 12)]
    reads = []
    lens = list(rng.integers(1, 257, 150)) + [16] * 16 + [33] * 17 + [256, 255, 241, 240, 1, 2]
    for R in lens:
        R = int(R)
        hap = haps[int(rng.integers(0, len(haps)))]
        a0 = int(rng.integers(0, len(hap) - R + 1))
        seq = hap[a0:a0 + R].copy()
        noise = 0.25 if rng.random() < 0.2 else 0.01
        bad = rng.random(R) < noise
        seq[bad] = acgt[rng.integers(0, 4, int(bad.sum()))]
        q = rng.choice(np.array([8, 20, 30, 37], np.uint8), R)
        g = np.full(R, 45, np.uint8)
        reads.append((seq, q, g, g.copy(), np.full(R, 10, np.uint8)))
    b = Batch.from_lists(reads, haps)  # the full cross product
    mapq = np.full(b.n_reads, 60, np.uint8)
    prep = (mapq, 0, pcr_model)  # (no quality threshold: the oracle below sees the same qualities)
    for per_class_queues in (False, True):
        if per_class_queues:
            monkeypatch.setenv("PHMM_FORCE_QUEUE_PER_CLASS", "1")
        monkeypatch.setenv("PHMM_NO_RAGGED", "1")
        monkeypatch.setenv("PHMM_NO_BIN_MERGE", "1")  # (or the small classes of this batch would all ride along in one launch)
        raw0, l10_0, st0 = engine.compute(b, prep=prep)
        monkeypatch.delenv("PHMM_NO_RAGGED")
        monkeypatch.delenv("PHMM_NO_BIN_MERGE")
        raw1, l10_1, st1 = engine.compute(b, prep=prep)  # the uniform-budget kernel: one launch (four pairs per wave forced: lanes 16 / 32)
        monkeypatch.setenv("PHMM_RAGGED_UNIFORM_BELOW", "0")
        raw4, l10_4, st4 = engine.compute(b, prep=prep)  # one kernel per K range that shares a register budget: up to three
        monkeypatch.setenv("PHMM_XCD_LOCAL_MIN_PAIRS", "1")
        raw2, l10_2, st2 = engine.compute(b, prep=prep)
        monkeypatch.delenv("PHMM_RAGGED_UNIFORM_BELOW")
        raw5, l10_5, st5 = engine.compute(b, prep=prep)  # (the XCD-local order on the uniform kernel's grid)
        monkeypatch.delenv("PHMM_XCD_LOCAL_MIN_PAIRS")
        # the fp64 re-run: queues that take the same kernel share a launch (the default) or get one launch each
        monkeypatch.setenv("PHMM_NO_RESCUE_MERGE", "1")
        raw3, l10_3, st3 = engine.compute(b, prep=prep)
        monkeypatch.delenv("PHMM_NO_RESCUE_MERGE")
        for raw, l10, st in ((raw1, l10_1, st1), (raw2, l10_2, st2), (raw3, l10_3, st3), (raw4, l10_4, st4), (raw5, l10_5, st5)):
            assert np.array_equal(raw.view(np.uint32), raw0.view(np.uint32))
            assert np.array_equal(l10, l10_0, equal_nan=True)
            assert int(st.n_rescued) == int(st0.n_rescued) > 0
        assert int(st1.n_launches) == 1 and int(st4.n_launches) <= 3 < int(st0.n_launches)
        # (one pair per wave: every class fits one kernel's budget anyway; four per wave: so does every class of the table sweep that
        # replaced the bit-plane sweeps in round 5 -- those needed three kernels of different budgets: PHMM_NO_TAB below)
        assert int(st4.n_launches) == 1
        monkeypatch.setenv("PHMM_NO_TAB", "1")
        monkeypatch.setenv("PHMM_RAGGED_UNIFORM_BELOW", "0")
        raw6, l10_6, st6 = engine.compute(b, prep=prep)
        monkeypatch.delenv("PHMM_NO_TAB")
        monkeypatch.delenv("PHMM_RAGGED_UNIFORM_BELOW")
        assert np.array_equal(raw6.view(np.uint32), raw0.view(np.uint32)) and int(st6.n_launches) == (1 if lanes == 64 else 3)
    # and the oracle, on a sample (the batch prepared on the host the same way)
    if pcr_model == 0:
        k = rng.choice(b.n_pairs, 400, replace=False)
        want = oracle.batch(b.subset(k))[0]
        assert np.array_equal(raw1[k].view(np.uint32), want.view(np.uint32))
