"""Region-descriptor batches (ABI 5) without a GPU: the host-only binning (pairhmm_binning.cpp) through phmm_plan_bins /
phmm_plan_region_bins, and phmm_compute_regions / the stage calls on the CPU backend of the C ABI against the oracle."""
import numpy as np
import pytest

import oracle
from rovaca_amd import capi, synth
from rovaca_amd.batch import Batch


def _class(R, H):
    K = (R + 15) // 16
    if K > 16 or H > 4096:
        return 17 + (63 - min((R + 127) // 128, 63))
    return K


def _bin_id(k):
    return k if k < 17 else 17 if k < 33 else 18 if k < 57 else 19 if k < 73 else 20


def test_pair_binning_orders_by_class_then_haplotype_length_then_index():
    b = synth.make_batch(seed=31, n_regions=9, reads=14, haps=5, read_len=(20, 900), hap_len=(900, 1400))
    for cross in (False,):
        order, bins = capi.plan_bins(b)
        assert sorted(order.tolist()) == list(range(b.n_pairs))  # a permutation
        R, H = b.read_len[b.pair_read[order]].astype(int), b.hap_len[b.pair_hap[order]].astype(int)
        key = [(_class(r, h), -h, int(p)) for r, h, p in zip(R, H, order)]
        assert key == sorted(key)
        # launches: contiguous, cover the order; striped classes grouped in four size ranges; a single-pass launch runs with the rows
        # per lane of its highest class, and classes share a launch only where everything below the highest is a small remainder
        # (fewer than 8 192 pairs: they ride along instead of paying a launch of their own)
        at = 0
        for rows, first, count in bins:
            assert first == at and count > 0
            ids = [_bin_id(_class(r, h)) for r, h in zip(R[first:first + count], H[first:first + count])]
            if rows:
                assert max(ids) == rows and ids == sorted(ids) and sum(i < rows for i in ids) < 8192
            else:
                assert len(set(ids)) == 1 and ids[0] >= 17
            at += count
        assert at == b.n_pairs


def test_small_launches_ride_along_with_the_next_class(monkeypatch):
    # 3 000 pairs of 151-base reads (10 rows per lane) and a sprinkling of trimmed ones: one launch instead of one per class; without the
    # merge (PHMM_NO_BIN_MERGE, the A/B hook) every class has its own.  A class that is not small keeps its rows per lane.
    reads = [(np.full(n, 65, np.uint8),) * 5 for n in [151] * 60 + [40, 70, 100, 120, 135]]
    haps = [np.full(300, 65, np.uint8)] * 50
    b = Batch.from_lists(reads, haps)
    _, bins = capi.plan_bins(b, cross=True)
    assert [(r, c) for r, _, c in bins] == [(10, 65 * 50)]
    monkeypatch.setenv("PHMM_NO_BIN_MERGE", "1")
    _, plain = capi.plan_bins(b, cross=True)
    assert [(r, c) for r, _, c in plain] == [(3, 50), (5, 50), (7, 50), (8, 50), (9, 50), (10, 3000)]
    # ... and without the master switch no hook is honoured (a production process that inherits PHMM_NO_BIN_MERGE keeps the merge)
    monkeypatch.delenv("PHMM_ENABLE_TEST_HOOKS")
    _, dead = capi.plan_bins(b, cross=True)
    assert [(r, c) for r, _, c in dead] == [(10, 65 * 50)]
    monkeypatch.setenv("PHMM_ENABLE_TEST_HOOKS", "0")
    assert capi.plan_bins(b, cross=True)[1] == dead
    monkeypatch.setenv("PHMM_ENABLE_TEST_HOOKS", "1")
    monkeypatch.delenv("PHMM_NO_BIN_MERGE")
    reads = [(np.full(n, 65, np.uint8),) * 5 for n in [100] * 170 + [151] * 170 + [250]]
    b = Batch.from_lists(reads, haps)
    _, bins = capi.plan_bins(b, cross=True)
    assert [(r, c) for r, _, c in bins] == [(7, 8500), (10, 8500), (16, 50)]  # 8 500 pairs at 7 rows are not made to run at 10


def test_pair_binning_sparse_lengths_take_the_comparison_sort_and_agree():
    # few pairs over a huge length range: the counting table would be larger than the batch
    rng = np.random.default_rng(2)
    reads = [(np.full(n, 65, np.uint8),) * 5 for n in (10, 300, 5000, 17)]
    haps = [np.full(n, 65, np.uint8) for n in (12000, 40, 5001, 700)]
    b = Batch.from_lists(reads, haps)
    order, bins = capi.plan_bins(b)
    R, H = b.read_len[b.pair_read[order]].astype(int), b.hap_len[b.pair_hap[order]].astype(int)
    # the length key is the READ's longest haplotype (round 5: the pairs of a read stay together for the table kernels)
    hr = np.zeros(b.n_reads, int)
    np.maximum.at(hr, b.pair_read, b.hap_len[b.pair_hap].astype(int))
    key = [(_class(r, h), -int(hr[q]), int(p)) for r, h, q, p in zip(R, H, b.pair_read[order], order)]
    assert key == sorted(key) and sum(c for _, _, c in bins) == b.n_pairs


def test_cross_product_binning_equals_the_spelled_out_pair_list():
    b = synth.make_batch(seed=32, n_regions=1, reads=30, haps=7, read_len=(40, 260), hap_len=(260, 420))
    o1, b1 = capi.plan_bins(b)
    o2, b2 = capi.plan_bins(b, cross=True)
    assert np.array_equal(o1, o2) and b1 == b2


def test_region_binning_places_every_read_and_matches_the_pair_plan():
    b = synth.make_batch(seed=33, n_regions=11, reads=9, haps=4, read_len=(30, 700), hap_len=(700, 900))
    reg = capi.regions_of(b)
    slot, pair_first, bins = capi.plan_region_bins(b, reg)
    _, pair_bins = capi.plan_bins(b)
    assert [(r, c) for r, _, c in bins] == [(r, c) for r, _, c in pair_bins]  # the same launches of the same sizes
    assert np.array_equal(pair_first, b.region_pair[:-1].astype(np.uint32))
    # every read's run of `order` lies inside the launch of its class and the runs tile the order exactly
    taken = np.zeros(b.n_pairs, bool)
    for g, (r0, nr, h0, nh) in enumerate(reg):
        hmax = int(b.hap_len[h0:h0 + nh].max())
        for r in range(r0, r0 + nr):
            k = _bin_id(_class(int(b.read_len[r]), hmax))
            if k < 17:  # inside one single-pass launch whose rows per lane are enough for this read
                rows, first, count = next(x for x in bins if x[1] <= slot[r] < x[1] + x[2])
                assert rows >= k and slot[r] + nh <= first + count
            assert not taken[slot[r]:slot[r] + nh].any()
            taken[slot[r]:slot[r] + nh] = True
    assert taken.all()


def test_region_binning_rejects_bad_descriptors():
    b = synth.make_batch(seed=34, n_regions=2, reads=4, haps=2, read_len=50, hap_len=90)
    reg = capi.regions_of(b)
    for bad in ([[0, 4, 0, 2], [2, 4, 2, 2]], [[0, 4, 0, 2], [4, 5, 2, 2]], [[0, 4, 0, 3], [4, 4, 2, 3]], [[0, 0, 0, 2]]):
        with pytest.raises(capi.PhmmError):
            capi.plan_region_bins(b, np.array(bad, np.uint32))
    capi.plan_region_bins(b, reg)


@pytest.mark.parametrize("model", [0, 3])
def test_compute_regions_on_the_cpu_backend_against_the_oracle(model):
    rng = np.random.default_rng(35)
    b = synth.make_batch(seed=35, n_regions=5, reads=12, haps=4, read_len=(40, 300), hap_len=(300, 420), sub_rate=0.08)
    b.read_qual = rng.choice(np.array([2, 11, 25, 37, 41], np.uint8), b.read_bases.shape[0], p=[0.02, 0.05, 0.15, 0.7, 0.08])
    mapq = rng.choice(np.array([60, 60, 29, 12], np.uint8), b.n_reads)
    eng = capi.CpuEngine(n_threads=4)
    raw, l10, st = eng.compute_regions(b, capi.regions_of(b), prep=(mapq, 18, model))
    oraw, ol10, oresc = oracle.batch(oracle.prepare_batch(b, mapq, 18, model), n_threads=4)
    assert np.array_equal(raw.view(np.uint32), oraw.view(np.uint32))
    assert np.array_equal(l10[~oresc], ol10[~oresc]) and st.n_rescued == int(oresc.sum())
    if oresc.any():
        assert np.abs(l10[oresc] - ol10[oresc]).max() < 1e-9
    # the stage calls themselves, written in place, two submissions through one stage
    stg = capi.Stage(eng, 8, 128, 32, 1 << 17, 1024)
    for _ in range(2):
        n = stg.fill(b, capi.regions_of(b), mapq)
        stg.submit(*n, 18, model)
        stg.wait()
        r2, l2, resc = stg.log10(0, b.n_pairs)
        assert np.array_equal(r2.view(np.uint32), raw.view(np.uint32)) and np.array_equal(l2, l10) and resc == st.n_rescued
    # a submission that FAILS (a read of length 0; more reads than the stage holds) must not leave the batch before it readable as its own:
    # the stage is empty afterwards -- phmm_stage_log10 refuses, phmm_stage_pair_first answers 0 -- until a submission succeeds again
    n = stg.fill(b, capi.regions_of(b), mapq)
    stg.read_len[3] = 0
    with pytest.raises(capi.PhmmError, match="length 0"):
        stg.submit(*n, 18, model)
    stg.wait()  # (nothing in flight: returns at once)
    with pytest.raises(capi.PhmmError, match="did not succeed"):
        stg.log10(0, b.n_pairs)
    assert stg.pair_first(1) == 0
    with pytest.raises(capi.PhmmError, match="capacity"):
        stg.submit(n[0], 129, n[2], 18, model)
    with pytest.raises(capi.PhmmError, match="did not succeed"):
        stg.log10(0, 1)
    n = stg.fill(b, capi.regions_of(b), mapq)
    stg.submit(*n, 18, model)
    stg.wait()
    r2, l2, _ = stg.log10(0, b.n_pairs)
    assert np.array_equal(r2.view(np.uint32), raw.view(np.uint32)) and np.array_equal(l2, l10)
    stg.close()
    eng.close()
