"""The drop-in adapter end to end on the GPU: rovaca::call_pairhmm (C++ host code over the C ABI) against the oracle's
restatement of call_old_pairhmm / call_new_pairhmm002 (pairhmm_internal.cpp:505-753), including the in/out mutation of
the read vector and the [hap][read] layout with NaN tail."""
import numpy as np
import pytest

import oracle
from rovaca_amd import hostapi, synth

pytestmark = pytest.mark.gpu


def _raw_region(seed, reads, haps, read_len, hap_len, sub_rate=0.01, bad_reads=0):
    rng = np.random.default_rng(seed)
    b = synth.make_batch(seed=seed, n_regions=1, reads=reads, haps=haps, read_len=read_len, hap_len=hap_len, sub_rate=sub_rate)
    n = b.read_bases.shape[0]
    qual = rng.choice(np.array([2, 11, 25, 37, 41], np.uint8), n, p=[0.02, 0.05, 0.15, 0.7, 0.08])  # raw BAM quals
    mq = rng.choice(np.array([60, 60, 60, 29, 12], np.uint8), b.n_reads)
    bases = b.read_bases.copy()
    for r in rng.choice(b.n_reads, bad_reads, replace=False):  # unrelated reads: must be filtered out (H8)
        o, L = int(b.read_off[r]), int(b.read_len[r])
        bases[o:o + L] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
    return bases, qual, b.read_off, b.read_len, mq, b.hap_bases, b.hap_off, b.hap_len


@pytest.fixture(scope="module", autouse=True)
def installed():
    assert hostapi.init_pairhmm_ptr_gpu(0)


@pytest.mark.parametrize("model", ["NONE", "CONSERVATIVE", "HOSTILE"])
def test_region_matches_oracle(model):
    args = _raw_region(21, reads=48, haps=9, read_len=(60, 151), hap_len=320, bad_reads=5)
    got, surv = hostapi.call_region(*args, min_quality_threshold=18, pcr_model=model)
    n_reads, n_haps = len(args[3]), len(args[7])
    want = np.full(n_haps * n_reads, np.nan)
    wsurv = np.zeros(n_reads, np.uint32)
    k = oracle.lib().phmm_oracle_call_region(args[0], args[1], args[2], args[3], args[4], n_reads, args[5], args[6], args[7],
                                             n_haps, 18, hostapi.PCR[model], want, wsurv)
    want = want.reshape(n_haps, n_reads)
    assert len(surv) == k and np.array_equal(surv, wsurv[:k])
    assert k < n_reads  # the unrelated reads were dropped
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[:, :k], want[:, :k])  # fp32 path bit-identical, rescued pairs capped to best - 4.5
    assert np.all(np.isnan(got[:, k:]))


def test_small_region_and_empty_inputs():
    args = _raw_region(22, reads=3, haps=2, read_len=40, hap_len=70)
    got, surv = hostapi.call_region(*args)
    assert got.shape == (2, 3) and len(surv) == 3 and np.all(np.isfinite(got))
    e8, e32 = np.zeros(0, np.uint8), np.zeros(0, np.uint32)
    got, surv = hostapi.call_region(e8, e8, e32, e32, e8, args[5], args[6], args[7])  # no reads -> {} (:508-510)
    assert got.shape == (2, 0) and len(surv) == 0


def test_concurrent_workers():
    # the caller runs this path from many asio workers at once (haplotypecaller_engine.cpp:119)
    import threading
    regions = [_raw_region(30 + i, reads=20, haps=5, read_len=(50, 150), hap_len=260) for i in range(8)]
    single = [hostapi.call_region(*r) for r in regions]
    out = [None] * len(regions)

    def work(i):
        out[i] = hostapi.call_region(*regions[i])

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(regions))]
    [t.start() for t in th]
    [t.join() for t in th]
    for (a, sa), (b, sb) in zip(single, out):
        assert np.array_equal(sa, sb) and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("depth", [1, 2, 5])
def test_begin_finish_form_keeps_regions_in_flight(depth):
    # SURVEY.md 8f next-1 ("return futures"): call_gpu_pairhmm_begin hands a region to the batcher, call_gpu_pairhmm_finish
    # collects it; with `depth` regions in flight per worker the results are those of the synchronous call, region by region,
    # reads dropped by H8 and empty regions included
    regions = [_raw_region(70 + i, reads=6 + 5 * (i % 7), haps=2 + i % 6, read_len=(40, 150), hap_len=200 + 17 * i,
                           bad_reads=(i % 3)) for i in range(14)]
    e8, e32 = np.zeros(0, np.uint8), np.zeros(0, np.uint32)
    regions.insert(5, (e8, e8, e32, e32, e8, regions[0][5], regions[0][6], regions[0][7]))  # a region without reads -> {}
    single = [hostapi.call_region(*r) for r in regions]
    r0, s0 = hostapi.batching_stats()
    got = hostapi.call_regions_in_flight(regions, depth)
    r1, s1 = hostapi.batching_stats()
    assert len(got) == len(regions)
    for (a, sa), (b, sb) in zip(single, got):
        assert a.shape == b.shape and np.array_equal(sa, sb) and np.array_equal(a, b, equal_nan=True)
    assert r1 - r0 == len(regions) - 1  # the empty region never reaches the device
    assert s1 - s0 <= r1 - r0  # (whether one worker's regions share a submission depends on which dispatcher is free)


def test_cross_region_batching_coalesces_concurrent_workers():
    # 32 workers x 8 regions each; the dispatcher must merge concurrent submissions into fewer
    # device batches and every worker must still get exactly its own region's matrix
    import threading
    import time
    regions = [_raw_region(100 + i, reads=24, haps=6, read_len=(60, 150), hap_len=280, bad_reads=2) for i in range(32)]
    want = []
    for r in regions:
        n_reads, n_haps = len(r[3]), len(r[7])
        w = np.full(n_haps * n_reads, np.nan)
        ws = np.zeros(n_reads, np.uint32)
        k = oracle.lib().phmm_oracle_call_region(r[0], r[1], r[2], r[3], r[4], n_reads, r[5], r[6], r[7], n_haps, 18, 3, w, ws)
        want.append((w.reshape(n_haps, n_reads), ws[:k].copy()))
    r0, s0 = hostapi.batching_stats()
    errors = []

    def work(i):
        for _ in range(8):
            got, surv = hostapi.call_region(*regions[i])
            if not (np.array_equal(surv, want[i][1]) and np.array_equal(got, want[i][0], equal_nan=True)):
                errors.append(i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(32)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    r1, s1 = hostapi.batching_stats()
    assert not errors
    assert r1 - r0 == 32 * 8
    assert s1 - s0 < r1 - r0, "no coalescing happened"
    print(f"256 regions from 32 workers in {dt*1e3:.1f} ms: {s1 - s0} device submissions")


def test_device_side_preparation_against_the_hand_derived_answers():
    # tests/golden/host_logic_known_answers.json: worked out on paper from pairhmm_internal.cpp (H3, H4).  The device-side
    # read preparation (setup_kernel) must give exactly those bytes.
    import json
    import os
    from rovaca_amd import capi
    from rovaca_amd.batch import Batch
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_logic_known_answers.json")))
    eng = capi.Engine(0)
    hap = np.frombuffer(b"ACGTACGTAC", np.uint8)
    for model in (0, 1, 2, 3):
        cases = [c for c in fx["apply_pcr_error_model"] if c["model"] == model and isinstance(c["gop_in"], int)]
        reads = []
        for c in cases:
            s = np.frombuffer(c["bases"].encode(), np.uint8)
            z = np.zeros(len(s), np.uint8)
            reads.append((s, np.full(len(s), 30, np.uint8), z, z, z))
        b = Batch.from_lists(reads, [hap], [(k, 0) for k in range(len(reads))])
        rb = eng.upload(b, prep=(np.full(len(reads), 60, np.uint8), 18, model, 45, 10))
        q, ins, dele, gcp = rb.read_table()
        rb.release()
        want = np.concatenate([np.array(c["expect"], np.uint8) for c in cases])
        assert np.array_equal(ins, want) and np.array_equal(dele, want), model
        assert (gcp == 10).all() and (q == 30).all()
    for c in fx["modify_read_qualities"]:
        qual = np.array(c["qual"], np.uint8)
        n = len(qual)
        s = np.frombuffer(b"ACGTACGTAC"[:n].ljust(n, b"A"), np.uint8)
        z = np.zeros(n, np.uint8)
        b = Batch.from_lists([(s, qual, z, z, z)], [hap], [(0, 0)])
        rb = eng.upload(b, prep=(np.array([c["mapq"]], np.uint8), c["threshold"], 0, 45, 10))
        q, _, _, _ = rb.read_table()
        rb.release()
        assert q.tolist() == c["expect"], c["why"]
    eng.close()


def _oracle_region(r, min_q=18, model=3):
    n_reads, n_haps = len(r[3]), len(r[7])
    w = np.full(n_haps * n_reads, np.nan)
    ws = np.zeros(n_reads, np.uint32)
    k = oracle.lib().phmm_oracle_call_region(r[0], r[1], r[2], r[3], r[4], n_reads, r[5], r[6], r[7], n_haps, min_q, model, w, ws)
    return w.reshape(n_haps, n_reads), ws[:k].copy()


def test_region_larger_than_a_stage_takes_the_direct_path():
    # the batcher's stages hold 64 Ki pairs / 4 096 reads / 1 MiB of sequence; a deeper region is evaluated on the worker's own
    # context (no batching) -- the same numbers, and it must not disturb the regions that ARE batched around it
    big = _raw_region(201, reads=4500, haps=3, read_len=(40, 120), hap_len=200, bad_reads=7)      # more reads than a stage holds
    wide = _raw_region(202, reads=700, haps=100, read_len=(30, 60), hap_len=90)                    # 70 000 pairs
    small = _raw_region(203, reads=20, haps=5, read_len=(50, 150), hap_len=260)
    r0, _ = hostapi.batching_stats()
    for r in (small, big, small, wide, small):
        got, surv = hostapi.call_region(*r)
        want, wsurv = _oracle_region(r)
        assert np.array_equal(surv, wsurv) and np.array_equal(got[:, :len(surv)], want[:, :len(surv)], equal_nan=True)
    r1, _ = hostapi.batching_stats()
    assert r1 - r0 == 3  # only the three small ones went through a stage


def test_regions_with_different_parameters_never_share_a_submission():
    # min_quality_threshold and the PCR model are per call: regions that differ in them are kept in separate stages
    import threading
    regions = [_raw_region(210 + i, reads=16, haps=4, read_len=(60, 150), hap_len=240, bad_reads=1) for i in range(12)]
    params = [(18, "CONSERVATIVE"), (10, "NONE"), (18, "HOSTILE"), (25, "AGGRESSIVE")]
    want = [_oracle_region(r, params[i % 4][0], hostapi.PCR[params[i % 4][1]]) for i, r in enumerate(regions)]
    errors = []

    def work(i):
        for _ in range(6):
            got, surv = hostapi.call_region(*regions[i], min_quality_threshold=params[i % 4][0], pcr_model=params[i % 4][1])
            if not (np.array_equal(surv, want[i][1]) and np.array_equal(got[:, :len(surv)], want[i][0][:, :len(surv)], equal_nan=True)):
                errors.append(i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(regions))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors


def test_long_reads_through_the_batcher():
    # reads of several thousand bases: striped kernels, fp64 re-run, and -- beyond 4 096 bases -- the PCR model's repeat search
    # working from global memory instead of the set-up launch's LDS copy
    r = _raw_region(220, reads=6, haps=3, read_len=(3000, 5200), hap_len=5600)
    got, surv = hostapi.call_region(*r)
    want, wsurv = _oracle_region(r)
    assert np.array_equal(surv, wsurv)
    k = len(surv)  # (reads this long with 1 % errors fall under the filter's -8: usually none survives; the filter must agree)
    fin = np.isfinite(want[:, :k])
    assert np.array_equal(np.isfinite(got[:, :k]), fin)
    if fin.any():
        assert np.abs(got[:, :k][fin] - want[:, :k][fin]).max() < 1e-9
    assert np.all(np.isnan(got[:, k:]))
