"""Host-side logic of the drop-in adapter (quality modification, PCR indel model, normalisation, read filter) against
the oracle restatement and hand-derived known answers.  CPU only."""
import ctypes
import os

import numpy as np
import pytest

import oracle
from rovaca_amd import hostapi


def test_host_library_exports():
    L = ctypes.CDLL(hostapi.LIB_PATH)
    for n in hostapi.EXPORTS:
        assert hasattr(L, n)


def test_modify_read_qualities_matches_oracle():
    rng = np.random.default_rng(1)
    H, O = hostapi.lib(), oracle.lib()
    for mq in (0, 5, 17, 18, 30, 60, 255):
        for thr in (6, 18, 30):
            q = rng.integers(0, 94, 300).astype(np.uint8)
            a, b = q.copy(), q.copy()
            H.rovaca_amd_modify_read_qualities(a, len(a), mq, thr)
            O.phmm_oracle_modify_quals(b, len(b), mq, thr)
            assert np.array_equal(a, b)
    q = np.array([2, 11, 17, 18, 25, 37, 70], np.uint8)
    H.rovaca_amd_modify_read_qualities(q, len(q), 60, 18)
    assert q.tolist() == [6, 6, 6, 18, 25, 37, 60]  # min(q, MQ) then < 18 -> 6


def _reads(rng, n):
    out = []
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for _ in range(n):
        L = int(rng.integers(1, 260))
        kind = rng.integers(0, 4)
        if kind == 0:
            s = rng.choice(acgt, L)
        elif kind == 1:  # short tandem repeats
            unit = rng.choice(acgt, int(rng.integers(1, 9)))
            s = np.resize(unit, L).copy()
            s[rng.random(L) < 0.05] = ord("A")
        elif kind == 2:  # homopolymer runs
            s = np.repeat(rng.choice(acgt, L // 5 + 1), rng.integers(1, 12, L // 5 + 1))[:L]
            s = np.resize(s, L)
        else:
            s = np.concatenate([rng.choice(np.frombuffer(b"ACGTN", np.uint8), L // 2),
                                np.resize(np.frombuffer(b"CAG", np.uint8), L - L // 2)])
        out.append(np.ascontiguousarray(s, np.uint8))
    return out


def test_pcr_indel_model_matches_oracle():
    rng = np.random.default_rng(2)
    H, O = hostapi.lib(), oracle.lib()
    for s in _reads(rng, 300):
        n = len(s)
        for off in rng.integers(0, n, min(n, 12)):
            assert H.rovaca_amd_tandem_repeat_length(s, n, int(off)) == O.phmm_oracle_find_tandem_repeat_units(s, n, int(off))
        for model in (0, 1, 2, 3):
            a = np.full(n, 45, np.uint8)
            b = a.copy()
            H.rovaca_amd_apply_pcr_error_model(s, a, n, model)
            O.phmm_oracle_apply_pcr_error_model(s, b, n, model)
            assert np.array_equal(a, b)


def test_pcr_indel_model_known_answers():
    H = hostapi.lib()
    # homopolymer A x 12: at offset i the run is i+1 back + (11-i) forward = 12 repeats of "A"
    s = np.frombuffer(b"A" * 12, np.uint8).copy()
    assert [H.rovaca_amd_tandem_repeat_length(s, 12, i) for i in range(11)] == [12] * 11
    g = np.full(12, 45, np.uint8)
    H.rovaca_amd_apply_pcr_error_model(s, g, 12, 3)  # CONSERVATIVE[12] = 37 (adapter.h:15); last base keeps 45
    assert g.tolist() == [37] * 11 + [45]
    g = np.full(12, 45, np.uint8)
    H.rovaca_amd_apply_pcr_error_model(s, g, 12, 1)  # HOSTILE[12] = 10
    assert g.tolist() == [10] * 11 + [45]
    g = np.full(12, 45, np.uint8)
    H.rovaca_amd_apply_pcr_error_model(s, g, 12, 0)  # NONE
    assert g.tolist() == [45] * 12
    # no repeat of length > 2 anywhere: every capped entry is 40 under CONSERVATIVE (table entries 0..3 are 40)
    s = np.frombuffer(b"ACGTCAGTTGCA", np.uint8).copy()
    g = np.full(12, 45, np.uint8)
    H.rovaca_amd_apply_pcr_error_model(s, g, 12, 3)
    assert set(g[:-1].tolist()) <= {40} and g[-1] == 45


def test_normalize_and_filter_match_oracle_and_known_answers():
    rng = np.random.default_rng(3)
    H, O = hostapi.lib(), oracle.lib()
    for _ in range(50):
        nr, nh = int(rng.integers(1, 40)), int(rng.integers(1, 20))
        lik = -rng.random((nr, nh)) * rng.choice([1.0, 5.0, 20.0, 60.0], (nr, 1))
        rl = rng.integers(1, 300, nr)
        a, b = lik.copy().reshape(-1), lik.copy().reshape(-1)
        H.rovaca_amd_normalize_likelihoods(a, nr, nh)
        O.phmm_oracle_normalize(b, nr, nh)
        assert np.array_equal(a, b)
        assert np.all(a.reshape(nr, nh) >= a.reshape(nr, nh).max(axis=1, keepdims=True) - 4.5 - 1e-12)
        drop = np.zeros(nr, np.uint8)
        keep = np.zeros(nr, np.uint8)
        nd = H.rovaca_amd_poorly_modelled_reads(a, rl.astype(np.int32), nr, nh, drop)
        nk = O.phmm_oracle_filter(a, rl.astype(np.uint32), nr, nh, keep)
        assert nd + nk == nr and np.array_equal(drop, 1 - keep)
    # threshold = min(2, ceil(0.02 * R)) * -4: R = 40 -> -4, R = 150 -> -8 (adapter.h:20-22)
    lik = np.array([[-3.9], [-4.1], [-7.9], [-8.1]])
    drop = np.zeros(4, np.uint8)
    H.rovaca_amd_poorly_modelled_reads(lik.reshape(-1).copy(), np.array([40, 40, 150, 150], np.int32), 4, 1, drop)
    assert drop.tolist() == [0, 1, 0, 1]
    v = np.array([-1.0, -10.0, -5.4, np.nan])
    H.rovaca_amd_normalize_likelihoods(v, 1, 4)
    assert v[:3].tolist() == [-1.0, -5.5, -5.4] and np.isnan(v[3])


def test_adapter_refuses_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    assert hostapi.init_pairhmm_ptr_gpu() is False  # no CPU path is substituted; the seam pointer stays unset
    with pytest.raises(RuntimeError, match="not installed"):
        hostapi.call_region(np.frombuffer(b"ACGT", np.uint8), np.full(4, 30, np.uint8), [0], [4], [60],
                            np.frombuffer(b"ACGTA", np.uint8), [0], [5])


# ---- hand-derived known answers (tests/golden/host_logic_known_answers.json): the only pin H3/H4/H8 can have ----
def _fixture():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_logic_known_answers.json")))


def _f(v):
    return float(v) if isinstance(v, str) else v


def test_known_answers_tandem_repeat_length():
    H, O = hostapi.lib(), oracle.lib()
    n_cases = 0
    for case in _fixture()["tandem_repeat_length"]:
        s = np.frombuffer(case["bases"].encode(), np.uint8).copy()
        for off, want in zip(case["offsets"], case["expect"]):
            assert H.rovaca_amd_tandem_repeat_length(s, len(s), off) == want, (case["bases"], off, case["why"])
            assert O.phmm_oracle_find_tandem_repeat_units(s, len(s), off) == want, (case["bases"], off)
            n_cases += 1
    assert n_cases >= 60


def test_known_answers_pcr_error_model():
    H, O = hostapi.lib(), oracle.lib()
    for case in _fixture()["apply_pcr_error_model"]:
        s = np.frombuffer(case["bases"].encode(), np.uint8).copy()
        g0 = np.full(len(s), case["gop_in"], np.uint8) if isinstance(case["gop_in"], int) else np.array(case["gop_in"], np.uint8)
        a, b = g0.copy(), g0.copy()
        H.rovaca_amd_apply_pcr_error_model(s, a, len(s), case["model"])
        O.phmm_oracle_apply_pcr_error_model(s, b, len(s), case["model"])
        assert a.tolist() == case["expect"], (case["bases"], case["model"], case["why"])
        assert b.tolist() == case["expect"]


def test_known_answers_quality_modification():
    H, O = hostapi.lib(), oracle.lib()
    for case in _fixture()["modify_read_qualities"]:
        a = np.array(case["qual"], np.uint8)
        b = a.copy()
        H.rovaca_amd_modify_read_qualities(a, len(a), case["mapq"], case["threshold"])
        O.phmm_oracle_modify_quals(b, len(b), case["mapq"], case["threshold"])
        assert a.tolist() == case["expect"], case["why"]
        assert b.tolist() == case["expect"]


def test_known_answers_filter_and_normalise():
    H, O = hostapi.lib(), oracle.lib()
    fx = _fixture()
    for case in fx["poorly_modelled_reads"]:
        n = len(case["read_len"])
        # two haplotypes per read: the best one decides
        lik = np.array([[b, b - 3.0] for b in case["best"]], np.float64).reshape(-1)
        drop, keep = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        H.rovaca_amd_poorly_modelled_reads(lik.copy(), np.array(case["read_len"], np.int32), n, 2, drop)
        O.phmm_oracle_filter(lik.copy(), np.array(case["read_len"], np.uint32), n, 2, keep)
        assert drop.tolist() == case["drop"], case["why"]
        assert (1 - keep).tolist() == case["drop"]
    for case in fx["normalize_likelihoods"]:
        want = np.array([_f(v) for v in case["expect"]])
        for fn in (H.rovaca_amd_normalize_likelihoods, O.phmm_oracle_normalize):
            v = np.array([_f(x) for x in case["row"]])
            fn(v, 1, len(v))
            assert np.array_equal(v, want, equal_nan=True), (case["why"], v)
