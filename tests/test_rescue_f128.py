"""The fp64 rescue pinned independently.  The reference's fp64 kernels are defective, so until round 3 the rescue
path was only ever compared with this repository's own double restatement.  tests/golden/rescue_f128.npz holds 569 pairs whose
fp32 sum underflows, with log10 likelihoods from a __float128 statement of the recurrence that shares no code with either
(oracle/phmm_oracle_f128.c; generator oracle/gen_golden_f128.py).  Tolerance, written here: |log10 difference| <= 1e-9 wherever the
scaled sum is a normal double; -inf exactly where the likelihood x 2^1020 lies below the smallest denormal."""
import os

import numpy as np
import pytest

import oracle
from rovaca_amd.batch import Batch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rescue_f128.npz")
TOL = 1e-9


@pytest.fixture(scope="module")
def fixture():
    z = np.load(GOLD)
    b = Batch(z["read_bases"], z["read_qual"], z["read_ins"], z["read_del"], z["read_gcp"], z["read_off"], z["read_len"], z["hap_bases"],
              z["hap_off"], z["hap_len"], z["pair_read"], z["pair_hap"])
    return b, z["log10_f128"], z["below_denormal"].astype(bool), z["group"]


def _check(l10, rescued, want, below, group):
    assert rescued.all(), "every pair of the fixture underflows fp32"
    assert len(want) >= 500 and set(np.unique(group)) == {0, 1, 2, 3, 4}
    # where even the scaled double has nothing left the reference's formula gives log10(0) - 307.05 = -inf
    assert below.sum() >= 3 and np.all(np.isneginf(l10[below])) and np.all(np.isneginf(want[below]) | (want[below] < -614.0))
    ok = ~below
    assert np.all(np.isfinite(l10[ok]))
    # the scaled sum of every other pair is a normal double (above 2^-1022 -> log10 above -614.7): full double precision applies
    assert want[ok].min() > -614.0
    err = np.abs(l10[ok] - want[ok])
    assert err.max() <= TOL, f"max |delta log10| = {err.max():.3e} at pair {np.nonzero(ok)[0][err.argmax()]}"
    return float(err.max())


def test_double_restatement_against_float128(fixture):
    b, want, below, group = fixture
    _, l10, resc = oracle.batch(b, n_threads=8)
    worst = _check(l10, resc, want, below, group)
    assert worst < 1e-10  # what is actually achieved (DESIGN.md section 2 quotes it); the contract above is 1e-9


def test_float128_statement_agrees_with_the_reference_fp32_kernels_where_they_do_not_underflow():
    # the other direction: on accepted pairs of the reference-made goldens the quad values must match the reference's own fp32
    # results to fp32 accuracy -- the independent statement computes the same quantity the reference does
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "pairhmm_golden.npz"))
    tabs = {"ph2pr_f64": z["ph2pr_f64"], "mm_f64": z["mm_f64"]}
    for tag in ("edge", "c3"):
        g = {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")}
        b = Batch(g["read_bases"], g["read_qual"], g["read_ins"], g["read_del"], g["read_gcp"], g["read_off"], g["read_len"],
                  g["hap_bases"], g["hap_off"], g["hap_len"], g["pair_read"], g["pair_hap"])
        acc = np.nonzero(g["raw_avx512_s"] >= np.float32(1e-28))[0][:40]
        for p in acc:
            v, below = oracle.pair_f128_log10(b, int(p), tabs)
            assert not below and abs(v - g["log10_ref"][p]) < 1e-5, (tag, p, v, g["log10_ref"][p])


@pytest.mark.gpu
def test_device_rescue_against_float128(fixture):
    from rovaca_amd import capi
    b, want, below, group = fixture
    eng = capi.Engine(0)
    raw, l10, st = eng.compute(b)
    assert st.n_rescued == b.n_pairs
    _check(l10, raw < np.float32(1e-28), want, below, group)
    # and pair by pair through every rescue kernel shape: the long pairs alone take the band kernel, the short ones the team kernel
    for sel in (group == 3, group == 1, group == 4):
        idx = np.nonzero(sel)[0]
        _, l2, _ = eng.compute(b.subset(idx))
        ok = ~below[idx]
        assert np.abs(l2[ok] - want[idx][ok]).max() <= TOL and np.all(np.isneginf(l2[~ok]))
    eng.close()
