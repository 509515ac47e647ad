"""allele marginalisation (genotype/allele_likelihoods.hpp:702-732, marginal_likelihoods_direct) --
the product's host routine, the adapter-level function with the reference's types, and the device form behind the C ABI,
each against the oracle's restatement.  A maximum of doubles: the bar is bit equality."""
import numpy as np
import pytest

import oracle
from rovaca_amd import hostapi


def _oracle(old, old2new, n_new, ev):
    old = np.ascontiguousarray(old, np.float64)
    out = np.empty((n_new, ev), np.float64)
    oracle.lib().phmm_oracle_marginalize(old.reshape(-1), old.shape[0], old.shape[1], ev, np.ascontiguousarray(old2new, np.int32),
                                         n_new, out.reshape(-1))
    return out


def _cases():
    rng = np.random.default_rng(7)
    # hand-made: two haplotypes per allele, one unmapped haplotype, one allele nobody maps to, NaN tail beyond the evidence
    # count (the reference's rows are as wide as the original read count, allele_likelihoods.hpp:426-435)
    old = np.array([[-1.0, -5.0, -3.0, np.nan], [-2.0, -4.0, -3.0, np.nan], [-9.0, -0.5, np.nan, np.nan], [-0.1, -0.1, -0.1, np.nan]])
    yield old, [0, 0, 1, -1], 3, 3, np.array([[-1.0, -4.0, -3.0], [-9.0, -0.5, -np.inf], [-np.inf] * 3])
    # signed zeros: the comparison is strict, so the first of +0 / -0 stays
    yield np.array([[0.0, -0.0], [-0.0, 0.0]]), [0, 0], 1, 2, np.array([[0.0, -0.0]])
    # -inf everywhere and an empty evidence list
    yield np.full((3, 5), -np.inf), [1, 0, 1], 2, 5, np.full((2, 5), -np.inf)
    yield np.zeros((2, 4)), [0, 1], 2, 0, np.zeros((2, 0))
    for _ in range(40):
        n_old, n_new = int(rng.integers(1, 40)), int(rng.integers(1, 12))
        stride = int(rng.integers(1, 300))
        ev = int(rng.integers(0, stride + 1))
        old = -rng.random((n_old, stride)) * rng.choice([1.0, 10.0, 300.0])
        old[rng.random(old.shape) < 0.05] = np.nan
        old[rng.random(old.shape) < 0.05] = -np.inf
        m = rng.integers(-1, n_new, n_old)
        yield old, m, n_new, ev, None


def test_host_marginalisation_matches_the_restatement_and_known_answers():
    n = 0
    for old, m, n_new, ev, want in _cases():
        ref = _oracle(old, m, n_new, ev)
        if want is not None:
            assert np.array_equal(ref.view(np.uint64), want.view(np.uint64))
        for through_adapter in (False, True):
            got = hostapi.marginal_likelihoods_direct(old, m, n_new, ev, through_adapter=through_adapter)
            assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))
        n += 1
    assert n >= 40


@pytest.mark.gpu
def test_device_marginalisation_matches_the_restatement():
    from rovaca_amd import capi
    eng = capi.Engine(0)
    for old, m, n_new, ev, _ in _cases():
        got, _ = eng.marginalize(old, m, n_new, ev)
        assert np.array_equal(got.view(np.uint64), _oracle(old, m, n_new, ev).view(np.uint64))
    # a batch-scale matrix: 64 haplotypes x 200 000 reads onto 5 alleles
    rng = np.random.default_rng(8)
    old = -rng.random((64, 200000)) * 30.0
    m = rng.integers(-1, 5, 64)
    got, ms = eng.marginalize(old, m, 5)
    assert np.array_equal(got.view(np.uint64), _oracle(old, m, 5, 200000).view(np.uint64)) and ms > 0
    with pytest.raises(capi.PhmmError):
        eng.marginalize(old[:4, :8], [0, 1, 2, 7], 3)  # index outside the new allele set
    eng.close()
