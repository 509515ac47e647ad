"""A bounded slice of the randomised parity fuzz (tests/fuzz_cases.py; open-ended form: tests/debug/fuzz_gpu.py) with fixed seeds, so that
every -m gpu run executes the tool that found the two real kernel bugs of rounds 1-2 (a team launch shaped for a long haplotype reading
the carry of a short one; a pair joining late beside a stripe that began mid-row, seed 417).  About twenty seconds in all: the short /
mid-length mixes with their big launches, the long-read-biased mix (striped and band kernels, zero-superstep shortcut; pair counts capped
because the CPU restatement needs a second per 1e8 cells), both C ABIs and the adapter, against the oracle."""
import numpy as np
import pytest

import fuzz_cases
from rovaca_amd import capi, hostapi, swapi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines():
    eng, al = capi.Engine(0), swapi.Aligner(0)
    yield eng, al
    al.close()
    eng.close()


@pytest.mark.parametrize("seed,iterations", [(417, 150), (1, 150), (21, 150), (41, 110)])
def test_random_batches_through_both_c_abis(engines, seed, iterations):
    eng, al = engines
    rng = np.random.default_rng(seed)
    n_ph = n_sw = 0
    for it in range(1, iterations + 1):
        n_ph += fuzz_cases.pairhmm_iteration(eng, rng, it, long_pair_cap=6)
        n_sw += fuzz_cases.smith_waterman_iteration(al, rng, it)
    assert n_ph > 1000 and n_sw > 300


@pytest.mark.parametrize("seed,iterations", [(31, 40), (417, 40)])
def test_random_long_read_batches(engines, seed, iterations):
    eng, _ = engines
    rng = np.random.default_rng(seed)
    for it in range(1, iterations + 1):
        fuzz_cases.pairhmm_iteration(eng, rng, it, long_bias=True, long_pair_cap=5)


@pytest.mark.parametrize("seed,iterations,order", [(77, 40, "1"), (417, 30, "1"), (78, 40, None)])
def test_random_long_read_batches_fp64_first(engines, seed, iterations, order, monkeypatch):
    # the same long-read mix with the fp64 kernel first on every striped range (what the library does by default once a range has been
    # underflowing): the fp32 kernel then runs only for the pairs that might not underflow -- random reads here, so both kinds occur.
    # order None: no hook at all, the library's default -- each batch goes by what the engine's earlier batches saw
    if order is None:
        monkeypatch.delenv("PHMM_F64_FIRST", raising=False)
    else:
        monkeypatch.setenv("PHMM_F64_FIRST", order)
    eng, _ = engines
    rng = np.random.default_rng(seed)
    for it in range(1, iterations + 1):
        fuzz_cases.pairhmm_iteration(eng, rng, it, long_bias=True, long_pair_cap=5, f64_first=True)


def test_random_regions_through_the_adapter():
    assert hostapi.init_pairhmm_ptr_gpu(0)
    rng = np.random.default_rng(1234)
    for _ in range(400):
        fuzz_cases.adapter_iteration(rng)
