"""Randomised parity fuzz on the GPU, open-ended (the bounded fixed-seed slice that pytest -m gpu runs is tests/test_gpu_fuzz.py): random
batches of random shapes through both C ABIs and the adapter against the CPU restatements (tests/fuzz_cases.py).
usage: python tests/debug/fuzz_gpu.py [seconds] [seed] [long]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PHMM_ENABLE_TEST_HOOKS", "1")
import fuzz_cases  # noqa: E402
from rovaca_amd import capi, hostapi, swapi  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
long_bias = len(sys.argv) > 3 and sys.argv[3] == "long"  # mostly long reads: team kernels, zero-superstep shortcut
rng = np.random.default_rng(seed)
eng, al = capi.Engine(0), swapi.Aligner(0)
t_end = time.time() + budget
n_ph = n_sw = it = 0
try:
    while time.time() < t_end:
        it += 1
        # long pairs: the reference's order of passes, fp64 first, and the library's default (which goes by what it has seen: the fp32
        # sum of a pair it proves underflowed may then be reported as 0, as under fp64 first) in turn
        mode = it % 3
        if mode == 2:
            os.environ.pop("PHMM_F64_FIRST", None)
        else:
            os.environ["PHMM_F64_FIRST"] = str(mode)
        n_ph += fuzz_cases.pairhmm_iteration(eng, rng, it, long_bias, long_pair_cap=8 if long_bias else None, f64_first=mode != 0)
        n_sw += fuzz_cases.smith_waterman_iteration(al, rng, it)
        if it % 20 == 0:
            print(f"iter {it}: {n_ph} pairhmm pairs, {n_sw} alignments ok", flush=True)
    print(f"fuzz ok: {it} iterations, {n_ph} PairHMM pairs, {n_sw} Smith-Waterman alignments, seed {seed}")
    assert hostapi.init_pairhmm_ptr_gpu(0)
    t_end = time.time() + budget / 4
    n_reg = 0
    while time.time() < t_end:
        n_reg += 1
        fuzz_cases.adapter_iteration(rng)
    print(f"fuzz ok: {n_reg} regions through the adapter")
except AssertionError as e:
    print("MISMATCH (seed %d): %s" % (seed, e))
    sys.exit(1)
