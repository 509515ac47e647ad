import os
import sys

import numpy as np
import pytest

# the PHMM_* / SWHIP_* test hooks the suite flips are dead without the master switch (rovaca_amd/csrc/pairhmm_binning.h)
os.environ.setdefault("PHMM_ENABLE_TEST_HOOKS", "1")
# Long pairs are swept fp32 first, as the reference does, throughout the suite -- so that the fp32 sum of EVERY pair can be compared
# with the oracle's.  The library's default is adaptive: a range of long pairs that has been underflowing runs fp64 first and reports
# the fp32 sums of the pairs it can prove underflowed as 0 (pairhmm_sweep.cpp, "fp64 first"); the tests of that order set the hook
# to 1 or remove it.
if not os.environ.get("ROVACA_AMD_TEST_LIBRARY_DEFAULT_ORDER"):  # (set it to run the suite under the adaptive default: only comparisons of
    os.environ.setdefault("PHMM_F64_FIRST", "0")               # the fp32 sums of long re-run pairs may then differ)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "pairhmm_golden.npz")
    return dict(np.load(path))


def golden_batch(g, tag):
    from rovaca_amd.batch import Batch
    kw = {k: g[f"{tag}_{k}"] for k in ("read_bases", "read_qual", "read_ins", "read_del", "read_gcp", "read_off",
                                        "read_len", "hap_bases", "hap_off", "hap_len", "pair_read", "pair_hap")}
    if f"{tag}_region_read" in g:
        kw.update(region_read=g[f"{tag}_region_read"], region_hap=g[f"{tag}_region_hap"],
                  region_pair=g[f"{tag}_region_pair"])
    return Batch(**kw)


GOLDEN_TAGS = ("edge", "c1", "c2", "c3", "noisy")
