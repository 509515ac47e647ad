import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["PHMM_ENABLE_TEST_HOOKS"] = "1"; os.environ["PHMM_F64_FIRST"] = "0"
import fuzz_cases, oracle
from rovaca_amd import capi, swapi
from rovaca_amd.batch import Batch
eng, al = capi.Engine(0), swapi.Aligner(0)
rng = np.random.default_rng(21)
# monkeypatch Batch.from_lists to capture the batch
captured = {}
orig = Batch.from_lists
def cap(reads, haps, pairs=None):
    b = orig(reads, haps, pairs); captured["b"] = b; captured["pairs"] = pairs; return b
Batch.from_lists = staticmethod(cap)
for it in range(1, 151):
    try:
        fuzz_cases.pairhmm_iteration(eng, rng, it, long_pair_cap=6)
    except AssertionError as e:
        print("FAIL it", it, str(e)[:200])
        b = captured["b"]
        raw, l10, st = eng.compute(b)
        oraw, ol10, oresc = oracle.batch(b, n_threads=8)
        bad = np.nonzero(raw.view(np.uint32) != oraw.view(np.uint32))[0]
        print("n_pairs", b.n_pairs, "bad", bad.size)
        hasN = np.array([np.any(b.hap(int(h)) == ord("N")) for h in range(b.n_haps)])
        for k in bad[:12]:
            r, h = int(b.pair_read[k]), int(b.pair_hap[k])
            print(" pair", k, "read", r, "R", int(b.read_len[r]), "hap", h, "H", int(b.hap_len[h]), "hapN", bool(hasN[h]), "gpu", raw[k], "oracle", oraw[k])
        rs = set(int(b.pair_read[k]) for k in bad)
        print("bad reads", rs, "all pairs of those reads bad?", [(r, int((b.pair_read == r).sum()), int((b.pair_read[bad] == r).sum())) for r in rs])
        os.environ["PHMM_NO_TAB"] = "1"
        raw2, _, _ = eng.compute(b)
        print("NO_TAB bad", int((raw2.view(np.uint32) != oraw.view(np.uint32)).sum()))
        del os.environ["PHMM_NO_TAB"]
        for s in (1, 2, 16):
            os.environ["PHMM_TAB_SLOTS"] = str(s)
            raw3, _, _ = eng.compute(b)
            print("TAB_SLOTS", s, "bad", int((raw3.view(np.uint32) != oraw.view(np.uint32)).sum()))
        del os.environ["PHMM_TAB_SLOTS"]
        break
    fuzz_cases.smith_waterman_iteration(al, rng, it)
