"""Compare the team kernel against the oracle on a few long-read pairs, per forced shape (debug aid; run on the GPU box)."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) == 1:
    for lanes in ("16", "64"):
        for waves in ("1", "4"):
            env = dict(os.environ, PHMM_ENABLE_TEST_HOOKS="1", PHMM_FORCE_TEAM_LANES=lanes, PHMM_FORCE_TEAM_WAVES=waves)
            subprocess.run([sys.executable, __file__, "child"], env=env)
    sys.exit(0)
import oracle
from rovaca_amd import capi
from rovaca_amd.batch import Batch
rng = np.random.default_rng(8)
acgt = np.frombuffer(b"ACGT", np.uint8)
reads, haps, pairs = [], [], []
for R, H in [(257, 300), (300, 700), (513, 600), (600, 1500), (1025, 1100), (1537, 2000), (2000, 2100), (129, 5000), (40, 4097), (64, 900), (700, 90), (1745, 314), (1745, 314), (900, 200), (2500, 100)]:
    hap = acgt[rng.integers(0, 4, H)]
    st = int(rng.integers(0, max(1, H - R + 1)))
    seq = np.resize(hap[st:st + R], R).copy()
    seq[rng.random(R) < 0.02] = ord("G")
    q = rng.choice(np.array([6, 25, 37], np.uint8), R, p=[0.05, 0.15, 0.8])
    g = np.full(R, 40, np.uint8)
    reads.append((seq, q, g, g, np.full(R, 10, np.uint8)))
    haps.append(hap)
    pairs.append((len(reads) - 1, len(haps) - 1))
b = Batch.from_lists(reads, haps, pairs)
eng = capi.Engine(0)
raw, l10, st = eng.compute(b)
oraw, ol10, oresc = oracle.batch(b, n_threads=4)
print("lanes", os.environ.get("PHMM_FORCE_TEAM_LANES"), "waves", os.environ.get("PHMM_FORCE_TEAM_WAVES"))
for p in range(b.n_pairs):
    R, H = int(b.read_len[b.pair_read[p]]), int(b.hap_len[b.pair_hap[p]])
    ok32 = raw[p].view(np.uint32) == oraw[p].view(np.uint32)
    ok = (l10[p] == ol10[p]) or (np.isnan(l10[p]) and np.isnan(ol10[p]))
    print(f"  R={R:5d} H={H:5d} fp32 {'ok ' if ok32 else 'BAD'} gpu={raw[p]:.6g} ref={oraw[p]:.6g}   final {'ok ' if ok else 'BAD'} gpu={l10[p]:.9g} ref={ol10[p]:.9g} resc={bool(oresc[p])}")
