import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["PHMM_ENABLE_TEST_HOOKS"] = "1"; os.environ["PHMM_F64_FIRST"] = "0"
os.environ["PHMM_FORCE_LANES"] = "16"; os.environ["PHMM_NO_UNIFORM_GOP"] = "1"
from rovaca_amd import capi, synth
from rovaca_amd.batch import Batch
eng = capi.Engine(0)
K = 16
parts = [synth.make_batch(seed=K, n_regions=6, reads=12, haps=5, read_len=16 * K - 3, hap_len=(40 + 10 * K, 60 + 20 * K)),
         synth.make_batch(seed=K + 50, n_regions=6, reads=12, haps=5, read_len=16 * (K - 1) - 2, hap_len=(40 + 10 * K, 60 + 20 * K))]
b = Batch.concat(parts)
os.environ["PHMM_NO_TAB"] = "1"
raw0, _, st0 = eng.compute(b)
del os.environ["PHMM_NO_TAB"]
def run(tag, **env):
    for k, v in env.items(): os.environ[k] = v
    raw, _, st = eng.compute(b)
    for k in env: del os.environ[k]
    bad = raw.view(np.uint32) != raw0.view(np.uint32)
    print(tag, "launches", int(st.n_launches), "bad", int(bad.sum()), flush=True)
    return raw, bad
raw, bad = run("default")
run("per-class", PHMM_NO_RAGGED="1", PHMM_NO_BIN_MERGE="1")
run("no-ragged (merge)", PHMM_NO_RAGGED="1")
run("no xcd", PHMM_NO_XCD_LOCAL_ORDER="1")
order, bins = capi.plan_bins(b)
pos = np.empty(b.n_pairs, int); pos[order] = np.arange(b.n_pairs)
idx = np.nonzero(bad)[0]
for k in idx[np.argsort(pos[idx])][:48]:
    r, h = int(b.pair_read[k]), int(b.pair_hap[k])
    print(" pos", int(pos[k]), "wg", int(pos[k]) // 16, "g", int(pos[k]) % 16, "read", r, "R", int(b.read_len[r]), "hap", h, "H", int(b.hap_len[h]), "d", int(b.hap_off[h]) % 4, "gpu", raw[k], "ref", raw0[k], "ratio", raw[k] / raw0[k] if raw0[k] else None)
# neighbourhood: positions 96..176 with flags
lo, hi = int(pos[idx].min()) - 8, int(pos[idx].max()) + 8
print("positions", lo, hi)
for p in range(max(lo, 0), min(hi, b.n_pairs)):
    k = order[p]; r, h = int(b.pair_read[k]), int(b.pair_hap[k])
    print("  p", p, "wg", p // 16, "g", p % 16, "read", r, "R", int(b.read_len[r]), "hap", h, "H", int(b.hap_len[h]), "BAD" if bad[k] else "")
