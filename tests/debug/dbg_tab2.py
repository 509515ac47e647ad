import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["PHMM_ENABLE_TEST_HOOKS"] = "1"; os.environ["PHMM_F64_FIRST"] = "0"
os.environ["PHMM_FORCE_LANES"] = "16"
from rovaca_amd import capi, synth
from rovaca_amd.batch import Batch
eng = capi.Engine(0)
for mode in ("ugo", "ug"):
    if mode == "ug": os.environ["PHMM_NO_UNIFORM_GOP"] = "1"
    for K in range(2, 17):
        parts = [synth.make_batch(seed=K, n_regions=6, reads=12, haps=5, read_len=16 * K - 3, hap_len=(40 + 10 * K, 60 + 20 * K)),
                 synth.make_batch(seed=K + 50, n_regions=6, reads=12, haps=5, read_len=16 * (K - 1) - 2, hap_len=(40 + 10 * K, 60 + 20 * K))]
        b = Batch.concat(parts)
        raw, l10, st = eng.compute(b)
        os.environ["PHMM_NO_TAB"] = "1"
        raw0, _, st0 = eng.compute(b)
        del os.environ["PHMM_NO_TAB"]
        bad = np.nonzero(raw.view(np.uint32) != raw0.view(np.uint32))[0]
        rl = b.read_len[b.pair_read]
        print(mode, "K", K, "pairs", b.n_pairs, "launches", int(st.n_launches), "bad", bad.size, "bad by read len", sorted(set(int(x) for x in rl[bad])), "nan", int(np.isnan(raw[bad]).sum()) if bad.size else 0, flush=True)
