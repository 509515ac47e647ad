import os, sys
os.environ["PHMM_ENABLE_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rovaca_amd import capi, synth
eng = capi.Engine(0)
for n_regions in (41, 75):
    b = synth.make_batch(seed=700 + n_regions, n_regions=n_regions, reads=40, haps=25, read_len=(60, 200), hap_len=(150, 400), sub_rate=0.06 if n_regions == 75 else 0.01, qual=30 if n_regions == 75 else "illumina")
    raw, l10, st = eng.compute(b)
    raw2, l10_2, _ = eng.compute(b)
    print("chunked ok", n_regions, flush=True)
    os.environ["PHMM_NO_CHUNKED_COMPUTE"] = "1"
    praw, pl10, pst = eng.compute(b)
    del os.environ["PHMM_NO_CHUNKED_COMPUTE"]
    print("plain ok", n_regions, np.array_equal(raw.view(np.uint32), praw.view(np.uint32)), flush=True)
