"""Print per-pair mismatches of the HIP path vs the oracle on a few batches (debug aid; run on the GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from rovaca_amd import capi, synth
from rovaca_amd.batch import Batch

eng = capi.Engine(0)
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pairhmm_golden.npz")))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import golden_batch
for tag in ("edge", "c1", "c3", "noisy"):
    b = golden_batch(g, tag)
    raw, l10, st = eng.compute(b)
    oraw, ol10, oresc = oracle.batch(b, n_threads=8)
    bad = np.nonzero(raw.view(np.uint32) != oraw.view(np.uint32))[0]
    print(f"== {tag}: {b.n_pairs} pairs, {bad.size} fp32 mismatches, rescued gpu {int(st.n_rescued)} oracle {int(oresc.sum())}")
    for p in bad[:40]:
        R, H = int(b.read_len[b.pair_read[p]]), int(b.hap_len[b.pair_hap[p]])
        print(f"   pair {p}: R={R} H={H} K={(R+15)//16} pad={16*((R+15)//16)-R} gpu={raw[p]:.9g} oracle={oraw[p]:.9g} rel={(raw[p]-oraw[p])/max(oraw[p],1e-45):.3g}")
    d = np.abs(l10 - ol10)
    d = d[np.isfinite(d)]
    print("   max |dlog10| =", d.max() if d.size else None, " nonfinite:", int((~np.isfinite(l10)).sum()))
