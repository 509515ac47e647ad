"""Where config 5's pass goes -- tuning aid.  1 000 pairs of one read length against haplotypes of one length: the fp32 pass and the
fp64 re-run (every pair underflows fp32 from ~2 000 bases on) for short reads against long haplotypes (the top strip alone), long
reads (top strip + band walk), and config 5 itself."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

eng = capi.Engine(0)
shapes = [(2000, 12000), (2000, 4000), (6000, 12000), (10000, 12000), (6000, 7000), None]
for sh in shapes:
    if sh is None:
        b, name = synth.make_config("c5"), "config 5"
    else:
        b, name = synth.make_batch(seed=5, n_regions=40, reads=5, haps=5, read_len=sh[0], hap_len=sh[1]), f"{sh[0]} x {sh[1]}"
    rb = eng.upload(b)
    for _ in range(2):
        rb.sweep()
        eng.synchronize()  # (the order of passes for long pairs goes by what the last FINISHED pass saw)
    ms, ms32 = rb.sweep_timed(5)
    _, _, st = rb.download()
    print(f"{name:>14s}: {b.n_pairs} pairs, pass {ms:7.3f} ms = fp32 {ms32:6.3f} + fp64 {ms - ms32:6.3f} (rescued {int(st.n_rescued)})  "
          f"{b.cells() / ms / 1e6:6.0f} GCUPS", flush=True)
    rb.release()
