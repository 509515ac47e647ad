"""How long ONE long pair takes, and what the fp64 kernel's residency does to a batch of them -- tuning aid (GPU box).  Pairs of one shape,
25 ... 1 000 of them: the fp32 band pass and the fp64 re-run ("Config 5's fp64 pass")."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from rovaca_amd import capi, synth
eng = capi.Engine(0)
for n_regions in (1, 4, 16, 20, 21, 40):  # x 25 pairs each: 25, 100, 400, 500, 525, 1000 pairs
    for sh in ((10000, 12000), (2000, 12000), (1000, 12000), (500, 12000), (6000, 7000)):
        b = synth.make_batch(seed=5, n_regions=n_regions, reads=5, haps=5, read_len=sh[0], hap_len=sh[1])
        rb = eng.upload(b)
        for _ in range(2): rb.sweep()
        ms, ms32 = rb.sweep_timed(5)
        print(f"{b.n_pairs:5d} pairs of {sh[0]:5d} x {sh[1]:5d}: pass {ms:7.3f} ms = fp32 {ms32:6.3f} + fp64 {ms - ms32:6.3f}", flush=True)
        rb.release()
