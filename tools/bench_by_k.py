"""Kernel throughput per rows-per-lane K (read length 16*K, haplotype 2*R) -- tuning aid, run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

eng = capi.Engine(0)
ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(2, 17))
for K in ks:
    R = 16 * K
    H = max(2 * R, 100)
    n_regions = max(8, int(4.5e10 / (40 * 25 * R * H)))
    b = synth.make_batch(seed=K, n_regions=n_regions, reads=40, haps=25, read_len=R, hap_len=H)
    rb = eng.upload(b)
    for _ in range(4):
        rb.sweep()
    ms, ms32 = rb.sweep_timed(10)
    print(f"K={K:2d} R={R:3d} H={H:3d} pairs={b.n_pairs:8d} fp32 {ms32:8.3f} ms  {b.cells()/ms32/1e6:8.1f} GCUPS", flush=True)
    rb.release()
