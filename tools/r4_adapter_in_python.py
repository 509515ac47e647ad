"""The adapter leg of bench.py on its own, repeated, with and without the one-shot pipeline having run in the process first (its helper
threads and extra streams stay around) -- tuning aid for the run-to-run spread of bench.py's adapter numbers."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np

import bench
from rovaca_amd import capi, synth

args = argparse.Namespace(adapter_workers=0)
if "torch" in sys.argv:  # (as bench.py does: before the engine)
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    print("torch has initialised its HIP context")
eng = capi.Engine(0)
if "resident" in sys.argv:
    big = synth.make_config("c2")
    rbig = eng.upload(big)
    for _ in range(5):
        rbig.sweep()
    eng.synchronize()
    print("a 1M-pair batch is resident and has been swept")
if "e2e" in sys.argv:
    b = synth.make_config("c2", scale=0.3)
    for _ in range(3):
        eng.compute(b)
    print("one-shot pipeline has run (helper pools and extra streams exist)")
for rep in range(3):
    t = time.time()
    a = bench.adapter_leg(eng, args)
    m = a["mixed_shapes"]
    print(f"rep {rep}: deep {a['value']} sync {a['synchronous_seam']['value']} (rps {a['synchronous_seam']['regions_per_submission']}, "
          f"dev wait {a['synchronous_seam']['host_us_per_region']['dispatcher_device_wait']}) mixed deep {m['value']} mixed sync "
          f"{m['synchronous_seam']['value']}  [{time.time() - t:.1f} s]", flush=True)
