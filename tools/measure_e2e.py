"""PCIe-inclusive rate of the one-shot C ABI call (phmm_compute: host buffers in, host buffers out) on BASELINE config 2.
Includes pair-list binning on the host, uploads, haplotype-plane construction, the sweep, downloads and the host log10.
This number goes into DESIGN.md; it is never bench.py's `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
b = synth.make_config(cfg)
eng = capi.Engine(0)
eng.compute(b.subset(range(1000)))  # warm the context
best = None
for _ in range(3):
    t0 = time.perf_counter()
    raw, l10, st = eng.compute(b)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
    print(f"{cfg}: {b.n_pairs} pairs, {b.cells()} cells: wall {dt*1e3:.1f} ms (stats.total_ms {st.total_ms:.1f}, kernel_ms {st.kernel_ms:.2f}) "
          f"-> {b.cells()/dt/1e9:.1f} GCUPS PCIe-inclusive, rescued {st.n_rescued}")
print(f"best {b.cells()/best/1e9:.1f} GCUPS")
