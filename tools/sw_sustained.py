"""Smith-Waterman under sustained load (tuning aid, GPU box): T host threads, a context each, call swhip_align on the benchmark's batch
N times; prints the wall-clock rate of all calls together and what the kernels took by their own HIP events.
A lone synchronous caller leaves the device idle between kernels (upload, binning, download: about as long as the kernel), and the
chip's clock follows the duty cycle ("the clock follows the launch pattern").
usage: python tools/sw_sustained.py [threads=3] [calls per thread=12] [regions=1000]"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rovaca_amd import swapi

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
regions = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
refs, seqs, pairs = bench.sw_workload(5, regions)
rb, ro, rl = swapi._tables(refs)
sb, so, sl = swapi._tables(seqs)
pr = np.ascontiguousarray([p[0] for p in pairs], np.uint32)
ps = np.ascontiguousarray([p[1] for p in pairs], np.uint32)
n = len(pairs)
ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
batch = swapi.SwBatch(len(refs), len(seqs), n, ptr(rb), ptr(ro), ptr(rl), ptr(sb), ptr(so), ptr(sl), ptr(pr), ptr(ps))
cap = int(rl[pr].astype(np.int64).sum() + sl[ps].astype(np.int64).sum() + 2 * n)
par = swapi.SwParams(*swapi.PARAMS_GENOTYPE)
L = swapi.lib()


class Worker:
    def __init__(self):
        self.al = swapi.Aligner(0)
        self.out = np.zeros(n, swapi.ALIGNMENT_DTYPE)
        self.cig = np.zeros(cap, np.uint32)
        self.kernel_ms, self.total_ms, self.cells = [], [], 0

    def call(self):
        used, st = C.c_uint64(), swapi.SwStats()
        rc = L.swhip_align(self.al._h, C.byref(par), C.byref(batch), ptr(self.out), ptr(self.cig), cap, C.byref(used), C.byref(st))
        assert rc == 0, L.swhip_last_error()
        self.kernel_ms.append(st.kernel_ms)
        self.total_ms.append(st.total_ms)
        self.cells = int(st.cells)

    def loop(self, k):
        for _ in range(k):
            self.call()


for threads in sorted({1, T}):
    ws = [Worker() for _ in range(threads)]
    for w in ws:
        w.loop(2)
        w.kernel_ms.clear()
        w.total_ms.clear()
    ths = [threading.Thread(target=w.loop, args=(N,)) for w in ws]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.perf_counter() - t0
    cells = ws[0].cells
    k = np.concatenate([w.kernel_ms for w in ws])
    c = np.concatenate([w.total_ms for w in ws])
    print(f"{threads} thread(s) x {N} calls of {n} alignments: wall {wall * 1e3 / (threads * N):.3f} ms per call -> "
          f"{cells * threads * N / wall / 1e9:.0f} GCUPS wall clock (host buffers in, results out); kernel by its events median {np.median(k):.3f} ms "
          f"({cells / np.median(k) / 1e6:.0f} GCUPS if alone on the device), a call median {np.median(c):.3f} ms", flush=True)
    # every context must have produced the same alignments (element offsets differ: they come from an atomic cursor)
    ref = ws[0]
    for w in ws[1:]:
        for f in ("score", "max_i", "max_j", "alignment_offset", "cigar_count"):
            assert (w.out[f] == ref.out[f]).all(), f"contexts disagree on {f}"
        for i in range(0, n, 97):
            a, b = ref.out[i], w.out[i]
            assert (ref.cig[a["cigar_off"]:a["cigar_off"] + a["cigar_count"]] == w.cig[b["cigar_off"]:b["cigar_off"] + b["cigar_count"]]).all()
    for w in ws:
        w.al.close()
