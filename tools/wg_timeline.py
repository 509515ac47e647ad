"""When, where and at what shader clock the workgroups of the table sweep ran (tuning aid, GPU box).
Needs a library built with the timeline instrumentation:  tools/build_tab_variant.sh wgtl -DPHMM_WG_TIMELINE -DPHMM_TAB_ONE_BUILD
usage: ROVACA_AMD_HIP_LIB=$PWD/rovaca_amd/lib/variants/lib_wgtl.so python tools/wg_timeline.py resident c2 [scale]
                                                                    python tools/wg_timeline.py oneshot c2
resident: the last of a row of resident sweeps -- span, workgroups in flight over time, duration and clock by start order.
oneshot:  the chunk launches of the last one-shot call (phmm_compute), one line each."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rovaca_amd import capi, synth

RING = 1 << 18


def read_ring(n_last):
    L = capi.lib()
    L.phmm_debug_wg_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    buf = np.zeros(RING * 4, np.uint64)
    cur = C.c_uint(0)
    rc = L.phmm_debug_wg_timeline(buf.ctypes.data_as(C.c_void_p), C.byref(cur))
    assert rc == 0, rc
    w = buf.reshape(-1, 4)
    n_last = min(n_last, RING, cur.value)
    idx = (cur.value - n_last + np.arange(n_last)) % RING
    return w[idx]


def describe(w, label):
    t0 = w[:, 0].astype(np.int64)
    t1 = w[:, 1].astype(np.int64)
    base = t0.min()
    st, en = (t0 - base) / 100.0, (t1 - base) / 100.0  # us (100 MHz clock)
    span, dur = en.max(), (t1 - t0) / 100.0
    ghz = w[:, 2].astype(np.float64) / np.maximum(dur, 0.01) / 1e3
    print(f"{label}: {len(w)} workgroups, span on device {span:.1f} us, {len(w) / span:.2f} workgroups/us")
    print(f"  workgroup duration: median {np.median(dur):.1f} us, p10 {np.percentile(dur, 10):.1f}, p90 {np.percentile(dur, 90):.1f}, max {dur.max():.1f}")
    mid = (np.linspace(0, span, 21)[:-1] + np.linspace(0, span, 21)[1:]) / 2
    print("  in flight at 5 %-steps of the span:", " ".join(str(int(((st <= m) & (en > m)).sum())) for m in mid))
    print(f"  average in flight {dur.sum() / span:.0f} of {256 * 4} slots")
    parts = np.array_split(np.argsort(st, kind="stable"), 10)
    print(f"  shader clock over a workgroup's life: median {np.median(ghz):.3f} GHz; by start decile:", " ".join(f"{np.median(ghz[p]):.2f}" for p in parts))
    print("  median duration by start decile:", " ".join(f"{np.median(dur[p]):.0f}" for p in parts))
    print("  median cycles (thousands) by start decile:", " ".join(f"{np.median(w[p, 2].astype(np.float64)) / 1e3:.0f}" for p in parts))


mode = sys.argv[1] if len(sys.argv) > 1 else "resident"
name = sys.argv[2] if len(sys.argv) > 2 else "c2"
eng = capi.Engine(0)
if mode == "resident":
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    b = synth.make_config(name, scale=scale)
    rb = eng.upload(b)
    for _ in range(3):
        rb.sweep()
    ms, ms32 = rb.sweep_timed(4)
    n_wg = (b.n_pairs + 15) // 16
    w = read_ring(RING)
    grid = int(w[-1, 3] >> 40)
    describe(w[-grid:][w[-grid:, 1] > 0], f"{name} x{scale} resident, {b.n_pairs} pairs, events {ms32:.3f} ms per launch; last launch (grid {grid})")
    rb.release()
else:
    b = synth.make_config(name)
    out = (np.zeros(b.n_pairs, np.float32), np.zeros(b.n_pairs, np.float64))
    import time
    for _ in range(4):
        t0 = time.perf_counter()
        eng.compute(b, out=out)
        wall = time.perf_counter() - t0
    w = read_ring((b.n_pairs + 15) // 16 + 4096)
    w = w[w[:, 1] > 0]
    # the last call's launches: everything that started after the biggest gap... simply: within `wall` + 1 ms of the newest end
    newest = int(w[:, 1].max())
    w = w[w[:, 0].astype(np.int64) > newest - int((wall + 0.0005) * 1e8)]
    base = int(w[:, 0].min())
    key = w[:, 3] >> 8  # grid and pair list
    print(f"{name} one-shot: wall {wall * 1e3:.3f} ms -> {b.cells() / wall / 1e9:.0f} GCUPS; {len(w)} workgroups of the last call, "
          f"device span {(newest - base) / 100.0:.1f} us")
    launches = []
    for k in np.unique(key):
        m = np.where(key == k)[0]
        # one pair list (a slot) serves several chunks one after the other: split where starts are far apart
        o = m[np.argsort(w[m, 0])]
        cuts = np.where(np.diff(w[o, 0].astype(np.int64)) > 20000)[0] + 1  # > 200 us without a start
        for part in np.split(o, cuts):
            launches.append(part)
    launches.sort(key=lambda p: int(w[p, 0].min()))
    print("  start us    end us   grid  recorded  wg/us   med dur us  med GHz  med kcycles")
    for p in launches:
        t0 = w[p, 0].astype(np.int64)
        t1 = w[p, 1].astype(np.int64)
        dur = (t1 - t0) / 100.0
        ghz = w[p, 2].astype(np.float64) / np.maximum(dur, 0.01) / 1e3
        s, e = (t0.min() - base) / 100.0, (t1.max() - base) / 100.0
        print(f"  {s:8.1f}  {e:8.1f}  {int(w[p[0], 3] >> 40):5d}  {len(p):7d}  {len(p) / (e - s):6.2f}  {np.median(dur):9.1f}  {np.median(ghz):7.3f}  {np.median(w[p, 2].astype(np.float64)) / 1e3:9.0f}")
    t0 = w[:, 0].astype(np.int64)
    t1 = w[:, 1].astype(np.int64)
    st, en = (t0 - base) / 100.0, (t1 - base) / 100.0
    span = en.max()
    mid = (np.linspace(0, span, 41)[:-1] + np.linspace(0, span, 41)[1:]) / 2
    print("  in flight at 2.5 %-steps of the span:", " ".join(str(int(((st <= m) & (en > m)).sum())) for m in mid))
