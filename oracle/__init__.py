"""ctypes bindings for the parity checker.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
package.  ``oracle.lib()`` is the C restatement (oracle/phmm_oracle.c); ``oracle.ref()`` is the
reference's own AVX kernels built from /root/reference into oracle/_ref/ (None when absent or when
the host CPU lacks AVX-512).  ``oracle.f128_lib()`` is the independent __float128 statement of the
recurrence (oracle/phmm_oracle_f128.c) that pins the fp64 rescue, whose reference kernels are defective.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None
_REF_TRIED = False

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def under_profiler():
    """rocprofv3 / rocprof preload a tool library that initialises the GPU before main(): such a process must not fork + exec."""
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ)


def build():
    """(Re)build oracle/_build and, where /root/reference exists, oracle/_ref.  Refused under a profiler: `make` and the compilers
    would be exec'd from a process whose GPU the profiler's preloaded library has initialised (forbidden on the GPU pool); build
    beforehand (`python -c 'import __graft_entry__ as g; g.build()'`)."""
    if under_profiler():
        raise RuntimeError("oracle: libraries are missing and a profiler is attached -- build them first (make -C oracle), "
                           "this process must not start child processes")
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def cpu_has_avx512():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return all(x in fl for x in ("avx512f", "avx512dq", "avx512vl", "avx2", "fma"))
    except OSError:
        pass
    return False


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "_build", "libphmm_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    pair_args = [u8p, u8p, u8p, u8p, u8p, C.c_uint32, u8p, C.c_uint32]
    L.phmm_oracle_pair_f32.argtypes = pair_args
    L.phmm_oracle_pair_f32.restype = C.c_float
    L.phmm_oracle_pair_f64.argtypes = pair_args
    L.phmm_oracle_pair_f64.restype = C.c_double
    L.phmm_oracle_finalize_f32.argtypes = [C.c_float]
    L.phmm_oracle_finalize_f32.restype = C.c_double
    L.phmm_oracle_finalize_f64.argtypes = [C.c_double]
    L.phmm_oracle_finalize_f64.restype = C.c_double
    for n, t in (("phmm_oracle_ph2pr_f32", C.c_float), ("phmm_oracle_ph2pr_f64", C.c_double),
                 ("phmm_oracle_mm_table_f32", C.c_float), ("phmm_oracle_mm_table_f64", C.c_double)):
        getattr(L, n).restype = C.POINTER(t)
    L.phmm_oracle_mm_prob_f32.argtypes = [C.c_int, C.c_int]
    L.phmm_oracle_mm_prob_f32.restype = C.c_float
    L.phmm_oracle_mm_prob_f64.argtypes = [C.c_int, C.c_int]
    L.phmm_oracle_mm_prob_f64.restype = C.c_double
    L.phmm_oracle_batch.argtypes = [u8p, u8p, u8p, u8p, u8p, u32p, u32p, u8p, u32p, u32p, u32p, u32p, C.c_uint64,
                                    f32p, f64p, u8p, C.c_int]
    L.phmm_oracle_batch.restype = None
    L.phmm_oracle_modify_quals.argtypes = [u8p, C.c_uint32, C.c_uint8, C.c_int32]
    L.phmm_oracle_modify_quals.restype = None
    L.phmm_oracle_find_tandem_repeat_units.argtypes = [u8p, C.c_uint32, C.c_uint32]
    L.phmm_oracle_find_tandem_repeat_units.restype = C.c_int32
    L.phmm_oracle_apply_pcr_error_model.argtypes = [u8p, u8p, C.c_uint32, C.c_int]
    L.phmm_oracle_apply_pcr_error_model.restype = None
    L.phmm_oracle_normalize.argtypes = [f64p, C.c_uint32, C.c_uint32]
    L.phmm_oracle_normalize.restype = None
    L.phmm_oracle_filter.argtypes = [f64p, u32p, C.c_uint32, C.c_uint32, u8p]
    L.phmm_oracle_filter.restype = C.c_uint32
    L.phmm_oracle_marginalize.argtypes = [f64p, C.c_uint32, C.c_uint32, C.c_uint32, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"),
                                          C.c_uint32, f64p]
    L.phmm_oracle_marginalize.restype = None
    L.phmm_oracle_call_region.argtypes = [u8p, u8p, u32p, u32p, u8p, C.c_uint32, u8p, u32p, u32p, C.c_uint32,
                                          C.c_int32, C.c_int, f64p, u32p]
    L.phmm_oracle_call_region.restype = C.c_uint32
    _LIB = L
    return L


def ref():
    """The reference's own kernels (oracle/_ref/libphmm_ref.so) or None."""
    global _REF, _REF_TRIED
    if _REF_TRIED:
        return _REF
    _REF_TRIED = True
    path = os.path.join(_HERE, "_ref", "libphmm_ref.so")
    if not os.path.exists(path) or not cpu_has_avx512():
        return None
    L = C.CDLL(path)
    pair_args = [u8p, u8p, u8p, u8p, u8p, C.c_uint32, u8p, C.c_uint32]
    for n, t in (("ref_pair_avx512_s", C.c_float), ("ref_pair_avx2_s", C.c_float),
                 ("ref_pair_avx512_d", C.c_double), ("ref_pair_avx2_d", C.c_double)):
        getattr(L, n).argtypes = pair_args
        getattr(L, n).restype = t
    L.ref_init.restype = None
    L.ref_group_16x1.argtypes = [u8p, u8p, u8p, u8p, u8p, u32p, u32p, u8p, C.c_uint32, f32p]
    L.ref_group_16x1.restype = None
    L.ref_tables.argtypes = [f32p, f32p, f64p, f64p]
    L.ref_tables.restype = None
    L.ref_regions_16x1.argtypes = [u8p, u8p, u8p, u8p, u8p, u32p, u32p, u8p, u32p, u32p, u32p, u32p, u64p, C.c_uint32,
                                   f32p, C.c_int]
    L.ref_regions_16x1.restype = C.c_double
    L.ref_pairs_avx512_s.argtypes = [u8p, u8p, u8p, u8p, u8p, u32p, u32p, u8p, u32p, u32p, u32p, u32p, C.c_uint64,
                                     f32p, C.c_int]
    L.ref_pairs_avx512_s.restype = C.c_double
    L.ref_init()
    _REF = L
    return L


# ------------------------------------------------------------------------------------------------
# numpy-level helpers
# ------------------------------------------------------------------------------------------------
MM_TABLE_SIZE = ((254 + 1) * (254 + 2)) >> 1


def tables():
    L = lib()
    return dict(
        ph2pr_f32=np.ctypeslib.as_array(L.phmm_oracle_ph2pr_f32(), (128,)).copy(),
        ph2pr_f64=np.ctypeslib.as_array(L.phmm_oracle_ph2pr_f64(), (128,)).copy(),
        mm_f32=np.ctypeslib.as_array(L.phmm_oracle_mm_table_f32(), (MM_TABLE_SIZE,)).copy(),
        mm_f64=np.ctypeslib.as_array(L.phmm_oracle_mm_table_f64(), (MM_TABLE_SIZE,)).copy(),
    )


def ref_tables():
    R = ref()
    a = np.zeros(128, np.float32)
    b = np.zeros(MM_TABLE_SIZE, np.float32)
    c = np.zeros(128, np.float64)
    d = np.zeros(MM_TABLE_SIZE, np.float64)
    R.ref_tables(a, b, c, d)
    return dict(ph2pr_f32=a, mm_f32=b, ph2pr_f64=c, mm_f64=d)


def batch(b, n_threads=1, want_raw=True):
    """Run the C restatement over a flat batch (rovaca_amd.batch.Batch).  Returns (raw_f32, log10, rescued)."""
    L = lib()
    n = int(b.n_pairs)
    raw = np.zeros(n, np.float32)
    l10 = np.zeros(n, np.float64)
    resc = np.zeros(n, np.uint8)
    L.phmm_oracle_batch(b.read_bases, b.read_qual, b.read_ins, b.read_del, b.read_gcp, b.read_off, b.read_len,
                        b.hap_bases, b.hap_off, b.hap_len, b.pair_read, b.pair_hap, n, raw, l10, resc, n_threads)
    return raw, l10, resc.astype(bool)


_F128 = None


def f128_lib():
    """The __float128 statement of the recurrence (oracle/phmm_oracle_f128.c), independent of phmm_oracle.c."""
    global _F128
    if _F128 is None:
        path = os.path.join(_HERE, "_build", "libphmm_oracle_f128.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        L.phmm_f128_pair_log10.argtypes = [u8p, u8p, u8p, u8p, u8p, C.c_uint32, u8p, C.c_uint32, f64p, f64p, C.POINTER(C.c_int)]
        L.phmm_f128_pair_log10.restype = C.c_double
        _F128 = L
    return _F128


def pair_f128_log10(b, p, tabs=None):
    """log10 likelihood of pair p of batch b in 113-bit arithmetic, and whether a double at the 2^1020 scale has underflowed to 0."""
    t = tabs or tables()
    r, h = int(b.pair_read[p]), int(b.pair_hap[p])
    ro, rl, ho, hl = int(b.read_off[r]), int(b.read_len[r]), int(b.hap_off[h]), int(b.hap_len[h])
    c = lambda a, o, n: np.ascontiguousarray(a[o:o + n])
    below = C.c_int(0)
    v = f128_lib().phmm_f128_pair_log10(c(b.read_bases, ro, rl), c(b.read_qual, ro, rl), c(b.read_ins, ro, rl), c(b.read_del, ro, rl),
                                        c(b.read_gcp, ro, rl), rl, c(b.hap_bases, ho, hl), hl, np.ascontiguousarray(t["ph2pr_f64"]),
                                        np.ascontiguousarray(t["mm_f64"]), C.byref(below))
    return v, bool(below.value)


def prepare_batch(b, mapq, min_q=18, pcr_model=3, gap_open=45, gap_cont=10):
    """The read table of a batch with RAW qualities as the reference's decompression_data leaves it (pairhmm_internal.cpp:413-448):
    phmm_oracle_modify_quals (H3) and phmm_oracle_apply_pcr_error_model (H4) per read, deletion gap-opens aliasing the insertion
    ones, gap continuation 10.  Returns a new Batch (the pair list and the haplotypes are shared)."""
    from rovaca_amd.batch import Batch
    L = lib()
    qual = b.read_qual.copy()
    gop = np.full(b.read_bases.shape[0], gap_open, np.uint8)
    for r in range(b.n_reads):
        o, n = int(b.read_off[r]), int(b.read_len[r])
        q = np.ascontiguousarray(qual[o:o + n])
        L.phmm_oracle_modify_quals(q, n, int(mapq[r]), int(min_q))
        qual[o:o + n] = q
        g = np.ascontiguousarray(gop[o:o + n])
        L.phmm_oracle_apply_pcr_error_model(np.ascontiguousarray(b.read_bases[o:o + n]), g, n, int(pcr_model))
        gop[o:o + n] = g
    return Batch(b.read_bases, qual, gop, gop.copy(), np.full_like(gop, gap_cont), b.read_off, b.read_len, b.hap_bases, b.hap_off,
                 b.hap_len, b.pair_read, b.pair_hap, region_read=b.region_read, region_hap=b.region_hap, region_pair=b.region_pair)


def ref_pairs_avx512_s(b, n_threads=1):
    """Reference per-pair AVX-512 fp32 kernel over a flat batch.  Returns (raw_f32, seconds)."""
    R = ref()
    raw = np.zeros(int(b.n_pairs), np.float32)
    sec = R.ref_pairs_avx512_s(b.read_bases, b.read_qual, b.read_ins, b.read_del, b.read_gcp, b.read_off, b.read_len,
                               b.hap_bases, b.hap_off, b.hap_len, b.pair_read, b.pair_hap, int(b.n_pairs), raw,
                               n_threads)
    return raw, sec


def ref_regions_16x1(b, n_threads=1):
    """Reference 16x1 AVX-512 kernel over a region-structured batch (needs b.region_*).  (raw_f32, seconds)."""
    R = ref()
    raw = np.zeros(int(b.n_pairs), np.float32)
    sec = R.ref_regions_16x1(b.read_bases, b.read_qual, b.read_ins, b.read_del, b.read_gcp, b.read_off, b.read_len,
                             b.hap_bases, b.hap_off, b.hap_len, b.region_read, b.region_hap, b.region_pair,
                             len(b.region_read) - 1, raw, n_threads)
    return raw, sec


# ------------------------------------------------------------------------------------------------
# Smith-Waterman: restatement (oracle/sw_oracle.c) and the reference's own sw_avx.c / sw_avx2.c
# ------------------------------------------------------------------------------------------------
SW_PARAMS_ASSEMBLE = (200, -150, -260, -11)  # sw_avx.c:13
SW_PARAMS_GENOTYPE = (10, -15, -30, -5)      # sw_avx.c:14
SW_MAX_SEQ_LEN = 2048                        # sw_avx.c:17
_SW_LIB = None
_SW_REF = None
_SW_REF_TRIED = False


class SwResult(C.Structure):
    _fields_ = [("score", C.c_int32), ("max_i", C.c_int16), ("max_j", C.c_int16), ("alignment_offset", C.c_int16),
                ("too_long", C.c_int16), ("cigar_count", C.c_uint32)]


def sw_lib():
    global _SW_LIB
    if _SW_LIB is None:
        path = os.path.join(_HERE, "_build", "libsw_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.sw_oracle_align.argtypes = [u8p, C.c_int32, u8p, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(SwResult),
                                      u32p, C.c_uint32]
        L.sw_oracle_align.restype = C.c_int
        _SW_LIB = L
    return _SW_LIB


def sw_ref():
    """The reference's own Smith-Waterman (oracle/_ref/libsw_ref.so) or None."""
    global _SW_REF, _SW_REF_TRIED
    if _SW_REF_TRIED:
        return _SW_REF
    _SW_REF_TRIED = True
    path = os.path.join(_HERE, "_ref", "libsw_ref.so")
    if not os.path.exists(path) or not cpu_has_avx512():
        return None
    L = C.CDLL(path)
    i32 = C.POINTER(C.c_int32)
    L.ref_sw_align.argtypes = [C.c_int, C.c_int, u8p, C.c_int32, u8p, C.c_int32, i32, i32, i32, i32, u32p, C.c_uint32,
                               C.POINTER(C.c_uint32)]
    L.ref_sw_align.restype = C.c_int
    L.ref_sw_params.argtypes = [C.c_int, i32]
    L.ref_sw_params.restype = C.c_int
    L.ref_sw_batch.argtypes = [C.c_int, u8p, u32p, u32p, u8p, u32p, u32p, u32p, u32p, C.c_uint64, C.c_int, C.POINTER(C.c_int64)]
    L.ref_sw_batch.restype = C.c_double
    _SW_REF = L
    return L


def _as_u8(s):
    return np.ascontiguousarray(np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else s, np.uint8)


def sw_align(seq1, seq2, params=SW_PARAMS_GENOTYPE, max_seq_len=SW_MAX_SEQ_LEN):
    """-> dict(score, max_i, max_j, alignment_offset, too_long, cigar=[u32 BAM elements]) from the restatement."""
    a, b = _as_u8(seq1), _as_u8(seq2)
    cap = len(a) + len(b) + 8
    cig = np.zeros(cap, np.uint32)
    res = SwResult()
    p = (C.c_int32 * 4)(*params)
    rc = sw_lib().sw_oracle_align(a if len(a) else np.zeros(1, np.uint8), len(a), b if len(b) else np.zeros(1, np.uint8),
                                  len(b), p, max_seq_len, C.byref(res), cig, cap)
    if rc != 0:
        raise ValueError("sw_oracle_align failed: %d" % rc)
    return dict(score=res.score, max_i=res.max_i, max_j=res.max_j, alignment_offset=res.alignment_offset,
                too_long=bool(res.too_long), cigar=cig[:res.cigar_count].copy())


def ref_sw_align(seq1, seq2, is_assemble=False, which=0):
    """The same through the reference's object code (which: 0 as selected by sw_avx_init, 1 AVX-512, 2 AVX2)."""
    L = sw_ref()
    a, b = _as_u8(seq1), _as_u8(seq2)
    cap = len(a) + len(b) + 8
    cig = np.zeros(cap, np.uint32)
    sc, mi, mj, off = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    cnt = C.c_uint32()
    rc = L.ref_sw_align(int(is_assemble), which, a, len(a), b, len(b), C.byref(sc), C.byref(mi), C.byref(mj), C.byref(off),
                        cig, cap, C.byref(cnt))
    if rc != 0:
        raise ValueError("ref_sw_align failed: %d" % rc)
    return dict(score=sc.value, max_i=mi.value, max_j=mj.value, alignment_offset=off.value, cigar=cig[:cnt.value].copy())


def cigar_string(cigar):
    return "".join("%d%s" % (int(c) >> 4, "MIDNSHP=XB"[int(c) & 15]) for c in cigar)
