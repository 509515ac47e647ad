"""ctypes binding of the C++ host adapter (rovaca_amd/lib/librovaca_pairhmm_host.so): the rovaca::call_pairhmm mirror
and the host logic around the kernels (quality modification, PCR indel model, normalisation, read filter)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librovaca_pairhmm_host.so")
_LIB = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

PCR = {"NONE": 0, "HOSTILE": 1, "AGGRESSIVE": 2, "CONSERVATIVE": 3}  # common/enum.h:8

EXPORTS = ("rovaca_amd_modify_read_qualities", "rovaca_amd_tandem_repeat_length", "rovaca_amd_apply_pcr_error_model",
           "rovaca_amd_normalize_likelihoods", "rovaca_amd_poorly_modelled_reads", "rovaca_amd_init_pairhmm_ptr_gpu",
           "rovaca_amd_call_region", "rovaca_amd_batching_stats", "rovaca_amd_marginal_likelihoods_direct", "rovaca_amd_marginalize_through_adapter")


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
    L = C.CDLL(LIB_PATH)
    L.rovaca_amd_modify_read_qualities.argtypes = [u8p, C.c_uint32, C.c_uint8, C.c_int32]
    L.rovaca_amd_modify_read_qualities.restype = None
    L.rovaca_amd_tandem_repeat_length.argtypes = [u8p, C.c_uint32, C.c_uint32]
    L.rovaca_amd_tandem_repeat_length.restype = C.c_int32
    L.rovaca_amd_apply_pcr_error_model.argtypes = [u8p, u8p, C.c_uint32, C.c_int]
    L.rovaca_amd_apply_pcr_error_model.restype = None
    L.rovaca_amd_normalize_likelihoods.argtypes = [f64p, C.c_uint32, C.c_uint32]
    L.rovaca_amd_normalize_likelihoods.restype = None
    L.rovaca_amd_poorly_modelled_reads.argtypes = [f64p, i32p, C.c_uint32, C.c_uint32, u8p]
    L.rovaca_amd_poorly_modelled_reads.restype = C.c_uint32
    L.rovaca_amd_marginal_likelihoods_direct.argtypes = [f64p, C.c_uint32, C.c_uint32, C.c_uint32, i32p, C.c_uint32, f64p]
    L.rovaca_amd_marginal_likelihoods_direct.restype = None
    L.rovaca_amd_marginalize_through_adapter.argtypes = L.rovaca_amd_marginal_likelihoods_direct.argtypes
    L.rovaca_amd_marginalize_through_adapter.restype = None
    L.rovaca_amd_init_pairhmm_ptr_gpu.argtypes = [C.c_int]
    L.rovaca_amd_init_pairhmm_ptr_gpu.restype = C.c_int
    L.rovaca_amd_call_region.argtypes = [u8p, u8p, u32p, u32p, u8p, C.c_uint32, u8p, u32p, u32p, C.c_uint32, C.c_int32,
                                         C.c_int, f64p, u32p]
    L.rovaca_amd_call_region.restype = C.c_int32
    L.rovaca_amd_batching_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rovaca_amd_batching_stats.restype = None
    L.rovaca_amd_batcher_stats.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rovaca_amd_batcher_stats.restype = C.c_int
    _LIB = L
    return L


def batching_stats():
    """(regions evaluated, device submissions) of the cross-region batcher."""
    r, s = C.c_uint64(), C.c_uint64()
    lib().rovaca_amd_batching_stats(C.byref(r), C.byref(s))
    return int(r.value), int(s.value)


def batcher_stats():
    """[(device, regions, submissions)] per batcher of the adapter (one batcher per device in use)."""
    out, n, i = [], 1, 0
    while i < n:
        d, r, s = C.c_int(-1), C.c_uint64(0), C.c_uint64(0)
        n = lib().rovaca_amd_batcher_stats(i, C.byref(d), C.byref(r), C.byref(s))
        if i < n:
            out.append((d.value, r.value, s.value))
        i += 1
    return out


def init_pairhmm_ptr_gpu(device=-1):
    """Install the MI355X engine into rovaca::call_pairhmm.  False when no device is visible."""
    return bool(lib().rovaca_amd_init_pairhmm_ptr_gpu(int(device)))


def call_region(read_bases, read_qual, read_off, read_len, read_mq, hap_bases, hap_off, hap_len, min_quality_threshold=18,
                pcr_model="CONSERVATIVE"):
    """One region through rovaca::call_pairhmm.  Returns (matrix [n_haps][n_reads], survivors)."""
    L = lib()
    n_reads, n_haps = len(read_len), len(hap_len)
    out = np.full((n_haps, n_reads), np.nan)
    surv = np.zeros(max(n_reads, 1), np.uint32)
    n = L.rovaca_amd_call_region(np.ascontiguousarray(read_bases, np.uint8), np.ascontiguousarray(read_qual, np.uint8),
                                 np.ascontiguousarray(read_off, np.uint32), np.ascontiguousarray(read_len, np.uint32),
                                 np.ascontiguousarray(read_mq, np.uint8), n_reads, np.ascontiguousarray(hap_bases, np.uint8),
                                 np.ascontiguousarray(hap_off, np.uint32), np.ascontiguousarray(hap_len, np.uint32), n_haps,
                                 int(min_quality_threshold), PCR[pcr_model] if isinstance(pcr_model, str) else int(pcr_model),
                                 out.reshape(-1), surv)
    if n < 0:
        raise RuntimeError("rovaca::call_pairhmm is not installed: init_pairhmm_ptr_gpu() found no device")
    return out, surv[:n].copy()


def call_regions_in_flight(regions, depth, min_quality_threshold=18, pcr_model="CONSERVATIVE"):
    """Several regions through the begin/finish form of the seam, ``depth`` in flight at a time.  ``regions``: a list of
    (read_bases, read_qual, read_off, read_len, read_mq, hap_bases, hap_off, hap_len) as for call_region.  Returns a list of
    (matrix [n_haps][n_reads], survivors) in the same order."""
    L = lib()
    rb, rq, ro, rl, mq, hb, ho, hl, rr, rh = [], [], [], [], [], [], [], [], [0], [0]
    rbytes = hbytes = 0
    for (b, q, off, ln, m, hbs, hof, hln) in regions:
        b, q, hbs = (np.ascontiguousarray(x, np.uint8) for x in (b, q, hbs))
        rb.append(b)
        rq.append(q)
        ro.append(np.asarray(off, np.uint32) + rbytes)
        rl.append(np.asarray(ln, np.uint32))
        mq.append(np.asarray(m, np.uint8))
        hb.append(hbs)
        ho.append(np.asarray(hof, np.uint32) + hbytes)
        hl.append(np.asarray(hln, np.uint32))
        rbytes += len(b)
        hbytes += len(hbs)
        rr.append(rr[-1] + len(ln))
        rh.append(rh[-1] + len(hln))
    cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(0, dt), dt)
    rr, rh = np.asarray(rr, np.uint32), np.asarray(rh, np.uint32)
    sizes = [(int(rh[g + 1] - rh[g]), int(rr[g + 1] - rr[g])) for g in range(len(regions))]
    out = np.full(max(1, sum(a * b for a, b in sizes)), np.nan)
    surv = np.zeros(max(1, int(rr[-1])), np.uint32)
    nsurv = np.zeros(max(1, len(regions)), np.uint32)
    f = L.rovaca_amd_call_regions_in_flight
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p] * 10 + [C.c_uint32, C.c_int32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    arrs = [cat(rb, np.uint8), cat(rq, np.uint8), cat(ro, np.uint32), cat(rl, np.uint32), cat(mq, np.uint8), cat(hb, np.uint8),
            cat(ho, np.uint32), cat(hl, np.uint32), rr, rh]
    rc = f(*[ptr(a) for a in arrs], len(regions), int(min_quality_threshold),
           PCR[pcr_model] if isinstance(pcr_model, str) else int(pcr_model), int(depth), ptr(out), ptr(surv), ptr(nsurv))
    if rc != 0:
        raise RuntimeError("the MI355X engine is not installed: init_pairhmm_ptr_gpu() found no device")
    res, at = [], 0
    for g, (nh, nr) in enumerate(sizes):
        res.append((out[at:at + nh * nr].reshape(nh, nr).copy(), surv[int(rr[g]):int(rr[g]) + int(nsurv[g])].copy()))
        at += nh * nr
    return res


def marginal_likelihoods_direct(old, old2new, n_new, evidence_count=None, through_adapter=False):
    """old: [n_old][stride] doubles; returns [n_new][evidence_count] (allele_likelihoods.hpp:702-732, one sample).
    through_adapter: go through rovaca::marginal_likelihoods_direct (pmr vectors) instead of the flat routine."""
    old = np.ascontiguousarray(old, np.float64)
    n_old, stride = old.shape
    ev = stride if evidence_count is None else int(evidence_count)
    out = np.empty((n_new, ev), np.float64)
    fn = lib().rovaca_amd_marginalize_through_adapter if through_adapter else lib().rovaca_amd_marginal_likelihoods_direct
    fn(old.reshape(-1), n_old, stride, ev, np.ascontiguousarray(old2new, np.int32), n_new, out.reshape(-1))
    return out
