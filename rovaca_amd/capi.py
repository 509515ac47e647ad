"""ctypes binding of the C ABI in include/pairhmm_hip.h (rovaca_amd/lib/librovaca_pairhmm_hip.so).

There is no fallback of any kind: if the HIP library is missing or no device is visible this module raises.
"""
import ctypes as C
import os

# before the process's first HIP call: the engines keep more streams busy than the runtime's default of 4 hardware queues
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ROVACA_AMD_HIP_LIB") or os.path.join(_HERE, "lib", "librovaca_pairhmm_hip.so")  # override: tuning only
_LIB = None

PHMM_OK = 0
MM_TABLE_SIZE = 32640

EXPORTS = ("phmm_abi_version", "phmm_device_count", "phmm_create", "phmm_destroy", "phmm_last_error", "phmm_compute",
           "phmm_upload", "phmm_sweep", "phmm_sweep_timed", "phmm_download", "phmm_synchronize", "phmm_release",
           "phmm_get_tables", "phmm_download_read_table",
           "phmm_create_multi", "phmm_destroy_multi", "phmm_multi_device_count", "phmm_multi_device_id", "phmm_multi_compute",
           "phmm_multi_upload", "phmm_multi_sweep", "phmm_multi_synchronize", "phmm_multi_sweep_timed", "phmm_multi_download",
           "phmm_multi_shard_cells", "phmm_multi_release", "phmm_marginalize", "phmm_create_cpu", "phmm_backend_name", "phmm_plan_slices",
           "phmm_create_multi_cpu", "phmm_compute_regions", "phmm_stage_create", "phmm_stage_destroy", "phmm_stage_get_view", "phmm_stage_submit", "phmm_stage_wait",
           "phmm_stage_pair_first", "phmm_stage_log10", "phmm_plan_bins", "phmm_plan_region_bins", "phmm_kernel_build_id")


class PhmmBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_haps", C.c_uint32), ("n_pairs", C.c_uint64),
                ("read_bases", C.c_void_p), ("read_qual", C.c_void_p), ("read_ins", C.c_void_p),
                ("read_del", C.c_void_p), ("read_gcp", C.c_void_p), ("read_off", C.c_void_p), ("read_len", C.c_void_p),
                ("hap_bases", C.c_void_p), ("hap_off", C.c_void_p), ("hap_len", C.c_void_p),
                ("pair_read", C.c_void_p), ("pair_hap", C.c_void_p), ("prep", C.c_void_p)]


class PhmmReadPrep(C.Structure):
    _fields_ = [("read_mapq", C.c_void_p), ("min_quality_threshold", C.c_int32), ("pcr_model", C.c_int32),
                ("gap_open", C.c_uint8), ("gap_cont", C.c_uint8)]


class PhmmStats(C.Structure):
    _fields_ = [("cells", C.c_uint64), ("n_rescued", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("n_launches", C.c_uint32), ("max_rows_per_lane", C.c_uint32)]


class PhmmRegion(C.Structure):
    _fields_ = [("read_first", C.c_uint32), ("n_reads", C.c_uint32), ("hap_first", C.c_uint32), ("n_haps", C.c_uint32)]


class PhmmStageCaps(C.Structure):
    _fields_ = [("max_regions", C.c_uint32), ("max_reads", C.c_uint32), ("max_haps", C.c_uint32), ("max_bytes", C.c_uint32),
                ("max_pairs", C.c_uint32)]


class PhmmStageView(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("read_off", C.c_void_p), ("read_qual_off", C.c_void_p), ("read_len", C.c_void_p),
                ("read_mapq", C.c_void_p), ("hap_off", C.c_void_p), ("hap_len", C.c_void_p), ("regions", C.c_void_p)]


class PhmmStageParams(C.Structure):
    _fields_ = [("min_quality_threshold", C.c_int32), ("pcr_model", C.c_int32), ("gap_open", C.c_uint8), ("gap_cont", C.c_uint8)]


class PhmmError(RuntimeError):
    pass


def build():
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU); a no-op when up to date."""
    subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc")], check=True)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PhmmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(the PairHMM engine has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    L.phmm_abi_version.restype = C.c_int
    L.phmm_device_count.restype = C.c_int
    L.phmm_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.phmm_create.restype = C.c_int
    L.phmm_destroy.argtypes = [C.c_void_p]
    L.phmm_destroy.restype = None
    L.phmm_last_error.restype = C.c_char_p
    L.phmm_kernel_build_id.restype = C.c_char_p
    L.phmm_compute.argtypes = [C.c_void_p, C.POINTER(PhmmBatch), C.c_void_p, C.c_void_p, C.POINTER(PhmmStats)]
    L.phmm_compute.restype = C.c_int
    L.phmm_upload.argtypes = [C.c_void_p, C.POINTER(PhmmBatch), C.POINTER(C.c_void_p)]
    L.phmm_upload.restype = C.c_int
    L.phmm_sweep.argtypes = [C.c_void_p, C.c_void_p]
    L.phmm_sweep.restype = C.c_int
    L.phmm_sweep_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.phmm_sweep_timed.restype = C.c_int
    L.phmm_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PhmmStats)]
    L.phmm_download.restype = C.c_int
    L.phmm_synchronize.argtypes = [C.c_void_p]
    L.phmm_synchronize.restype = C.c_int
    L.phmm_release.argtypes = [C.c_void_p, C.c_void_p]
    L.phmm_release.restype = None
    L.phmm_download_read_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.phmm_download_read_table.restype = C.c_int
    L.phmm_get_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.phmm_get_tables.restype = C.c_int
    L.phmm_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.phmm_create_multi.restype = C.c_int
    L.phmm_destroy_multi.argtypes = [C.c_void_p]
    L.phmm_destroy_multi.restype = None
    L.phmm_multi_device_count.argtypes = [C.c_void_p]
    L.phmm_multi_device_count.restype = C.c_int
    L.phmm_multi_device_id.argtypes = [C.c_void_p, C.c_int]
    L.phmm_multi_device_id.restype = C.c_int
    L.phmm_multi_compute.argtypes = [C.c_void_p, C.POINTER(PhmmBatch), C.c_void_p, C.c_void_p, C.POINTER(PhmmStats)]
    L.phmm_multi_compute.restype = C.c_int
    L.phmm_multi_upload.argtypes = [C.c_void_p, C.POINTER(PhmmBatch), C.POINTER(C.c_void_p)]
    L.phmm_multi_upload.restype = C.c_int
    L.phmm_multi_sweep.argtypes = [C.c_void_p, C.c_void_p]
    L.phmm_multi_sweep.restype = C.c_int
    L.phmm_multi_synchronize.argtypes = [C.c_void_p]
    L.phmm_multi_synchronize.restype = C.c_int
    L.phmm_multi_sweep_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
    L.phmm_multi_sweep_timed.restype = C.c_int
    L.phmm_multi_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PhmmStats)]
    L.phmm_multi_download.restype = C.c_int
    L.phmm_multi_shard_cells.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.phmm_multi_shard_cells.restype = C.c_int
    L.phmm_multi_release.argtypes = [C.c_void_p, C.c_void_p]
    L.phmm_multi_release.restype = None
    L.phmm_marginalize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                   C.POINTER(C.c_double)]
    L.phmm_marginalize.restype = C.c_int
    L.phmm_plan_slices.argtypes = [C.POINTER(PhmmBatch), C.c_int, C.c_void_p, C.c_void_p]
    L.phmm_plan_slices.restype = C.c_int
    L.phmm_create_cpu.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.phmm_create_cpu.restype = C.c_int
    L.phmm_backend_name.argtypes = [C.c_void_p]
    L.phmm_backend_name.restype = C.c_char_p
    L.phmm_create_multi_cpu.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.phmm_create_multi_cpu.restype = C.c_int
    L.phmm_compute_regions.argtypes = [C.c_void_p, C.POINTER(PhmmBatch), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(PhmmStats)]
    L.phmm_compute_regions.restype = C.c_int
    L.phmm_stage_create.argtypes = [C.c_void_p, C.POINTER(PhmmStageCaps), C.POINTER(C.c_void_p)]
    L.phmm_stage_create.restype = C.c_int
    L.phmm_stage_destroy.argtypes = [C.c_void_p, C.c_void_p]
    L.phmm_stage_destroy.restype = None
    L.phmm_stage_get_view.argtypes = [C.c_void_p, C.POINTER(PhmmStageView), C.POINTER(PhmmStageCaps)]
    L.phmm_stage_get_view.restype = C.c_int
    L.phmm_stage_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PhmmStageParams)]
    L.phmm_stage_submit.restype = C.c_int
    L.phmm_stage_wait.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PhmmStats)]
    L.phmm_stage_wait.restype = C.c_int
    L.phmm_stage_pair_first.argtypes = [C.c_void_p, C.c_uint32]
    L.phmm_stage_pair_first.restype = C.c_uint64
    L.phmm_stage_log10.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.phmm_stage_log10.restype = C.c_int
    L.phmm_plan_bins.argtypes = [C.POINTER(PhmmBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.phmm_plan_bins.restype = C.c_int
    L.phmm_plan_region_bins.argtypes = [C.POINTER(PhmmBatch), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int]
    L.phmm_plan_region_bins.restype = C.c_int
    _LIB = L
    return L


def _check(rc, what):
    if rc != PHMM_OK:
        raise PhmmError(f"{what}: status {rc}: {lib().phmm_last_error().decode()}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c_batch(b, cross=False, prep=None):
    """prep: None, or (read_mapq uint8 array, min_quality_threshold, pcr_model[, gap_open, gap_cont]) for device-side
    quality modification + PCR indel model; b.read_qual then holds raw qualities and b.read_ins/del/gcp are ignored."""
    cb = PhmmBatch()
    cb._keep = None
    if prep is not None:
        mq = np.ascontiguousarray(prep[0], np.uint8)
        pp = PhmmReadPrep(_ptr(mq), int(prep[1]), int(prep[2]), int(prep[3]) if len(prep) > 3 else 45,
                          int(prep[4]) if len(prep) > 4 else 10)
        cb._keep = (mq, pp)
        cb.prep = C.cast(C.pointer(pp), C.c_void_p)
    cb.n_reads, cb.n_haps, cb.n_pairs = b.n_reads, b.n_haps, b.n_pairs
    for k in ("read_bases", "read_qual", "read_ins", "read_del", "read_gcp", "read_off", "read_len", "hap_bases",
              "hap_off", "hap_len"):
        setattr(cb, k, _ptr(getattr(b, k)))
    cb.pair_read = None if cross else _ptr(b.pair_read)
    cb.pair_hap = None if cross else _ptr(b.pair_hap)
    return cb


def regions_of(batch):
    """The batch's region structure (Batch.region_read / region_hap) as an array of phmm_region; its pair list must be the
    regions' cross products in order (the synthetic generator's batches are)."""
    assert batch.region_read is not None, "this batch carries no region structure"
    n = len(batch.region_read) - 1
    reg = np.zeros((n, 4), np.uint32)
    reg[:, 0] = batch.region_read[:-1]
    reg[:, 1] = np.diff(batch.region_read.astype(np.int64))
    reg[:, 2] = batch.region_hap[:-1]
    reg[:, 3] = np.diff(batch.region_hap.astype(np.int64))
    return np.ascontiguousarray(reg)


def plan_bins(batch, cross=False):
    """The launch plan of a pair-list batch (host only): (order, [(rows_per_lane, first, count), ...])."""
    cb = _c_batch(batch, cross)
    n = batch.n_reads * batch.n_haps if cross else batch.n_pairs
    order = np.zeros(n, np.uint32)
    rows, first, count = np.zeros(64, np.int32), np.zeros(64, np.uint32), np.zeros(64, np.uint32)
    nb = lib().phmm_plan_bins(C.byref(cb), _ptr(order), _ptr(rows), _ptr(first), _ptr(count), 64)
    if nb < 0:
        raise PhmmError(f"phmm_plan_bins: status {-nb}: {lib().phmm_last_error().decode()}")
    return order, [(int(rows[i]), int(first[i]), int(count[i])) for i in range(nb)]


def plan_region_bins(batch, regions):
    """The launch plan of a region batch (host only): (slot_first per read, pair_first per region, launches)."""
    cb = _c_batch(batch)
    regions = np.ascontiguousarray(regions, np.uint32).reshape(-1, 4)
    slot = np.zeros(batch.n_reads, np.uint32)
    pf = np.zeros(len(regions) + 1, np.uint32)
    rows, first, count = np.zeros(64, np.int32), np.zeros(64, np.uint32), np.zeros(64, np.uint32)
    nb = lib().phmm_plan_region_bins(C.byref(cb), _ptr(regions), len(regions), _ptr(slot), _ptr(pf), _ptr(rows), _ptr(first), _ptr(count), 64)
    if nb < 0:
        raise PhmmError(f"phmm_plan_region_bins: status {-nb}: {lib().phmm_last_error().decode()}")
    return slot, pf[:-1], [(int(rows[i]), int(first[i]), int(count[i])) for i in range(nb)]


class Stage:
    """Pinned staging for region batches (phmm_stage_*): numpy views of the engine's pinned tables, written in place."""

    def __init__(self, engine, max_regions, max_reads, max_haps, max_bytes, max_pairs):
        self.engine = engine
        caps = PhmmStageCaps(max_regions, max_reads, max_haps, max_bytes, max_pairs)
        h = C.c_void_p()
        _check(lib().phmm_stage_create(engine.handle, C.byref(caps), C.byref(h)), "phmm_stage_create")
        self.handle = h
        v = PhmmStageView()
        _check(lib().phmm_stage_get_view(h, C.byref(v), None), "phmm_stage_get_view")

        def arr(ptr, ctype, n):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,))
        self.bytes = arr(v.bytes, C.c_uint8, max_bytes)
        self.read_off, self.read_qual_off = arr(v.read_off, C.c_uint32, max_reads), arr(v.read_qual_off, C.c_uint32, max_reads)
        self.read_len, self.read_mapq = arr(v.read_len, C.c_uint32, max_reads), arr(v.read_mapq, C.c_uint8, max_reads)
        self.hap_off, self.hap_len = arr(v.hap_off, C.c_uint32, max_haps), arr(v.hap_len, C.c_uint32, max_haps)
        self.regions = arr(v.regions, C.c_uint32, 4 * max_regions).reshape(max_regions, 4)
        self.n_pairs = 0

    def fill(self, batch, regions, mapq):
        """Copy a batch's tables into the stage's arena as [bases][qualities][haplotypes] (tests; the adapter's workers write
        their regions in place instead, read by read)."""
        regions = np.asarray(regions, np.uint32).reshape(-1, 4)
        rb, hb = batch.read_bases.shape[0], batch.hap_bases.shape[0]
        self.bytes[:rb], self.bytes[rb:2 * rb], self.bytes[2 * rb:2 * rb + hb] = batch.read_bases, batch.read_qual, batch.hap_bases
        nr, nh = batch.n_reads, batch.n_haps
        self.read_off[:nr], self.read_qual_off[:nr], self.read_len[:nr] = batch.read_off, batch.read_off + np.uint32(rb), batch.read_len
        self.read_mapq[:nr] = mapq
        self.hap_off[:nh], self.hap_len[:nh] = batch.hap_off + np.uint32(2 * rb), batch.hap_len
        self.regions[:len(regions)] = regions
        self.n_pairs = int((regions[:, 1].astype(np.int64) * regions[:, 3]).sum())
        return len(regions), nr, nh

    def submit(self, n_regions, n_reads, n_haps, min_q=18, pcr_model=3, gap_open=45, gap_cont=10):
        prm = PhmmStageParams(int(min_q), int(pcr_model), int(gap_open), int(gap_cont))
        _check(lib().phmm_stage_submit(self.engine.handle, self.handle, n_regions, n_reads, n_haps, C.byref(prm)), "phmm_stage_submit")

    def wait(self):
        st = PhmmStats()
        _check(lib().phmm_stage_wait(self.engine.handle, self.handle, C.byref(st)), "phmm_stage_wait")
        return st

    def pair_first(self, region):
        return int(lib().phmm_stage_pair_first(self.handle, region))

    def log10(self, first, n):
        out, raw, resc = np.zeros(n, np.float64), np.zeros(n, np.float32), C.c_uint64()
        _check(lib().phmm_stage_log10(self.handle, first, n, _ptr(out), _ptr(raw), C.byref(resc)), "phmm_stage_log10")
        return raw, out, int(resc.value)

    def close(self):
        if self.handle:
            lib().phmm_stage_destroy(self.engine.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tables():
    L = lib()
    a, b = np.zeros(128, np.float32), np.zeros(MM_TABLE_SIZE, np.float32)
    c, d = np.zeros(128, np.float64), np.zeros(MM_TABLE_SIZE, np.float64)
    _check(L.phmm_get_tables(_ptr(a), _ptr(b), _ptr(c), _ptr(d)), "phmm_get_tables")
    return dict(ph2pr_f32=a, mm_f32=b, ph2pr_f64=c, mm_f64=d)


class Resident:
    """A batch living in HBM (phmm_upload)."""

    def __init__(self, engine, handle, n_pairs):
        self.engine, self.handle, self.n_pairs = engine, handle, n_pairs

    def sweep(self):
        _check(lib().phmm_sweep(self.engine.handle, self.handle), "phmm_sweep")

    def sweep_timed(self, iters=1):
        ms, ms32 = C.c_double(), C.c_double()
        _check(lib().phmm_sweep_timed(self.engine.handle, self.handle, iters, C.byref(ms), C.byref(ms32)), "phmm_sweep_timed")
        return ms.value, ms32.value

    def download(self):
        raw = np.zeros(self.n_pairs, np.float32)
        l10 = np.zeros(self.n_pairs, np.float64)
        st = PhmmStats()
        _check(lib().phmm_download(self.engine.handle, self.handle, _ptr(raw), _ptr(l10), C.byref(st)), "phmm_download")
        return raw, l10, st

    def read_table(self):
        """(qual, ins, del, gcp) as the kernels see them -- diagnostic, used to check device-side read preparation."""
        out = [np.zeros(self.read_bytes, np.uint8) for _ in range(4)]
        _check(lib().phmm_download_read_table(self.engine.handle, self.handle, *[_ptr(a) for a in out]),
               "phmm_download_read_table")
        return out

    def release(self):
        if self.handle:
            lib().phmm_release(self.engine.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Engine:
    """One context (device + stream + tables).  ``stream``: integer hipStream_t or None for a private stream."""

    def __init__(self, device=0, stream=None):
        L = lib()
        h = C.c_void_p()
        _check(L.phmm_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)), "phmm_create")
        self.handle = h

    def compute(self, batch, cross=False, prep=None, out=None):
        """Host buffers in, (raw_f32, log10, stats) out.  ``out``: (raw float32[n], log10 float64[n]) to write into -- a caller
        that evaluates batch after batch keeps its result buffers, as the C++ adapter does; fresh arrays are made otherwise."""
        cb = _c_batch(batch, cross, prep)
        n = batch.n_reads * batch.n_haps if cross else batch.n_pairs
        if out is not None:
            raw, l10 = out
            assert raw.dtype == np.float32 and l10.dtype == np.float64 and raw.size >= n and l10.size >= n
            assert raw.flags.c_contiguous and l10.flags.c_contiguous
        else:
            raw = np.zeros(n, np.float32)
            l10 = np.zeros(n, np.float64)
        st = PhmmStats()
        _check(lib().phmm_compute(self.handle, C.byref(cb), _ptr(raw), _ptr(l10), C.byref(st)), "phmm_compute")
        return raw, l10, st

    def compute_regions(self, batch, regions, prep=None):
        """Region-descriptor form of compute (phmm_compute_regions): pair ids count region by region, [read][hap] inside."""
        cb = _c_batch(batch, True, prep)
        regions = np.ascontiguousarray(regions, np.uint32).reshape(-1, 4)
        n = int((regions[:, 1].astype(np.int64) * regions[:, 3]).sum())
        raw, l10, st = np.zeros(n, np.float32), np.zeros(n, np.float64), PhmmStats()
        _check(lib().phmm_compute_regions(self.handle, C.byref(cb), _ptr(regions), len(regions), _ptr(raw), _ptr(l10), C.byref(st)),
               "phmm_compute_regions")
        return raw, l10, st

    def upload(self, batch, cross=False, prep=None):
        cb = _c_batch(batch, cross, prep)
        h = C.c_void_p()
        _check(lib().phmm_upload(self.handle, C.byref(cb), C.byref(h)), "phmm_upload")
        r = Resident(self, h, batch.n_reads * batch.n_haps if cross else batch.n_pairs)
        r.read_bytes = int(batch.read_bases.shape[0])
        return r

    def marginalize(self, old, old2new, n_new, evidence_count=None):
        """Device form of marginal_likelihoods_direct (one sample): old [n_old][stride] -> ([n_new][evidence_count], kernel ms)."""
        old = np.ascontiguousarray(old, np.float64)
        n_old, stride = old.shape
        ev = stride if evidence_count is None else int(evidence_count)
        m = np.ascontiguousarray(old2new, np.int32)
        out = np.empty((n_new, ev), np.float64)
        ms = C.c_double()
        _check(lib().phmm_marginalize(self.handle, _ptr(old), n_old, stride, ev, _ptr(m), n_new, _ptr(out), C.byref(ms)),
               "phmm_marginalize")
        return out, ms.value

    def synchronize(self):
        _check(lib().phmm_synchronize(self.handle), "phmm_synchronize")

    def close(self):
        if self.handle:
            lib().phmm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CpuEngine(Engine):
    """BASELINE config 1: the CPU backend of the same C ABI (phmm_create_cpu) -- this repository's scalar and AVX2 + FMA
    kernels, asked for by name; a device Engine never falls back to it."""

    def __init__(self, n_threads=0, simd=True):
        h = C.c_void_p()
        _check(lib().phmm_create_cpu(int(n_threads), 1 if simd else 0, C.byref(h)), "phmm_create_cpu")
        self.handle = h

    @property
    def backend(self):
        return lib().phmm_backend_name(self.handle).decode()


def plan_slices(batch, n, cross=False):
    """The multi-device cut without a device: (slice index per pair, cells per slice)."""
    cb = _c_batch(batch, cross)
    n_pairs = batch.n_reads * batch.n_haps if cross else batch.n_pairs
    which = np.zeros(n_pairs, np.uint32)
    cells = np.zeros(n, np.uint64)
    _check(lib().phmm_plan_slices(C.byref(cb), int(n), _ptr(which), _ptr(cells)), "phmm_plan_slices")
    return which, cells


class MultiResident:
    """A batch cut into equal-cells slices, one resident slice per device (phmm_multi_upload)."""

    def __init__(self, engine, handle, n_pairs):
        self.engine, self.handle, self.n_pairs = engine, handle, n_pairs

    def sweep(self):
        _check(lib().phmm_multi_sweep(self.engine.handle, self.handle), "phmm_multi_sweep")

    def sweep_timed(self, iters=1):
        """(wall ms per pass = slowest device, per-device ms per pass by HIP events, per-device fp32 kernel ms)."""
        n = self.engine.n_devices
        wall = C.c_double()
        dev, f32 = np.zeros(n, np.float64), np.zeros(n, np.float64)
        _check(lib().phmm_multi_sweep_timed(self.engine.handle, self.handle, iters, C.byref(wall), _ptr(dev), _ptr(f32)),
               "phmm_multi_sweep_timed")
        return wall.value, dev, f32

    def shard_cells(self):
        n = self.engine.n_devices
        cells, pairs = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        _check(lib().phmm_multi_shard_cells(self.handle, _ptr(cells), _ptr(pairs)), "phmm_multi_shard_cells")
        return cells, pairs

    def download(self):
        raw = np.zeros(self.n_pairs, np.float32)
        l10 = np.zeros(self.n_pairs, np.float64)
        st = PhmmStats()
        _check(lib().phmm_multi_download(self.engine.handle, self.handle, _ptr(raw), _ptr(l10), C.byref(st)), "phmm_multi_download")
        return raw, l10, st

    def release(self):
        if self.handle:
            lib().phmm_multi_release(self.engine.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class MultiEngine:
    """Several devices behind one handle (phmm_create_multi): the pair list is cut into equal-cells slices, one host thread
    and one context per device, no collective."""

    def __init__(self, devices=None, cpu_slices=0, cpu_threads_each=1):
        """devices: HIP device ids (distinct, unless PHMM_MULTI_ALLOW_SHARED_DEVICE is set); or cpu_slices > 0 for a handle over
        CPU contexts (phmm_create_multi_cpu)."""
        h = C.c_void_p()
        if cpu_slices:
            _check(lib().phmm_create_multi_cpu(int(cpu_slices), int(cpu_threads_each), 1, C.byref(h)), "phmm_create_multi_cpu")
        else:
            devices = [int(d) for d in devices]
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().phmm_create_multi(arr, len(devices), C.byref(h)), "phmm_create_multi")
        self.handle = h
        self.n_devices = lib().phmm_multi_device_count(h)
        self.devices = [lib().phmm_multi_device_id(h, i) for i in range(self.n_devices)]

    def compute(self, batch, cross=False, prep=None):
        cb = _c_batch(batch, cross, prep)
        n = batch.n_reads * batch.n_haps if cross else batch.n_pairs
        raw = np.zeros(n, np.float32)
        l10 = np.zeros(n, np.float64)
        st = PhmmStats()
        _check(lib().phmm_multi_compute(self.handle, C.byref(cb), _ptr(raw), _ptr(l10), C.byref(st)), "phmm_multi_compute")
        return raw, l10, st

    def upload(self, batch, cross=False, prep=None):
        cb = _c_batch(batch, cross, prep)
        h = C.c_void_p()
        _check(lib().phmm_multi_upload(self.handle, C.byref(cb), C.byref(h)), "phmm_multi_upload")
        return MultiResident(self, h, batch.n_reads * batch.n_haps if cross else batch.n_pairs)

    def synchronize(self):
        _check(lib().phmm_multi_synchronize(self.handle), "phmm_multi_synchronize")

    def close(self):
        if self.handle:
            lib().phmm_destroy_multi(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
