"""ctypes binding of include/smithwaterman_hip.h (librovaca_sw_hip.so): the MI355X Smith-Waterman aligner.
No CPU fallback: the library refuses to create a context without a GPU, and this module raises if it is missing."""
import ctypes as C
import os

# before the process's first HIP call: the engines keep more streams busy than the runtime's default of 4 hardware queues
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PARAMS_ASSEMBLE = (200, -150, -260, -11)  # sw_avx.c:13
PARAMS_GENOTYPE = (10, -15, -30, -5)      # sw_avx.c:14
MAX_SEQ_LEN = 2048


class SwParams(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open", C.c_int32), ("gap_extend", C.c_int32)]


class SwBatch(C.Structure):
    _fields_ = [("n_refs", C.c_uint32), ("n_seqs", C.c_uint32), ("n_pairs", C.c_uint64),
                ("ref_bases", C.c_void_p), ("ref_off", C.c_void_p), ("ref_len", C.c_void_p),
                ("seq_bases", C.c_void_p), ("seq_off", C.c_void_p), ("seq_len", C.c_void_p),
                ("pair_ref", C.c_void_p), ("pair_seq", C.c_void_p)]


class SwStats(C.Structure):
    _fields_ = [("cells", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("n_launches", C.c_uint32)]


ALIGNMENT_DTYPE = np.dtype([("score", np.int32), ("max_i", np.int16), ("max_j", np.int16), ("alignment_offset", np.int16),
                            ("too_long", np.uint16), ("cigar_count", np.uint32), ("cigar_off", np.uint32)])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("ROVACA_AMD_SW_LIB") or os.path.join(_HERE, "lib", "librovaca_sw_hip.so")
        if not os.path.exists(path):
            raise RuntimeError("librovaca_sw_hip.so is missing: run __graft_entry__.build() (there is no CPU fallback)")
        L = C.CDLL(path)
        L.swhip_abi_version.restype = C.c_int
        L.swhip_last_error.restype = C.c_char_p
        L.swhip_kernel_build_id.restype = C.c_char_p
        L.swhip_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.swhip_create.restype = C.c_int
        L.swhip_destroy.argtypes = [C.c_void_p]
        L.swhip_destroy.restype = None
        L.swhip_align.argtypes = [C.c_void_p, C.POINTER(SwParams), C.POINTER(SwBatch), C.c_void_p, C.c_void_p, C.c_uint64,
                                  C.POINTER(C.c_uint64), C.POINTER(SwStats)]
        L.swhip_align.restype = C.c_int
        _LIB = L
    return _LIB


def _tables(seqs):
    arrs = [np.ascontiguousarray(np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else s, np.uint8) for s in seqs]
    lens = np.array([len(a) for a in arrs], np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint32) if len(arrs) else np.zeros(0, np.uint32)
    bases = np.concatenate(arrs) if arrs else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases), offs, lens


class Aligner:
    """One device context (stream + buffers).  align() mirrors what sw_avx512_run leaves in the reference's handle, per pair."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().swhip_create(device, None, C.byref(self._h))
        if rc != 0:
            raise RuntimeError("swhip_create failed (%d): %s" % (rc, lib().swhip_last_error().decode()))

    def close(self):
        if self._h:
            lib().swhip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def align(self, refs, seqs, pairs, params=PARAMS_GENOTYPE):
        """refs, seqs: lists of byte strings / uint8 arrays; pairs: list of (ref index, seq index).
        -> (structured array ALIGNMENT_DTYPE per pair, uint32 array of BAM elements, SwStats)"""
        return self.align_packed(PackedBatch(refs, seqs, pairs), params=params)

    def align_packed(self, pb, params=PARAMS_GENOTYPE, out=None, cig=None):
        """The same on a batch flattened once (PackedBatch), optionally into the caller's result arrays: what a caller that aligns the
        same batch repeatedly, or from several threads (a context each; ctypes releases the GIL during the call), uses."""
        n = pb.n
        out = np.zeros(n, ALIGNMENT_DTYPE) if out is None else out
        cig = np.zeros(max(pb.cap, 1), np.uint32) if cig is None else cig
        used = C.c_uint64()
        st = SwStats()
        par = SwParams(*params)
        rc = lib().swhip_align(self._h, C.byref(par), C.byref(pb.batch), out.ctypes.data_as(C.c_void_p), cig.ctypes.data_as(C.c_void_p),
                               pb.cap, C.byref(used), C.byref(st))
        if rc != 0:
            raise RuntimeError("swhip_align failed (%d): %s" % (rc, lib().swhip_last_error().decode()))
        return out, cig[:used.value], st


class PackedBatch:
    """refs, seqs, pairs flattened into the arrays swhip_align takes (kept alive here), and the worst-case element count."""

    def __init__(self, refs, seqs, pairs):
        self.rb, self.ro, self.rl = _tables(refs)
        self.sb, self.so, self.sl = _tables(seqs)
        self.pr = np.ascontiguousarray([p[0] for p in pairs], np.uint32)
        self.ps = np.ascontiguousarray([p[1] for p in pairs], np.uint32)
        self.n = n = len(pairs)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None  # noqa: E731
        self.batch = SwBatch(len(refs), len(seqs), n, ptr(self.rb), ptr(self.ro), ptr(self.rl), ptr(self.sb), ptr(self.so), ptr(self.sl),
                             ptr(self.pr), ptr(self.ps))
        self.cap = int(sum(int(self.rl[a]) + int(self.sl[c]) + 2 for a, c in pairs if a < len(self.rl) and c < len(self.sl))) if n else 0


def cigar_of(out_row, cigars):
    return cigars[int(out_row["cigar_off"]):int(out_row["cigar_off"]) + int(out_row["cigar_count"])]
