"""Smith-Waterman on the MI355X through the C ABI, bit for bit against the golden vectors generated from
the reference's sw_avx.c and against the CPU restatement on seeded batches."""
import os

import numpy as np
import pytest

import oracle
from rovaca_amd import swapi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "sw_golden.npz")


@pytest.fixture(scope="module")
def aligner():
    a = swapi.Aligner(0)
    yield a
    a.close()


def _check(aligner, refs, seqs, pairs, params):
    out, cig, st = aligner.align(refs, seqs, pairs, params)
    for p, (r, s) in enumerate(pairs):
        o = oracle.sw_align(refs[r], seqs[s], params)
        row = out[p]
        if o["too_long"]:
            assert row["too_long"] == 1 and row["alignment_offset"] == 1 and row["cigar_count"] == 0
            continue
        got = (int(row["score"]), int(row["max_i"]), int(row["max_j"]), int(row["alignment_offset"]))
        assert got == (o["score"], o["max_i"], o["max_j"], o["alignment_offset"]), (p, len(refs[r]), len(seqs[s]), got, o)
        assert np.array_equal(swapi.cigar_of(row, cig), o["cigar"]), (p, oracle.cigar_string(swapi.cigar_of(row, cig)),
                                                                      oracle.cigar_string(o["cigar"]))
    return out, cig, st


@pytest.mark.parametrize("by_copy", [False, True])
@pytest.mark.parametrize("tag", ["genotype", "assemble"])
def test_golden_vectors_bit_exact(aligner, tag, by_copy, monkeypatch):
    # CIGAR elements come back either written by the kernel into pinned host memory (region-sized calls) or through a copy of the
    # used part of the device array (batches whose worst case is above 4 M elements); results records always the first way
    if by_copy:
        monkeypatch.setenv("SWHIP_CIGARS_BY_COPY", "1")
    g = np.load(GOLD)
    params = tuple(int(x) for x in g["params"][0 if tag == "genotype" else 1])
    n = len(g["len1"])
    refs = [g["seq1"][g["off1"][k]:g["off1"][k] + g["len1"][k]] for k in range(n)]
    seqs = [g["seq2"][g["off2"][k]:g["off2"][k] + g["len2"][k]] for k in range(n)]
    out, cig, _ = aligner.align(refs, seqs, [(k, k) for k in range(n)], params)
    res, gc, coff = g[tag + "_result"], g[tag + "_cigar"], g[tag + "_cigar_off"]
    for k in range(n):
        if res[k][4]:
            assert out[k]["too_long"] == 1 and out[k]["alignment_offset"] == 1 and out[k]["cigar_count"] == 0
            continue
        got = (int(out[k]["score"]), int(out[k]["max_i"]), int(out[k]["max_j"]), int(out[k]["alignment_offset"]))
        assert got == tuple(int(x) for x in res[k][:4]), (k, got, res[k])
        assert np.array_equal(swapi.cigar_of(out[k], cig), gc[coff[k]:coff[k + 1]]), k


@pytest.mark.parametrize("lanes", [16, 64])
def test_region_shaped_batch_reads_vs_best_haplotype(aligner, lanes, monkeypatch):
    monkeypatch.setenv("SWHIP_FORCE_LANES", str(lanes))  # both shapes: four alignments per wave, one per wave
    # the genotyping call site: every read of a region against its best haplotype (alignment_utils.cpp:572-579)
    rng = np.random.default_rng(7)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    refs, seqs, pairs = [], [], []
    for region in range(12):
        base = acgt[rng.integers(0, 4, int(rng.integers(250, 450)))]
        haps = []
        for h in range(5):
            hap = base.copy()
            for _ in range(h):
                hap[int(rng.integers(0, len(hap)))] = acgt[rng.integers(0, 4)]
            haps.append(hap)
        h0 = len(refs)
        refs += haps
        for r in range(30):
            hap = haps[int(rng.integers(0, 5))]
            n = int(rng.integers(30, 151))
            a = int(rng.integers(0, len(hap) - n))
            read = hap[a:a + n].copy()
            read[rng.random(n) < 0.01] = ord("A")
            if r % 5 == 0:
                k = int(rng.integers(1, n - 1))
                read = np.concatenate([read[:k], acgt[rng.integers(0, 4, int(rng.integers(1, 9)))], read[k:]])
            if r % 7 == 0:
                k = int(rng.integers(1, len(read) - 10))
                read = np.concatenate([read[:k], read[k + int(rng.integers(1, 6)):]])
            seqs.append(read)
            pairs.append((h0 + int(rng.integers(0, 5)), len(seqs) - 1))
    _check(aligner, refs, seqs, pairs, swapi.PARAMS_GENOTYPE)


def test_backtrack_memory_is_reused_across_launches(aligner, monkeypatch):
    # a batch whose backtrack flags exceed the per-launch budget is cut into launches that reuse the same memory
    monkeypatch.setenv("SWHIP_BT_WORDS_PER_LAUNCH", "60000")
    rng = np.random.default_rng(12)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    refs = [acgt[rng.integers(0, 4, int(rng.integers(200, 400)))] for _ in range(8)]
    seqs, pairs = [], []
    for k in range(60):
        ref = refs[k % 8]
        n = int(rng.integers(40, 150))
        a = int(rng.integers(0, len(ref) - n))
        s = ref[a:a + n].copy()
        s[rng.random(n) < 0.04] = ord("A")
        seqs.append(s)
        pairs.append((k % 8, k))
    out, cig, st = _check(aligner, refs, seqs, pairs, swapi.PARAMS_GENOTYPE)
    assert st.n_launches > 3


@pytest.mark.parametrize("merge", [False, True])
@pytest.mark.parametrize("lanes,lds_flags", [(16, False), (64, False), (64, True)])
def test_every_geometry_and_long_sequences(aligner, lanes, lds_flags, merge, monkeypatch):
    # both shapes; with one alignment per wave the backtrack flags live in LDS for small batches, in global memory otherwise.  In a
    # small batch a columns-per-lane class rides along with the next one up (its alignments leave the upper lanes idle); without the
    # merge every instantiation gets its own launch
    if not merge:
        monkeypatch.setenv("SWHIP_NO_CLASS_MERGE", "1")
    monkeypatch.setenv("SWHIP_FORCE_LANES", str(lanes))
    if lds_flags:
        monkeypatch.setenv("SWHIP_LDS_FLAGS", "1")
    # seq2 lengths across every columns-per-lane instantiation of both shapes (16 and 64 lanes per alignment)
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    refs, seqs, pairs = [], [], []
    for n2 in [1, 2, 3, 31, 32, 33, 64, 65, 96, 97, 128, 129, 160, 161, 192, 193, 224, 225, 255, 256, 257, 300, 350, 511, 512, 513, 600,
               700, 1023, 1024, 1025, 1500, 2047]:
        n1 = int(rng.integers(max(1, n2 // 2), min(2047, 2 * n2 + 40) + 1))
        ref = acgt[rng.integers(0, 4, n1)]
        if n2 <= n1:
            a = int(rng.integers(0, n1 - n2 + 1))
            seq = ref[a:a + n2].copy()
            seq[rng.random(n2) < 0.03] = ord("T")
        else:
            seq = np.concatenate([ref, acgt[rng.integers(0, 4, n2 - n1)]])
        refs.append(ref)
        seqs.append(seq)
        pairs.append((len(refs) - 1, len(seqs) - 1))
    for params in (swapi.PARAMS_GENOTYPE, swapi.PARAMS_ASSEMBLE):
        _check(aligner, refs, seqs, pairs, params)


@pytest.mark.parametrize("lanes", [16, 64])
def test_gap_rich_paths_leave_the_diagonal_often(aligner, lanes, monkeypatch):
    # the traceback is walked by all lanes of an alignment together, fetching the flags of the next cells along the diagonal
    # (sw_kernels.hip): paths with many short and long gaps, runs that open and extend in both directions, unrelated sequences
    # and low-complexity repeats make it re-fetch at every kind of move; parameters with cheap gaps make the gaps frequent
    monkeypatch.setenv("SWHIP_FORCE_LANES", str(lanes))
    rng = np.random.default_rng(31 + lanes)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    refs, seqs, pairs = [], [], []
    for case in range(120):
        n1 = int(rng.integers(20, 700))
        ref = acgt[rng.integers(0, 4, n1)] if case % 5 else np.tile(acgt[rng.integers(0, 4, 3)], n1 // 3 + 1)[:n1]
        kind = case % 4
        if kind == 0:  # unrelated
            seq = acgt[rng.integers(0, 4, int(rng.integers(10, 400)))]
        else:          # a copy with insertions and deletions of 1..12 bases every 8..40 bases
            out, x = [], 0
            while x < n1:
                step = int(rng.integers(8, 40))
                out.append(ref[x:x + step])
                x += step
                g = int(rng.integers(1, 13))
                if rng.random() < 0.5:
                    out.append(acgt[rng.integers(0, 4, g)])  # insertion
                else:
                    x += g                                   # deletion
            seq = np.concatenate(out)[:600]
            if kind == 3 and len(seq) > 30:
                seq = seq[int(rng.integers(0, 15)):len(seq) - int(rng.integers(0, 15))]
        refs.append(ref)
        seqs.append(seq)
        pairs.append((len(refs) - 1, len(seqs) - 1))
    for params in (swapi.PARAMS_GENOTYPE, swapi.PARAMS_ASSEMBLE, (3, -2, -3, -1), (1, -1, -1, -1)):
        _check(aligner, refs, seqs, pairs, params)


def test_too_long_and_invalid(aligner):
    rng = np.random.default_rng(10)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    big = acgt[rng.integers(0, 4, 2048)]
    out, cig, _ = _check(aligner, [big, big[:100]], [big[:50], big], [(0, 0), (1, 1), (1, 0)], swapi.PARAMS_GENOTYPE)
    assert out[0]["too_long"] == 1 and out[1]["too_long"] == 1 and out[2]["too_long"] == 0
    with pytest.raises(RuntimeError):
        aligner.align([b"ACGT"], [b""], [(0, 0)])
    with pytest.raises(RuntimeError):
        aligner.align([b"ACGT"], [b"AC"], [(0, 1)])
    out, cig, _ = aligner.align([], [], [])
    assert len(out) == 0 and len(cig) == 0


def test_drop_in_candidate_through_the_reference_handle():
    # sw_hip_run installed into lib_sw_avx.avx_function and called the way the reference's call sites do
    # (alignment_utils.cpp:572-579): same fields out as sw_avx512_run, for both parameter sets, including the length guard
    import ctypes as C
    L = C.CDLL(os.path.join(os.path.dirname(swapi.__file__), "lib", "librovaca_sw_host.so"))
    i32 = C.POINTER(C.c_int32)
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    L.rovaca_amd_sw_run_through_handle.argtypes = [u8p, C.c_int32, u8p, C.c_int32, C.c_int, i32, i32, i32, i32, u32p, C.c_uint32,
                                                   C.POINTER(C.c_uint32)]
    L.rovaca_amd_sw_run_through_handle.restype = C.c_int
    rng = np.random.default_rng(21)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for t in range(12):
        n1 = int(rng.integers(20, 600)) if t < 10 else 2048
        ref = acgt[rng.integers(0, 4, n1)]
        n2 = int(rng.integers(5, min(n1, 300)))
        a = int(rng.integers(0, n1 - n2 + 1))
        seq = np.ascontiguousarray(ref[a:a + n2].copy())
        seq[rng.random(n2) < 0.05] = ord("G")
        for asm in (0, 1):
            sc, mi, mj, off, cnt = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint32()
            cig = np.zeros(n1 + n2 + 8, np.uint32)
            rc = L.rovaca_amd_sw_run_through_handle(np.ascontiguousarray(ref), n1, seq, n2, asm, C.byref(sc), C.byref(mi),
                                                    C.byref(mj), C.byref(off), cig, len(cig), C.byref(cnt))
            assert rc == 0
            o = oracle.sw_align(ref, seq, oracle.SW_PARAMS_ASSEMBLE if asm else oracle.SW_PARAMS_GENOTYPE)
            if o["too_long"]:
                assert off.value == 1 and cnt.value == 0
                continue
            assert (sc.value, mi.value, mj.value, off.value) == (o["score"], o["max_i"], o["max_j"], o["alignment_offset"])
            assert np.array_equal(cig[:cnt.value], o["cigar"])


def test_concurrent_callers_are_batched_and_each_gets_its_own_result():
    # sw_hip_run -> sw_hip_align_batch -> dispatcher threads that merge concurrent requests into one device call
    import ctypes as C
    import threading
    L = C.CDLL(os.path.join(os.path.dirname(swapi.__file__), "lib", "librovaca_sw_host.so"))
    i32 = C.POINTER(C.c_int32)
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    L.rovaca_amd_sw_run_through_handle.argtypes = [u8p, C.c_int32, u8p, C.c_int32, C.c_int, i32, i32, i32, i32, u32p, C.c_uint32,
                                                   C.POINTER(C.c_uint32)]
    L.rovaca_amd_sw_run_through_handle.restype = C.c_int
    rng = np.random.default_rng(31)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    jobs = []
    for t in range(64):
        ref = acgt[rng.integers(0, 4, int(rng.integers(100, 500)))]
        n2 = int(rng.integers(20, 100))
        a = int(rng.integers(0, len(ref) - n2 + 1))
        seq = np.ascontiguousarray(ref[a:a + n2].copy())
        seq[rng.random(n2) < 0.05] = ord("C")
        jobs.append((np.ascontiguousarray(ref), seq, t % 2))
    results = [None] * len(jobs)

    def work(lo, hi):
        for k in range(lo, hi):
            ref, seq, asm = jobs[k]
            sc, mi, mj, off, cnt = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint32()
            cig = np.zeros(len(ref) + len(seq) + 8, np.uint32)
            rc = L.rovaca_amd_sw_run_through_handle(ref, len(ref), seq, len(seq), asm, C.byref(sc), C.byref(mi), C.byref(mj),
                                                    C.byref(off), cig, len(cig), C.byref(cnt))
            results[k] = (rc, sc.value, mi.value, mj.value, off.value, cig[:cnt.value].copy())

    th = [threading.Thread(target=work, args=(4 * i, 4 * i + 4)) for i in range(16)]
    [t.start() for t in th]
    [t.join() for t in th]
    for (ref, seq, asm), (rc, sc, mi, mj, off, cig) in zip(jobs, results):
        o = oracle.sw_align(ref, seq, oracle.SW_PARAMS_ASSEMBLE if asm else oracle.SW_PARAMS_GENOTYPE)
        assert rc == 0 and (sc, mi, mj, off) == (o["score"], o["max_i"], o["max_j"], o["alignment_offset"])
        assert np.array_equal(cig, o["cigar"])


@pytest.mark.parametrize("depth", [1, 3])
def test_align_batch_in_two_halves_keeps_requests_in_flight(depth):
    # sw_hip_align_batch_begin / _finish (SURVEY 8f next-1's futures, for the second seam): several requests of one caller in
    # flight at a time give the results of the synchronous call, request by request, empty requests and both parameter sets included
    import ctypes as C
    L = C.CDLL(os.path.join(os.path.dirname(swapi.__file__), "lib", "librovaca_sw_host.so"))
    f = L.rovaca_amd_sw_batches_in_flight
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 7 + [C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(57)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for asm in (0, 1):
        refs, seqs, starts = [], [], [0]
        for g in range(9):
            for _ in range(0 if g == 4 else int(rng.integers(1, 30))):  # request 4 is empty
                ref = acgt[rng.integers(0, 4, int(rng.integers(60, 420)))]
                n2 = int(rng.integers(15, min(len(ref), 160)))
                a = int(rng.integers(0, len(ref) - n2 + 1))
                seq = ref[a:a + n2].copy()
                seq[rng.random(n2) < 0.04] = ord("G")
                if rng.random() < 0.3:
                    k = int(rng.integers(1, n2 - 1))
                    seq = np.concatenate([seq[:k], acgt[rng.integers(0, 4, 4)], seq[k:]])
                refs.append(ref)
                seqs.append(seq)
            starts.append(len(refs))
        n = len(refs)
        rlen, slen = np.array([len(x) for x in refs], np.uint32), np.array([len(x) for x in seqs], np.uint32)
        roff = np.concatenate([[0], np.cumsum(rlen)[:-1]]).astype(np.uint32)
        soff = np.concatenate([[0], np.cumsum(slen)[:-1]]).astype(np.uint32)
        rb, sb = np.ascontiguousarray(np.concatenate(refs)), np.ascontiguousarray(np.concatenate(seqs))
        cap = (rlen + slen + 8).astype(np.uint32)
        cat = np.concatenate([[0], np.cumsum(cap)[:-1]]).astype(np.uint32)
        result, cigars = np.zeros((n, 6), np.int32), np.zeros(int(cap.sum()), np.uint32)
        st = np.array(starts, np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        assert f(p(rb), p(roff), p(rlen), p(sb), p(soff), p(slen), p(st), len(starts) - 1, asm, depth, p(result), p(cigars), p(cat)) == 0
        for k in range(n):
            o = oracle.sw_align(refs[k], seqs[k], oracle.SW_PARAMS_ASSEMBLE if asm else oracle.SW_PARAMS_GENOTYPE)
            assert tuple(int(x) for x in result[k][:4]) == (o["score"], o["max_i"], o["max_j"], o["alignment_offset"]), k
            assert np.array_equal(cigars[cat[k]:cat[k] + result[k][5]], o["cigar"]), k


@pytest.mark.parametrize("mixed", [0, 1])
def test_c_batch_entry_fills_every_handle_like_the_one_alignment_seam(mixed):
    # sw_hip_align_many: the C entry for the assembler's call site (hc_assemble_cigar_cacl.c:417-428) -- n handles laid out as
    # sw_avx_init does, one device call per parameter set, every handle left as sw_hip_run / sw_avx512_run would leave it.
    # The committed "assemble" goldens (generated from the reference's sw_avx.c) through it, plus the length guard.
    import ctypes as C
    L = C.CDLL(os.path.join(os.path.dirname(swapi.__file__), "lib", "librovaca_sw_host.so"))
    f = L.rovaca_amd_sw_many_through_handles
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    refs, seqs = [], []
    for t in range(40):
        n1 = int(rng.integers(30, 700)) if t % 13 else 2048  # a few at the length guard: alignment_offset = 1, nothing else
        ref = acgt[rng.integers(0, 4, n1)]
        n2 = int(rng.integers(10, min(n1, 400)))
        a = int(rng.integers(0, n1 - n2 + 1))
        seq = ref[a:a + n2].copy()
        seq[rng.random(n2) < 0.04] = ord("T")
        if t % 5 == 0:
            k = int(rng.integers(1, n2 - 1))
            seq = np.concatenate([seq[:k], acgt[rng.integers(0, 4, 4)], seq[k:]])
        refs.append(np.ascontiguousarray(ref))
        seqs.append(np.ascontiguousarray(seq))
    rb, ro, rl = swapi._tables(refs)
    sb, so, sl = swapi._tables(seqs)
    n = len(refs)
    cigar_at = np.concatenate([[0], np.cumsum(rl + sl + 8)[:-1]]).astype(np.uint32)
    result = np.zeros((n, 5), np.int32)
    cigars = np.zeros(int((rl + sl + 8).sum()), np.uint32)
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    for asm in (0, 1):
        rc = f(ptr(rb), ptr(ro), ptr(rl), ptr(sb), ptr(so), ptr(sl), n, asm, mixed, ptr(result), ptr(cigars), ptr(cigar_at))
        assert rc == 0
        for k in range(n):
            asm_k = (k & 1) if mixed else asm
            o = oracle.sw_align(refs[k], seqs[k], oracle.SW_PARAMS_ASSEMBLE if asm_k else oracle.SW_PARAMS_GENOTYPE)
            if o["too_long"]:
                assert result[k, 3] == 1 and result[k, 4] == 0
                continue
            assert tuple(result[k, :4]) == (o["score"], o["max_i"], o["max_j"], o["alignment_offset"])
            assert np.array_equal(cigars[cigar_at[k]:cigar_at[k] + result[k, 4]], o["cigar"])


@pytest.mark.parametrize("tag", ["genotype", "assemble"])
def test_c_batch_entry_on_the_reference_made_goldens(tag):
    # the committed goldens (values the reference's own sw_avx.c returned, oracle/gen_golden_sw.py) through sw_hip_align_many
    import ctypes as C
    L = C.CDLL(os.path.join(os.path.dirname(swapi.__file__), "lib", "librovaca_sw_host.so"))
    f = L.rovaca_amd_sw_many_through_handles
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    g = np.load(GOLD)
    n = len(g["len1"])
    rb, ro, rl = np.ascontiguousarray(g["seq1"], np.uint8), np.ascontiguousarray(g["off1"], np.uint32), np.ascontiguousarray(g["len1"], np.uint32)
    sb, so, sl = np.ascontiguousarray(g["seq2"], np.uint8), np.ascontiguousarray(g["off2"], np.uint32), np.ascontiguousarray(g["len2"], np.uint32)
    cigar_at = np.concatenate([[0], np.cumsum(rl + sl + 8)[:-1]]).astype(np.uint32)
    result, cigars = np.zeros((n, 5), np.int32), np.zeros(int((rl + sl + 8).sum()), np.uint32)
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    assert f(ptr(rb), ptr(ro), ptr(rl), ptr(sb), ptr(so), ptr(sl), n, 1 if tag == "assemble" else 0, 0, ptr(result), ptr(cigars), ptr(cigar_at)) == 0
    res, gc, coff = g[tag + "_result"], g[tag + "_cigar"], g[tag + "_cigar_off"]
    for k in range(n):
        if res[k][4]:
            assert result[k, 3] == 1 and result[k, 4] == 0
            continue
        assert tuple(int(x) for x in result[k, :4]) == tuple(int(x) for x in res[k][:4]), k
        assert np.array_equal(cigars[cigar_at[k]:cigar_at[k] + result[k, 4]], gc[coff[k]:coff[k + 1]]), k


def _mutated_reads(rng, n_refs, n_reads, ref_len, read_len):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    refs = [acgt[rng.integers(0, 4, ref_len)] for _ in range(n_refs)]
    seqs, pairs = [], []
    for i in range(n_reads):
        h = refs[i % n_refs]
        a = int(rng.integers(0, ref_len - read_len + 1))
        r = h[a:a + read_len].copy()
        r[rng.random(read_len) < 0.03] = ord("G")
        if i % 5 == 0:
            k = int(rng.integers(1, read_len - 1))
            r = np.concatenate([r[:k], acgt[rng.integers(0, 4, 2)], r[k:]])
        seqs.append(r)
        pairs.append((i % n_refs, i))
    return refs, seqs, pairs


@pytest.mark.parametrize("lanes,read_len", [(16, 40), (16, 100), (16, 150), (16, 250), (64, 500), (64, 900)])
def test_score_table_and_compare_sweeps_agree_and_other_bytes_fall_back(aligner, lanes, read_len, monkeypatch):
    # The kernels with up to 16 columns per lane look the scores up in a per-alignment table of the bases A, C, G, T (sw_kernels.hip;
    # with one alignment per wave from 8 columns per lane on); a workgroup with any other byte in its sequences -- N, lower case,
    # anything: the reference compares bytes -- sweeps with the compare.  Clean and unclean alignments in the same launch, against
    # the CPU restatement, and the same batch with the table switched off.
    monkeypatch.setenv("SWHIP_FORCE_LANES", str(lanes))
    rng = np.random.default_rng(300 + read_len)
    refs, seqs, pairs = _mutated_reads(rng, 12, 160, max(400, read_len + 250), read_len)
    # (a launch takes the alignments in order of their index here -- equal seq1 lengths --, sixteen or four per workgroup: the first
    # workgroups see other bytes, the rest take the table)
    for i in range(0, 48, 7):  # a byte that is none of ACGT in some reads
        b = [ord("N"), ord("a"), 0, 255, ord("R")][(i // 7) % 5]
        seqs[i] = seqs[i].copy()
        seqs[i][int(rng.integers(0, len(seqs[i])))] = b
    with_n = refs[3].copy()
    with_n[[5, 77, 300]] = ord("N")  # and a haplotype with Ns (N == N is a match, as any equal bytes are)
    refs.append(with_n)
    seqs[2] = seqs[2].copy()
    seqs[2][10] = ord("N")
    for i in (2, 3, 20):
        pairs[i] = (len(refs) - 1, i)
    for params in (swapi.PARAMS_GENOTYPE, swapi.PARAMS_ASSEMBLE):
        out, cig, _ = _check(aligner, refs, seqs, pairs, params)
        monkeypatch.setenv("SWHIP_NO_SCORE_TABLE", "1")
        out2, cig2, _ = aligner.align(refs, seqs, pairs, params)
        monkeypatch.delenv("SWHIP_NO_SCORE_TABLE")
        for f in ("score", "max_i", "max_j", "alignment_offset", "cigar_count"):
            assert np.array_equal(out[f], out2[f]), f
        for p in range(len(pairs)):
            assert np.array_equal(swapi.cigar_of(out[p], cig), swapi.cigar_of(out2[p], cig2)), p


def test_score_table_at_the_parameter_limits(aligner, monkeypatch):
    # (scoring parameters are limited to [-512, 512]: the table's 16-bit entries always hold them)
    monkeypatch.setenv("SWHIP_FORCE_LANES", "16")
    rng = np.random.default_rng(77)
    refs, seqs, pairs = _mutated_reads(rng, 4, 48, 300, 120)
    for params in ((512, -512, -512, -1), (-512, 512, -3, -2), (1, -1, -512, -512)):
        _check(aligner, refs, seqs, pairs, params)


def test_several_contexts_on_several_threads_agree():
    # what bench.py's `sustained` leg and the adapter's dispatchers do: a context per host thread, calls in flight together
    import threading

    rng = np.random.default_rng(11)
    refs, seqs, pairs = _mutated_reads(rng, 40, 4000, 400, 150)
    pb = swapi.PackedBatch(refs, seqs, pairs)
    als = [swapi.Aligner(0) for _ in range(3)]
    res = [None] * len(als)

    def work(i):
        for _ in range(3):
            res[i] = als[i].align_packed(pb)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(als))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    out0, cig0, _ = res[0]
    for out, cig, _ in res[1:]:
        for f in ("score", "max_i", "max_j", "alignment_offset", "cigar_count"):
            assert np.array_equal(out[f], out0[f]), f
        for p in range(0, len(pairs), 37):  # (element offsets come from an atomic cursor: compare the elements themselves)
            assert np.array_equal(swapi.cigar_of(out[p], cig), swapi.cigar_of(out0[p], cig0)), p
    for p in range(0, len(pairs), 101):
        o = oracle.sw_align(refs[pairs[p][0]], seqs[pairs[p][1]])
        assert int(out0[p]["score"]) == o["score"] and np.array_equal(swapi.cigar_of(out0[p], cig0), o["cigar"])
    for a in als:
        a.close()
