"""Generate tests/golden/sw_golden.npz from the REFERENCE's own Smith-Waterman (oracle/_ref/libsw_ref.so, i.e.
common/base/lib/sw_avx/sw_avx.c compiled as it is).  Run in the build container, where /root/reference exists:

    python -m oracle.gen_golden_sw

The fixture is data only: input sequences and the reference's outputs (score, end cell, alignment offset, CIGAR)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

ACGT = np.frombuffer(b"ACGT", np.uint8)


def mutate(rng, s, n_edits):
    s = list(int(x) for x in s)
    for _ in range(n_edits):
        if not s:
            break
        k = int(rng.integers(0, len(s)))
        op = int(rng.integers(0, 3))
        if op == 0:
            s[k] = int(ACGT[rng.integers(0, 4)])
        elif op == 1:
            s[k:k] = [int(x) for x in ACGT[rng.integers(0, 4, int(rng.integers(1, 12)))]]
        else:
            del s[k:k + int(rng.integers(1, 12))]
    return np.array(s if s else [65], np.uint8)


def cases():
    rng = np.random.default_rng(20261003)
    out = []
    rnd = lambda n: ACGT[rng.integers(0, 4, n)]  # noqa: E731
    # edge shapes
    out += [(b"A", b"A"), (b"A", b"C"), (b"ACGT", b"ACGT"), (b"ACGTACGTAC", b"GTAC"), (b"GTAC", b"ACGTACGTAC"),
            (b"AAAAAAAAAA", b"AAAAA"), (b"AAAAACCCCC", b"AAAAATTCCCCC"), (b"AAAAATTCCCCC", b"AAAAACCCCC"),
            (b"ACACACACACACACAC", b"ACACACAC"), (b"TTTTTTTT", b"AAAAAAAA"), (b"N" * 20, b"N" * 10),
            (b"ACGTNACGT", b"ACGTNACGT")]
    # read vs haplotype (the genotyping call site, alignment_utils.cpp:572-579): a window of the haplotype with edits
    for _ in range(90):
        hap = rnd(int(rng.integers(60, 520)))
        a = int(rng.integers(0, len(hap) - 30))
        b = min(len(hap), a + int(rng.integers(20, 260)))
        out.append((hap, mutate(rng, hap[a:b], int(rng.integers(0, 7)))))
    # haplotype vs padded reference (the assembly call site, hc_assemble_cigar_cacl.c:417-428): similar lengths
    for _ in range(50):
        ref = rnd(int(rng.integers(100, 700)))
        out.append((ref, mutate(rng, ref, int(rng.integers(0, 10)))))
    # unrelated sequences, lengths around the vector widths 8 / 16 / 32
    for n1 in (1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129):
        for n2 in (1, 3, 8, 16, 17, 33, 64, 100):
            out.append((rnd(n1), rnd(n2)))
    # long: up to the 2047 limit and just past it
    big = rnd(2047)
    out += [(big, mutate(rng, big[100:1900], 8)), (mutate(rng, big[:2040], 5), big), (rnd(2047), rnd(40)),
            (rnd(2048), rnd(40)), (rnd(40), rnd(2048)), (rnd(2100), rnd(2100))]
    return [(np.frombuffer(a, np.uint8) if isinstance(a, bytes) else a, np.frombuffer(b, np.uint8) if isinstance(b, bytes) else b)
            for a, b in out]


def main():
    if oracle.sw_ref() is None:
        raise SystemExit("oracle/_ref/libsw_ref.so is not available (needs /root/reference and an AVX-512 host)")
    cs = cases()
    s1 = np.concatenate([a for a, _ in cs])
    s2 = np.concatenate([b for _, b in cs])
    len1 = np.array([len(a) for a, _ in cs], np.uint32)
    len2 = np.array([len(b) for _, b in cs], np.uint32)
    off1 = np.concatenate([[0], np.cumsum(len1)[:-1]]).astype(np.uint32)
    off2 = np.concatenate([[0], np.cumsum(len2)[:-1]]).astype(np.uint32)
    data = dict(seq1=s1, seq2=s2, off1=off1, off2=off2, len1=len1, len2=len2)
    for tag, asm in (("genotype", False), ("assemble", True)):
        rows, cigs, coff = [], [], [0]
        for a, b in cs:
            r512 = oracle.ref_sw_align(a, b, asm, 1)
            r256 = oracle.ref_sw_align(a, b, asm, 2)
            too_long = max(len(a), len(b)) >= oracle.SW_MAX_SEQ_LEN
            if not too_long:
                assert all(np.array_equal(np.asarray(r512[k]), np.asarray(r256[k])) for k in r512), "AVX-512 and AVX2 paths differ"
                rows.append((r512["score"], r512["max_i"], r512["max_j"], r512["alignment_offset"], 0))
                cigs.append(r512["cigar"])
            else:
                assert r512["alignment_offset"] == 1 and r256["alignment_offset"] == 1
                rows.append((0, 0, 0, 1, 1))
                cigs.append(np.zeros(0, np.uint32))
            coff.append(coff[-1] + len(cigs[-1]))
        data[tag + "_result"] = np.array(rows, np.int32)  # score, max_i, max_j, alignment_offset, too_long
        data[tag + "_cigar"] = np.concatenate(cigs).astype(np.uint32)
        data[tag + "_cigar_off"] = np.array(coff, np.uint32)
    p = np.zeros((2, 4), np.int32)
    import ctypes as C
    for k in (0, 1):
        buf = (C.c_int32 * 4)()
        oracle.sw_ref().ref_sw_params(k, buf)
        p[k] = list(buf)
    data["params"] = p  # [genotype, assemble] as the reference's sw_avx_init sets them
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sw_golden.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, len(cs), "cases", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
