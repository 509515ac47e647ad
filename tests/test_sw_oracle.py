"""Smith-Waterman: the CPU restatement (oracle/sw_oracle.c) against golden vectors generated from the
reference's own sw_avx.c, against the reference's object code where it is available, and against hand-derivable answers."""
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sw_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _cases(g):
    for k in range(len(g["len1"])):
        yield k, g["seq1"][g["off1"][k]:g["off1"][k] + g["len1"][k]], g["seq2"][g["off2"][k]:g["off2"][k] + g["len2"][k]]


@pytest.mark.parametrize("tag", ["genotype", "assemble"])
def test_restatement_matches_golden_vectors(gold, tag):
    params = tuple(int(x) for x in gold["params"][0 if tag == "genotype" else 1])
    assert params == (oracle.SW_PARAMS_GENOTYPE if tag == "genotype" else oracle.SW_PARAMS_ASSEMBLE)
    res, cig, coff = gold[tag + "_result"], gold[tag + "_cigar"], gold[tag + "_cigar_off"]
    n_long = 0
    for k, a, b in _cases(gold):
        o = oracle.sw_align(a, b, params)
        if res[k][4]:
            assert o["too_long"] and o["alignment_offset"] == 1  # sw_avx.c:490-495
            n_long += 1
            continue
        assert (o["score"], o["max_i"], o["max_j"], o["alignment_offset"]) == tuple(int(x) for x in res[k][:4]), k
        assert np.array_equal(o["cigar"], cig[coff[k]:coff[k + 1]]), (k, oracle.cigar_string(o["cigar"]))
    assert n_long == 3


def test_known_answers():
    g = oracle.SW_PARAMS_GENOTYPE
    # identical sequences: one match run, offset 0, score = n * match
    o = oracle.sw_align(b"ACGTACGTAC", b"ACGTACGTAC", g)
    assert oracle.cigar_string(o["cigar"]) == "10M" and o["alignment_offset"] == 0 and o["score"] == 100
    # the read is a window of the reference: the offset is where it starts (leading reference bases are free)
    o = oracle.sw_align(b"TTTTTACGTACGGATTACATTTTT", b"ACGTACGGATTACA", g)
    assert oracle.cigar_string(o["cigar"]) == "14M" and o["alignment_offset"] == 5 and o["score"] == 140
    # a 3-base insertion in the read: 10*match*2 + open + 2*extend
    o = oracle.sw_align(b"AACCGGTTAACCGGTTAACC", b"AACCGGTTAAGGGCCGGTTAACC", g)
    assert oracle.cigar_string(o["cigar"]) == "10M3I10M" and o["score"] == 200 - 30 - 2 * 5
    # a 2-base deletion
    o = oracle.sw_align(b"AACCGGTTAACATCGGTTAACC", b"AACCGGTTAACCGGTTAACC", g)
    assert oracle.cigar_string(o["cigar"]) == "11M2D9M" or oracle.cigar_string(o["cigar"]) == "10M2D10M"
    assert o["score"] == 200 - 30 - 5
    # read overhangs are soft-clipped
    o = oracle.sw_align(b"ACGTACGTACGT", b"GGGGGGACGTACGTACGT", g)
    assert oracle.cigar_string(o["cigar"]).endswith("12M") and oracle.cigar_string(o["cigar"]).startswith("6S")


def test_length_guard_and_empty():
    rng = np.random.default_rng(3)
    long = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 2048)]
    o = oracle.sw_align(long, long[:10])
    assert o["too_long"] and o["alignment_offset"] == 1 and len(o["cigar"]) == 0
    o = oracle.sw_align(long[:2047], long[:2047])
    assert not o["too_long"] and oracle.cigar_string(o["cigar"]) == "2047M"
    with pytest.raises(ValueError):
        oracle.sw_align(b"", b"ACGT")


@pytest.mark.skipif(oracle.sw_ref() is None, reason="oracle/_ref/libsw_ref.so not available (needs /root/reference)")
@pytest.mark.parametrize("asm", [False, True])
def test_restatement_matches_reference_object_code(asm):
    rng = np.random.default_rng(11 + asm)
    acgt = np.frombuffer(b"ACGTN", np.uint8)
    params = oracle.SW_PARAMS_ASSEMBLE if asm else oracle.SW_PARAMS_GENOTYPE
    for t in range(150):
        n1, n2 = int(rng.integers(1, 500)), int(rng.integers(1, 300))
        s1 = acgt[rng.integers(0, 5 if t % 7 == 0 else 4, n1)]
        if t % 3 and n1 > 8:
            a = int(rng.integers(0, n1 - 4))
            s2 = s1[a:a + n2].copy()
            for _ in range(int(rng.integers(0, 5))):
                s2[int(rng.integers(0, len(s2)))] = acgt[rng.integers(0, 4)]
            if t % 2:
                k = int(rng.integers(0, len(s2)))
                s2 = np.concatenate([s2[:k], acgt[rng.integers(0, 4, int(rng.integers(1, 20)))], s2[k:]])
        else:
            s2 = acgt[rng.integers(0, 4, n2)]
        o = oracle.sw_align(s1, s2, params)
        for which in (1, 2):  # AVX-512 and AVX2 builds of the reference
            r = oracle.ref_sw_align(s1, s2, asm, which)
            assert (o["score"], o["max_i"], o["max_j"], o["alignment_offset"]) == (r["score"], r["max_i"], r["max_j"], r["alignment_offset"])
            assert np.array_equal(o["cigar"], r["cigar"])
