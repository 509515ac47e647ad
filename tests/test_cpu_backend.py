"""BASELINE.json config 1 ("1k synthetic pairs, 100x150, fp32, CPU AVX2 path (plumbing, no GPU)") and SURVEY.md 8b ("same
entry point must have a CPU backend"): the C ABI's CPU backend (phmm_create_cpu -- repo-owned scalar and AVX2 + FMA inter-pair
kernels, rovaca_amd/csrc/pairhmm_cpu.cpp) against the reference-generated golden vectors and the oracle.  Runs without a GPU."""
import numpy as np
import pytest

import oracle
from conftest import GOLDEN_TAGS, golden_batch
from rovaca_amd import capi, synth
from rovaca_amd.batch import Batch


@pytest.fixture(scope="module", params=[True, False], ids=["avx2", "scalar"])
def cpu(request):
    e = capi.CpuEngine(n_threads=4, simd=request.param)
    yield e
    e.close()


def test_backend_is_named(cpu):
    assert cpu.backend in ("scalar", "avx2+fma inter-pair (8 pairs per vector)")


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_golden_vectors_bit_exact_on_the_cpu_backend(cpu, golden, tag):
    b = golden_batch(golden, tag)
    raw, l10, st = cpu.compute(b)
    ref = golden[f"{tag}_raw_avx512_s"]  # what the reference's compute_fp_avx512_s returned
    assert np.array_equal(raw.view(np.uint32), ref.view(np.uint32))
    ok = ~np.isnan(golden[f"{tag}_log10_ref"])
    assert np.array_equal(l10[ok], golden[f"{tag}_log10_ref"][ok])
    assert int(st.n_rescued) == int((~ok).sum()) and int(st.cells) == b.cells()


def _against_oracle(cpu, b, **kw):
    raw, l10, st = cpu.compute(b, **kw)
    oraw, ol10, oresc = oracle.batch(b, n_threads=4)
    assert np.array_equal(raw.view(np.uint32), oraw.view(np.uint32))
    assert np.array_equal(l10[~oresc], ol10[~oresc])
    if oresc.any():
        fin = np.isfinite(ol10[oresc])
        assert np.array_equal(np.isfinite(l10[oresc]), fin)
        assert np.array_equal(l10[oresc][fin], ol10[oresc][fin])  # same operation order in double: identical
    assert int(st.n_rescued) == int(oresc.sum())
    return raw, l10


def test_config1_1k_pairs_100x150(cpu):
    b = synth.make_config("c1")
    assert b.n_pairs == 1000
    _against_oracle(cpu, b)


def test_mixed_lengths_rescue_and_odd_symbols(cpu):
    # shapes that do not fill a vector go to the scalar kernel; noisy reads underflow fp32 and are re-run in fp64
    b = synth.make_batch(seed=61, n_regions=3, reads=14, haps=5, read_len=(1, 260), hap_len=(1, 400), n_rate=0.01, hap_n_rate=0.01)
    rng = np.random.default_rng(6)
    odd = np.frombuffer(b"acgtnRYKM-*.", np.uint8)
    idx = np.nonzero(rng.random(b.read_bases.shape[0]) < 0.03)[0]
    b.read_bases[idx] = rng.choice(odd, idx.size)
    idx = np.nonzero(rng.random(b.hap_bases.shape[0]) < 0.02)[0]
    b.hap_bases[idx] = rng.choice(odd, idx.size)
    _against_oracle(cpu, b)
    noisy = synth.make_batch(seed=62, n_regions=2, reads=16, haps=4, read_len=150, hap_len=300, sub_rate=0.2, qual=30)
    raw, _ = _against_oracle(cpu, noisy)
    assert (raw < np.float32(1e-28)).any()
    # general gap parameters (per-base i / d / c, the & 127 masking)
    g = synth.make_batch(seed=63, n_regions=2, reads=8, haps=3, read_len=(30, 90), hap_len=(80, 160))
    g.read_ins = rng.integers(10, 60, g.read_ins.shape[0]).astype(np.uint8)
    g.read_del = rng.integers(10, 60, g.read_ins.shape[0]).astype(np.uint8) | np.uint8(128)
    g.read_gcp = rng.integers(5, 30, g.read_ins.shape[0]).astype(np.uint8)
    _against_oracle(cpu, g)


def test_resident_form_cross_product_and_read_preparation(cpu):
    b = synth.make_batch(seed=64, n_regions=1, reads=12, haps=6, read_len=(40, 120), hap_len=(100, 200))
    raw, l10, _ = cpu.compute(b)
    rb = cpu.upload(b)
    rb.sweep()
    ms, ms32 = rb.sweep_timed(2)
    rraw, rl10, st = rb.download()
    rb.release()
    assert ms > 0 and np.array_equal(rraw.view(np.uint32), raw.view(np.uint32)) and np.array_equal(rl10, l10)
    craw, _, _ = cpu.compute(b, cross=True)  # one region: the pair list is the cross product in (read, hap) order
    assert np.array_equal(craw.view(np.uint32), raw.view(np.uint32))
    # read preparation on the CPU backend = the host logic (H3 + H4), byte for byte what the oracle prepares
    O = oracle.lib()
    mapq = np.random.default_rng(9).choice(np.array([60, 29, 12], np.uint8), b.n_reads)
    rawq = np.random.default_rng(10).choice(np.array([2, 11, 17, 18, 25, 37, 41], np.uint8), b.read_qual.shape[0])
    prepared = Batch(b.read_bases, rawq.copy(), b.read_ins.copy(), b.read_del.copy(), b.read_gcp.copy(), b.read_off, b.read_len,
                     b.hap_bases, b.hap_off, b.hap_len, b.pair_read, b.pair_hap)
    for r in range(b.n_reads):
        o, n = int(b.read_off[r]), int(b.read_len[r])
        q = prepared.read_qual[o:o + n].copy()
        O.phmm_oracle_modify_quals(q, n, int(mapq[r]), 18)
        prepared.read_qual[o:o + n] = q
        g = np.full(n, 45, np.uint8)
        O.phmm_oracle_apply_pcr_error_model(np.ascontiguousarray(b.read_bases[o:o + n]), g, n, 3)
        prepared.read_ins[o:o + n] = g
        prepared.read_del[o:o + n] = g
        prepared.read_gcp[o:o + n] = 10
    unprepared = Batch(b.read_bases, rawq, b.read_ins * 0, b.read_del * 0, b.read_gcp * 0, b.read_off, b.read_len, b.hap_bases,
                       b.hap_off, b.hap_len, b.pair_read, b.pair_hap)
    rb = cpu.upload(unprepared, prep=(mapq, 18, 3))
    q, i, d, c = rb.read_table()
    rb.sweep()
    praw, pl10, _ = rb.download()
    rb.release()
    assert np.array_equal(q, prepared.read_qual) and np.array_equal(i, prepared.read_ins) and np.array_equal(d, prepared.read_del)
    assert np.array_equal(c, prepared.read_gcp)
    eraw, el10, _ = cpu.compute(prepared)
    assert np.array_equal(praw.view(np.uint32), eraw.view(np.uint32)) and np.array_equal(pl10, el10, equal_nan=True)


def test_bad_input_is_refused(cpu):
    b = synth.make_batch(seed=65, n_regions=1, reads=3, haps=2, read_len=20, hap_len=30)
    bad = b.subset([0, 1])
    bad.pair_read = np.array([0, 9], np.uint32)
    with pytest.raises(capi.PhmmError):
        cpu.compute(bad)
    out, _ = cpu.marginalize(np.array([[-1.0, -2.0], [-3.0, -0.5]]), [0, 0], 1)  # the marginalisation entry point works here too
    assert out.tolist() == [[-1.0, -0.5]]


@pytest.mark.gpu
def test_cpu_backend_and_device_agree_bit_for_bit():
    b = synth.make_batch(seed=66, n_regions=4, reads=20, haps=6, read_len=(30, 250), hap_len=(100, 450))
    dev, cpu_e = capi.Engine(0), capi.CpuEngine(n_threads=8)
    draw, dl10, _ = dev.compute(b)
    craw, cl10, _ = cpu_e.compute(b)
    assert np.array_equal(draw.view(np.uint32), craw.view(np.uint32)) and np.array_equal(dl10, cl10, equal_nan=True)
    dev.close()
    cpu_e.close()
