"""Property tests (hypothesis) of the host logic against the oracle: arbitrary byte strings, not just ACGT."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle
from rovaca_amd import hostapi

seqs = st.binary(min_size=1, max_size=120).map(lambda b: np.frombuffer(b, np.uint8).copy())
dna = st.lists(st.sampled_from(b"ACGTN"), min_size=1, max_size=200).map(lambda l: np.array(l, np.uint8))
repeats = st.tuples(st.lists(st.sampled_from(b"ACGT"), min_size=1, max_size=8), st.integers(2, 30)).map(
    lambda t: np.array(t[0] * t[1], np.uint8))


@settings(max_examples=150, deadline=None)
@given(st.one_of(seqs, dna, repeats), st.integers(0, 3))
def test_pcr_model_equals_oracle_on_any_bytes(s, model):
    H, O = hostapi.lib(), oracle.lib()
    n = len(s)
    a = np.full(n, 45, np.uint8)
    b = a.copy()
    H.rovaca_amd_apply_pcr_error_model(s, a, n, model)
    O.phmm_oracle_apply_pcr_error_model(s, b, n, model)
    assert np.array_equal(a, b)
    assert a[-1] == 45 and np.all(a <= 45)
    if model == 0:
        assert np.all(a == 45)


@settings(max_examples=100, deadline=None)
@given(st.lists(st.integers(0, 255), min_size=1, max_size=300), st.integers(0, 255), st.integers(6, 127))
def test_quality_modification_properties(q, mq, thr):
    H = hostapi.lib()
    a = np.array(q, np.uint8)
    out = a.copy()
    H.rovaca_amd_modify_read_qualities(out, len(out), mq, thr)
    capped = np.minimum(a, mq)
    assert np.array_equal(out, np.where(capped < thr, 6, capped))
