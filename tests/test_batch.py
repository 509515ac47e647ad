"""Batch.concat (bench.py joins generated pieces into resident chunks with it): the joined batch is the parts laid end to end."""
import numpy as np

import oracle
from rovaca_amd import synth
from rovaca_amd.batch import Batch


def test_concat_is_the_parts_end_to_end():
    parts = [synth.make_batch(seed=5, n_regions=3, reads=4, haps=3, read_len=(20, 60), hap_len=(70, 90)),
             synth.make_batch(seed=6, n_regions=2, reads=5, haps=2, read_len=30, hap_len=80),
             synth.make_batch(seed=7, n_regions=1, reads=2, haps=4, read_len=(10, 40), hap_len=50)]
    whole = Batch.concat(parts)
    assert whole.n_pairs == sum(p.n_pairs for p in parts) and whole.n_reads == sum(p.n_reads for p in parts)
    assert whole.cells() == sum(p.cells() for p in parts) and whole.algorithmic_bytes() == sum(p.algorithmic_bytes() for p in parts)
    # every pair still points at its own read and haplotype bytes
    at = 0
    for p in parts:
        for k in range(p.n_pairs):
            r, h = int(whole.pair_read[at + k]), int(whole.pair_hap[at + k])
            assert all(np.array_equal(x, y) for x, y in zip(whole.read(r), p.read(int(p.pair_read[k]))))
            assert np.array_equal(whole.hap(h), p.hap(int(p.pair_hap[k])))
        at += p.n_pairs
    assert np.array_equal(whole.region_pair, np.concatenate([[0], np.cumsum([q for p in parts for q in np.diff(p.region_pair)])]))
    got = oracle.batch(whole)[0]
    want = np.concatenate([oracle.batch(p)[0] for p in parts])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert Batch.concat([parts[0]]) is parts[0]
