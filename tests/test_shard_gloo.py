"""The N > 1 path on CPU: equal-cells partition of the pair list and the gather by pair id, run as two gloo ranks.  The
per-rank compute is the oracle here (no GPU in this test); on a GPU box the same function is given Engine.compute."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from rovaca_amd import shard, synth


def test_partition_covers_every_pair_once_and_balances_cells():
    b = synth.make_batch(seed=9, n_regions=12, reads=20, haps=6, read_len=(50, 250), hap_len=(100, 500))
    for world in (1, 2, 3, 8):
        parts = shard.partition_equal_cells(b, world)
        allidx = np.concatenate(parts)
        assert len(allidx) == b.n_pairs and np.array_equal(np.sort(allidx), np.arange(b.n_pairs))
        cells = [b.subset(p).cells() for p in parts]
        assert sum(cells) == b.cells()
        assert max(cells) - min(cells) <= 2 * 250 * 500  # each cut is within one pair of the ideal
    e = b.subset(np.zeros(0, np.int64))
    assert [len(p) for p in shard.partition_equal_cells(e, 2)] == [0, 0]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = synth.make_batch(seed=10, n_regions=4, reads=12, haps=5, read_len=(40, 200), hap_len=(100, 300))

    def fn(sub):
        raw, l10, _ = oracle.batch(sub)
        return raw, l10

    raw, l10 = shard.compute_sharded(b, fn, rank, world, dist)
    t = torch.tensor([b.subset(shard.partition_equal_cells(b, world)[rank]).cells()], dtype=torch.float64)
    dist.all_reduce(t)  # the bench's aggregate: cells summed over ranks
    if rank == 0:
        ret["raw"], ret["l10"], ret["cells"] = raw, l10, float(t.item())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    b = synth.make_batch(seed=10, n_regions=4, reads=12, haps=5, read_len=(40, 200), hap_len=(100, 300))
    raw, l10, _ = oracle.batch(b)
    assert np.array_equal(ret["raw"].view(np.uint32), raw.view(np.uint32))
    assert np.array_equal(ret["l10"], l10)
    assert ret["cells"] == b.cells()


def test_c_abi_cut_is_the_python_cut():
    # phmm_create_multi's slices (rovaca_amd/csrc/pairhmm_capi.cpp: partition_equal_cells) against shard.partition_equal_cells:
    # the same order (rows per lane, haplotype length, pair index) and the same cuts, for explicit pair lists and cross products
    from rovaca_amd import capi
    b = synth.make_batch(seed=9, n_regions=12, reads=20, haps=6, read_len=(50, 250), hap_len=(100, 500))
    rng = np.random.default_rng(3)
    shuffled = b.subset(rng.permutation(b.n_pairs))
    for batch in (b, shuffled):
        for world in (1, 2, 3, 8):
            which, cells = capi.plan_slices(batch, world)
            parts = shard.partition_equal_cells(batch, world)
            want = np.zeros(batch.n_pairs, np.uint32)
            for g, idx in enumerate(parts):
                want[idx] = g
            assert np.array_equal(which, want)
            assert [int(c) for c in cells] == [batch.subset(p).cells() for p in parts]
    one = synth.make_batch(seed=10, n_regions=1, reads=30, haps=7, read_len=(40, 200), hap_len=(100, 300))
    which, cells = capi.plan_slices(one, 4, cross=True)  # the region's pair list IS the cross product in (read, hap) order
    which2, cells2 = capi.plan_slices(one, 4)
    assert np.array_equal(which, which2) and np.array_equal(cells, cells2) and int(cells.sum()) == one.cells()


@pytest.mark.parametrize("n_slices", [2, 3])
def test_multi_handle_over_cpu_contexts(n_slices):
    # phmm_create_multi's whole N > 1 path -- the equal-cells cut, one host thread per slice, per-slice tables holding only what the
    # slice refers to, the scatter of the results to the pairs' own positions, the common start gate of the timed sweep -- run over
    # CPU contexts (phmm_create_multi_cpu), against the oracle.  On a GPU box tests/test_gpu_shards.py runs the same with device slices.
    from rovaca_amd import capi
    b = synth.make_batch(seed=9, n_regions=12, reads=20, haps=6, read_len=(50, 250), hap_len=(100, 500), sub_rate=0.05)
    oraw, ol10, oresc = oracle.batch(b, n_threads=8)
    m = capi.MultiEngine(cpu_slices=n_slices, cpu_threads_each=2)
    assert m.n_devices == n_slices and m.devices == [-1 - i for i in range(n_slices)]
    raw, l10, st = m.compute(b)
    assert np.array_equal(raw.view(np.uint32), oraw.view(np.uint32)) and np.array_equal(l10[~oresc], ol10[~oresc])
    assert np.abs(l10[oresc] - ol10[oresc]).max() < 1e-9 and int(st.cells) == b.cells() and int(st.n_rescued) == int(oresc.sum()) > 0
    rb = m.upload(b)
    cells, pairs = rb.shard_cells()
    assert int(cells.sum()) == b.cells() and int(pairs.sum()) == b.n_pairs and (pairs > 0).all()
    assert int(cells.max()) - int(cells.min()) <= 2 * 250 * 500  # each cut within one pair of the ideal
    which, pcells = capi.plan_slices(b, n_slices)
    assert [int(c) for c in pcells] == [int(c) for c in cells]
    rb.sweep()
    wall, dev, _ = rb.sweep_timed(2)
    assert (dev > 0).all() and wall >= dev.max() * 0.5 and wall < 50 * dev.max()  # the gate's wall time is the slowest slice's, not a sum of waits
    rraw, rl10, _ = rb.download()
    assert np.array_equal(rraw.view(np.uint32), oraw.view(np.uint32)) and np.array_equal(rl10, l10)
    rb.release()
    # a shuffled pair list lands every result at its own position again
    perm = np.random.default_rng(4).permutation(b.n_pairs)
    praw, _, _ = m.compute(b.subset(perm))
    assert np.array_equal(praw.view(np.uint32), oraw[perm].view(np.uint32))
    # the cross-product form and the raw-quality form (read preparation inside every slice) go through the same cut
    one = synth.make_batch(seed=10, n_regions=1, reads=30, haps=7, read_len=(40, 200), hap_len=(100, 300))
    craw, _, _ = m.compute(one, cross=True)
    assert np.array_equal(craw.view(np.uint32), oracle.batch(one, n_threads=4)[0].view(np.uint32))
    mapq = np.random.default_rng(5).choice(np.array([60, 29, 12], np.uint8), b.n_reads)
    qraw, _, _ = m.compute(b, prep=(mapq, 18, 3))
    assert np.array_equal(qraw.view(np.uint32), oracle.batch(oracle.prepare_batch(b, mapq, 18, 3), n_threads=8)[0].view(np.uint32))
    # more slices than pairs: the surplus slices are empty and harmless
    tiny = b.subset(np.arange(2))
    traw, _, _ = m.compute(tiny)
    assert np.array_equal(traw.view(np.uint32), oraw[:2].view(np.uint32))
    m.close()
