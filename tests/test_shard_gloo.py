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
