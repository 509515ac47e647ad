"""The sharded path with the HIP engine as the per-slice compute, the multi-device handle of the C ABI
(include/pairhmm_hip.h: phmm_create_multi), BASELINE config 4's chunked resident form and the corner of config 5.

The sharded result must be the unsharded one BIT FOR BIT: a pair's value does not depend on which pairs travel with it
(H6 note of SURVEY.md 8a; the reference's 16-lane grouping is an AVX artefact)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from rovaca_amd import capi, shard, synth
from rovaca_amd.batch import Batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def engine():
    e = capi.Engine(0)
    yield e
    e.close()


def _mixed_batch():
    return synth.make_batch(seed=41, n_regions=30, reads=24, haps=7, read_len=(40, 250), hap_len=(100, 500))


def test_compute_sharded_world1_with_the_hip_engine(engine):
    b = _mixed_batch()
    raw, l10, _ = engine.compute(b)

    def fn(sub):
        r, l, _ = engine.compute(sub)
        return r, l

    idx, sraw, sl10 = shard.compute_sharded(b, fn, 0, 1)
    assert np.array_equal(np.sort(idx), np.arange(b.n_pairs))
    assert np.array_equal(sraw.view(np.uint32), raw[idx].view(np.uint32)) and np.array_equal(sl10, l10[idx])
    # every slice of a three-way cut, computed one after the other on this card, lands where the unsharded run put it
    out = np.zeros(b.n_pairs, np.float32)
    for rank in range(3):
        i, r, _ = shard.compute_sharded(b, fn, rank, 3)
        out[i] = r
    assert np.array_equal(out.view(np.uint32), raw.view(np.uint32))


_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
from rovaca_amd import capi, shard, synth
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)  # the gather only; the data path has no collective
b = synth.make_batch(seed=41, n_regions=30, reads=24, haps=7, read_len=(40, 250), hap_len=(100, 500))
eng = capi.Engine(0)  # both ranks share the one card of the test box
def fn(sub):
    r, l, _ = eng.compute(sub)
    return r, l
raw, l10 = shard.compute_sharded(b, fn, rank, world, dist)
if rank == 0:
    np.savez(sys.argv[4], raw=raw, l10=l10)
dist.barrier()
eng.close()
dist.destroy_process_group()
"""


def test_compute_sharded_two_ranks_sharing_the_card(engine, tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT))
    out = tmp_path / "gathered.npz"
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port), str(out)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    b = _mixed_batch()
    raw, l10, _ = engine.compute(b)
    assert np.array_equal(got["raw"].view(np.uint32), raw.view(np.uint32))
    assert np.array_equal(got["l10"], l10)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_multi_device_handle(engine, n, monkeypatch):
    # the C-ABI form of the same split (one host thread and one context per slice).  On a box with fewer than n GPUs the slices
    # share the card (PHMM_MULTI_ALLOW_SHARED_DEVICE, a hook for exactly this): n contexts, n host threads, n slices of the cut,
    # the common start gate of the timed sweep, the scatter to the pairs' own positions -- everything an 8-GPU run does but the PCIe
    n_dev = capi.lib().phmm_device_count()
    devices = list(range(n)) if n <= n_dev else [i % n_dev for i in range(n)]
    if n > n_dev:
        monkeypatch.setenv("PHMM_MULTI_ALLOW_SHARED_DEVICE", "1")
    b = _mixed_batch()
    raw, l10, st = engine.compute(b)
    m = capi.MultiEngine(devices)
    assert m.n_devices == n and m.devices == devices
    mraw, ml10, mst = m.compute(b)
    assert np.array_equal(mraw.view(np.uint32), raw.view(np.uint32)) and np.array_equal(ml10, l10)
    assert int(mst.cells) == b.cells() and int(mst.n_rescued) == int(st.n_rescued)
    rb = m.upload(b)
    cells, pairs = rb.shard_cells()
    assert int(cells.sum()) == b.cells() and int(pairs.sum()) == b.n_pairs and (pairs > 0).all()
    assert int(cells.max()) - int(cells.min()) <= 2 * 250 * 500  # each cut within one pair of the ideal
    which, pcells = capi.plan_slices(b, n)
    assert [int(c) for c in pcells] == [int(c) for c in cells]
    rb.sweep()
    wall, dev, f32 = rb.sweep_timed(2)
    assert wall > 0 and (dev > 0).all() and (f32 > 0).all() and wall < 100 * dev.max()
    rraw, rl10, _ = rb.download()
    assert np.array_equal(rraw.view(np.uint32), raw.view(np.uint32)) and np.array_equal(rl10, l10)
    rb.release()
    # a shuffled pair list: results land at the pairs' own positions
    perm = np.random.default_rng(6).permutation(b.n_pairs)
    praw, _, _ = m.compute(b.subset(perm))
    assert np.array_equal(praw.view(np.uint32), raw[perm].view(np.uint32))
    # the full cross product form (pair arrays NULL) goes through the same cut
    cross = Batch(b.read_bases, b.read_qual, b.read_ins, b.read_del, b.read_gcp, b.read_off[:40], b.read_len[:40],
                  b.hap_bases, b.hap_off[:9], b.hap_len[:9], np.repeat(np.arange(40), 9), np.tile(np.arange(9), 40))
    craw, _, _ = m.compute(cross, cross=True)
    eraw, _, _ = engine.compute(cross)
    assert np.array_equal(craw.view(np.uint32), eraw.view(np.uint32))
    # raw qualities: every slice prepares its own reads on its device
    mapq = np.random.default_rng(7).choice(np.array([60, 29, 12], np.uint8), b.n_reads)
    qraw, ql10, _ = m.compute(b, prep=(mapq, 18, 3))
    eraw, el10, _ = engine.compute(b, prep=(mapq, 18, 3))
    assert np.array_equal(qraw.view(np.uint32), eraw.view(np.uint32)) and np.array_equal(ql10, el10, equal_nan=True)
    m.close()


def test_multi_device_handle_rejects_bad_device_lists(monkeypatch):
    n_dev = capi.lib().phmm_device_count()
    monkeypatch.delenv("PHMM_MULTI_ALLOW_SHARED_DEVICE", raising=False)
    with pytest.raises(capi.PhmmError):
        capi.MultiEngine([0, 0])  # one context per device (without the rehearsal hook)
    with pytest.raises(capi.PhmmError):
        capi.MultiEngine([n_dev])  # no such device


def test_bench_refuses_more_gpus_than_there_are():
    n_dev = capi.lib().phmm_device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_dev + 1), "--steps", "1", "--warmup", "0",
                        "--scale", "0.01", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_dev + 1), "--oversubscribe", "--steps", "2",
                        "--warmup", "1", "--scale", "0.02", "--cpu-threads", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1  # one line, nothing else on stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    # the N-slice line carries what the one-GPU line carries: the CPU leg with its bit check, the PCIe-inclusive one-shot call through
    # phmm_multi_compute, and the drop-in adapter over N batchers (one per slice's device, dealt round-robin)
    assert line["cpu_baseline"]["bit_identical_to_gpu_fp32"] is True and line["max_abs_delta_log10_vs_cpu"] == 0.0
    assert line["e2e"]["identical_to_resident_path"] is True and line["value_e2e_pcie_inclusive"] > 0
    ad = line["adapter"]
    assert "error" not in ad, ad
    assert ad["bit_identical_to_resident_path"] is True and ad["value"] > 0 and ad["synchronous_seam"]["value"] > 0
    assert len(ad["batchers"]) == n_dev + 1 and all(b["regions"] > 0 and b["submissions"] > 0 for b in ad["batchers"])
    per = [b["regions"] for b in ad["batchers"]]
    assert max(per) - min(per) <= max(4, sum(per) // 100)
    # the rehearsal runs n_dev + 1 SLICES (the whole N-slice path) on the devices there are and says so: n_gpus stays physical
    assert line["n_gpus"] == n_dev and line["metric"] == "PairHMM GCUPS" and line["value"] > 0
    assert line["config"]["slices"] == n_dev + 1 and len(line["per_device_ms_per_step"]) == n_dev + 1 and "note" in line
    assert line["roofline_valu"]["frac"] > 0 and line["imbalance_max_over_mean"] >= 1.0 and "traffic_source" in line["roofline"]
    assert sum(line["config"]["shard_cells"]) == line["config"]["cells_per_gpu"] * line["config"]["slices"]


def test_config4_chunked_resident_batch(engine, monkeypatch):
    # BASELINE config 4 as bench.py --config c4 runs it on one GPU: the pair stream is kept resident in chunks (one upload
    # each), a step sweeps every chunk.  Two chunks of config 4's length distribution here; a seeded sample of every chunk
    # against the oracle, and the order of the pair list must not matter.
    chunks = [synth.make_batch(**dict(synth.CONFIGS["c4"], n_regions=120, seed=4 + 7919 * k)) for k in range(2)]
    residents = [engine.upload(b) for b in chunks]
    for _ in range(2):  # two steps: results of a pass do not depend on the one before
        for r in residents:
            r.sweep()
    rng = np.random.default_rng(44)
    for b, r in zip(chunks, residents):
        raw, l10, st = r.download()
        assert int(st.cells) == b.cells()
        idx = np.sort(rng.choice(b.n_pairs, 6000, replace=False))
        oraw, ol10, oresc = oracle.batch(b.subset(idx), n_threads=8)
        assert np.array_equal(raw[idx].view(np.uint32), oraw.view(np.uint32))
        assert np.array_equal(l10[idx][~oresc], ol10[~oresc])
        perm = rng.permutation(b.n_pairs)
        rp = engine.upload(b.subset(perm))
        rp.sweep()
        praw, pl10, _ = rp.download()
        rp.release()
        assert np.array_equal(praw.view(np.uint32), raw[perm].view(np.uint32)) and np.array_equal(pl10, l10[perm], equal_nan=True)
        r.release()
    # the same stream cut across slices instead of chunks: as many devices as there are, and two and three slices whatever there is
    eraw, el10, _ = engine.compute(chunks[0])
    n_dev = capi.lib().phmm_device_count()
    monkeypatch.setenv("PHMM_MULTI_ALLOW_SHARED_DEVICE", "1")
    for n in sorted({n_dev, 2, 3}):
        m = capi.MultiEngine([i % n_dev for i in range(n)])
        mraw, ml10, _ = m.compute(chunks[0])
        assert np.array_equal(mraw.view(np.uint32), eraw.view(np.uint32)) and np.array_equal(ml10, el10, equal_nan=True)
        rb = m.upload(chunks[0])
        cells, _ = rb.shard_cells()
        assert int(cells.sum()) == chunks[0].cells() and int(cells.max()) - int(cells.min()) <= 2 * 250 * 500
        rb.sweep()
        rraw, _, _ = rb.download()
        assert np.array_equal(rraw.view(np.uint32), eraw.view(np.uint32))
        rb.release()
        m.close()


def _long_pair(rng, R, H, sub_rate=0.01):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, H)]
    a = int(rng.integers(0, H - R + 1))
    seq = hap[a:a + R].copy()
    sub = rng.random(R) < sub_rate
    seq[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
    q = rng.choice(np.array([6, 25, 37], np.uint8), R, p=[0.05, 0.15, 0.8])
    g = np.full(R, 40, np.uint8)
    return (seq, q, g, g, np.full(R, 10, np.uint8)), hap


def test_config5_corner_10k_by_12k(engine, monkeypatch):
    # the corner of BASELINE config 5: the longest read against the longest haplotype (largest carry slab, largest LDS plane
    # staging, the dynamic-LDS attribute path) and a short read against it.  The reference sizes its row buffers as VLAs on
    # the stack for this (core/avx512_s.cpp:269-277); here the carry lives in HBM.
    rng = np.random.default_rng(55)
    reads, haps = [], []
    for R, H in ((10000, 12000), (2000, 12000)):
        rd, hp = _long_pair(rng, R, H)
        reads.append(rd)
        haps.append(hp)
    b = Batch.from_lists(reads, haps, [(0, 0), (1, 1), (1, 0)])
    raw, l10, st = engine.compute(b)
    oraw, ol10, oresc = oracle.batch(b, n_threads=3)
    assert np.array_equal(raw.view(np.uint32), oraw.view(np.uint32))
    assert np.array_equal(np.isfinite(l10), np.isfinite(ol10))
    fin = np.isfinite(ol10)
    assert np.array_equal(l10[fin], ol10[fin])
    assert int(st.n_rescued) == int(oresc.sum())
    # the same with the fp64 kernel first (what the library does by default once such pairs have been seen to underflow)
    monkeypatch.setenv("PHMM_F64_FIRST", "1")
    fraw, fl10, fst = engine.compute(b)
    assert np.array_equal(fl10, l10, equal_nan=True) and int(fst.n_rescued) == int(st.n_rescued) and (fraw < np.float32(1e-28)).all()


def test_config5_shape_under_the_library_default_order_of_passes(engine, monkeypatch):
    # No hook at all (the suite otherwise pins PHMM_F64_FIRST=0): a resident batch of long pairs that all underflow fp32 is swept fp32
    # first on its first pass and fp64 first from the second pass on (pairhmm_sweep.cpp, "fp64 first").  The log10 values and the rescue
    # decisions must be the same on every pass and equal to the oracle's; the fp32 sums of the later passes are 0 where the library
    # proved the underflow (include/pairhmm_hip.h).
    monkeypatch.delenv("PHMM_F64_FIRST", raising=False)
    rng = np.random.default_rng(56)
    reads, haps = [], []
    for R, H in ((2100, 2600), (2500, 3000), (2300, 4000)):
        rd, hp = _long_pair(rng, R, H)
        reads.append(rd)
        haps.append(hp)
    b = Batch.from_lists(reads, haps, [(0, 0), (1, 1), (2, 2), (0, 2), (1, 0)])
    oraw, ol10, oresc = oracle.batch(b, n_threads=5)
    assert oresc.all()
    eng2 = capi.Engine(0)  # a fresh context: no history
    rb = eng2.upload(b)
    passes = []
    for _ in range(3):
        rb.sweep()
        passes.append(rb.download())
    rb.release()
    eng2.close()
    raw1, l10_1, st1 = passes[0]
    assert np.array_equal(raw1.view(np.uint32), oraw.view(np.uint32))  # first pass: the reference's order, every fp32 sum computed
    for raw, l10, st in passes:
        fin = np.isfinite(ol10)
        assert np.array_equal(np.isfinite(l10), fin) and np.abs(l10[fin] - ol10[fin]).max() <= 1e-9
        assert np.array_equal(l10, l10_1, equal_nan=True) and int(st.n_rescued) == int(oresc.sum())
        assert ((raw.view(np.uint32) == oraw.view(np.uint32)) | (raw == 0)).all()


def test_config5_full_size_properties(engine, monkeypatch):
    # all 1 000 pairs of config 5 (the oracle would need hours): every result finite or -inf and never NaN, the same bits in
    # any pair order, and the same bits with the zero-superstep shortcut switched off
    b = synth.make_config("c5")
    assert b.n_pairs == 1000 and int(b.read_len.max()) > 9000 and int(b.hap_len.max()) > 11000
    raw, l10, st = engine.compute(b)
    assert not np.isnan(l10).any() and (np.isfinite(l10) | (l10 == -np.inf)).all()
    assert (l10[np.isfinite(l10)] < 0).all()
    assert int(st.cells) == b.cells()
    perm = np.random.default_rng(5).permutation(b.n_pairs)
    praw, pl10, _ = engine.compute(b.subset(perm))
    assert np.array_equal(praw.view(np.uint32), raw[perm].view(np.uint32)) and np.array_equal(pl10, l10[perm])
    # the three size ranges of this batch all get four waves of 64 lanes and are therefore one launch, longest pair first; as three
    # launches (the A/B hook) the bits are the same
    monkeypatch.setenv("PHMM_NO_STRIPED_MERGE", "1")
    sraw, sl10, sst = engine.compute(b)
    monkeypatch.delenv("PHMM_NO_STRIPED_MERGE")
    assert np.array_equal(sraw.view(np.uint32), raw.view(np.uint32)) and np.array_equal(sl10, l10) and int(sst.n_launches) > int(st.n_launches)
    monkeypatch.setenv("PHMM_NO_EARLY_EXIT", "1")
    sub = np.arange(0, b.n_pairs, 4)  # the unshortened sweep is several times slower: a quarter of the pairs
    nraw, nl10, _ = engine.compute(b.subset(sub))
    assert np.array_equal(nraw.view(np.uint32), raw[sub].view(np.uint32)) and np.array_equal(nl10, l10[sub])


@pytest.mark.parametrize("n_regions,one_lane", [(41, False), (75, False), (130, True), (330, False)])
def test_mid_size_one_shot_calls_take_the_chunk_pipeline_too(engine, n_regions, one_lane, monkeypatch):
    # from 40 000 pairs on a one-shot call is cut into four to twelve chunks that alternate between two sweep lanes (two stream sets of
    # the context; PHMM_CHUNK_ONE_LANE keeps one): same bits as one upload + one sweep and as the oracle, with a noisy batch that sends
    # pairs through the fp64 re-run of both lanes, and called twice through the same context (slots and lanes are reused)
    b = synth.make_batch(seed=700 + n_regions, n_regions=n_regions, reads=40, haps=25, read_len=(60, 200), hap_len=(150, 400),
                         sub_rate=0.06 if n_regions == 75 else 0.01, qual=30 if n_regions == 75 else "illumina")
    assert b.n_pairs >= 40000
    if one_lane:
        monkeypatch.setenv("PHMM_CHUNK_ONE_LANE", "1")
    raw, l10, st = engine.compute(b)
    raw2, l10_2, _ = engine.compute(b)
    assert np.array_equal(raw.view(np.uint32), raw2.view(np.uint32)) and np.array_equal(l10, l10_2, equal_nan=True)
    monkeypatch.setenv("PHMM_NO_CHUNKED_COMPUTE", "1")
    praw, pl10, pst = engine.compute(b)
    monkeypatch.delenv("PHMM_NO_CHUNKED_COMPUTE")
    assert np.array_equal(raw.view(np.uint32), praw.view(np.uint32)) and np.array_equal(l10, pl10, equal_nan=True)
    assert int(st.n_rescued) == int(pst.n_rescued) and int(st.cells) == b.cells()
    if n_regions == 75:
        assert int(st.n_rescued) > 100
    idx = np.sort(np.random.default_rng(n_regions).choice(b.n_pairs, 2000, replace=False))
    oraw, ol10, oresc = oracle.batch(b.subset(idx), n_threads=8)
    assert np.array_equal(raw[idx].view(np.uint32), oraw.view(np.uint32)) and np.array_equal(l10[idx][~oresc], ol10[~oresc])
    if oresc.any():
        assert np.abs(l10[idx][oresc] - ol10[oresc]).max() < 1e-9


def test_chunked_one_shot_compute_is_the_plain_one(engine, monkeypatch):
    # phmm_compute cuts a big batch into chunks (binning, packing and upload of chunk k + 1 under the sweep of chunk k, host
    # log10 of chunk k - 1 meanwhile): same bits as one upload + one sweep, for an explicit pair list, for the full cross
    # product, with device-side read preparation, and for a pair list that jumps around the tables (no chunking then)
    b = synth.make_batch(seed=52, n_regions=560, reads=40, haps=25, read_len=(60, 200), hap_len=(150, 400))
    assert b.n_pairs >= 1 << 19  # twelve chunks
    raw, l10, st = engine.compute(b)
    assert int(st.cells) == b.cells()
    monkeypatch.setenv("PHMM_NO_CHUNKED_COMPUTE", "1")
    praw, pl10, pst = engine.compute(b)
    monkeypatch.delenv("PHMM_NO_CHUNKED_COMPUTE")
    assert np.array_equal(raw.view(np.uint32), praw.view(np.uint32)) and np.array_equal(l10, pl10, equal_nan=True)
    assert int(st.n_rescued) == int(pst.n_rescued)
    idx = np.sort(np.random.default_rng(1).choice(b.n_pairs, 4000, replace=False))
    oraw, ol10, oresc = oracle.batch(b.subset(idx), n_threads=8)
    assert np.array_equal(raw[idx].view(np.uint32), oraw.view(np.uint32)) and np.array_equal(l10[idx][~oresc], ol10[~oresc])
    # shuffled pair list: every chunk would need the whole tables -> falls back to one upload; still the same numbers
    perm = np.random.default_rng(2).permutation(b.n_pairs)
    sraw, _, _ = engine.compute(b.subset(perm))
    assert np.array_equal(sraw.view(np.uint32), raw[perm].view(np.uint32))
    # full cross product of a read table and a haplotype table (pair arrays NULL): chunked over the reads
    cross = synth.make_batch(seed=53, n_regions=1, reads=9000, haps=64, read_len=(50, 120), hap_len=200)
    craw, cl10, _ = engine.compute(cross, cross=True)
    monkeypatch.setenv("PHMM_NO_CHUNKED_COMPUTE", "1")
    draw, dl10, _ = engine.compute(cross, cross=True)
    monkeypatch.delenv("PHMM_NO_CHUNKED_COMPUTE")
    assert cross.n_pairs >= 1 << 19 and np.array_equal(craw.view(np.uint32), draw.view(np.uint32)) and np.array_equal(cl10, dl10)
    # device-side read preparation travels with the chunks (mapping qualities are windowed like the read table)
    mapq = np.random.default_rng(3).choice(np.array([60, 29, 12], np.uint8), b.n_reads)
    qraw, ql10, _ = engine.compute(b, prep=(mapq, 18, 3))
    monkeypatch.setenv("PHMM_NO_CHUNKED_COMPUTE", "1")
    rraw, rl10, _ = engine.compute(b, prep=(mapq, 18, 3))
    assert np.array_equal(qraw.view(np.uint32), rraw.view(np.uint32)) and np.array_equal(ql10, rl10, equal_nan=True)
