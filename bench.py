#!/usr/bin/env python3
"""bench.py -- PairHMM GCUPS on N x MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path (fp32 systolic sweep + rescue collection + fp64 rescue) over one synthetic batch of
BASELINE config 2 (1 000 000 pairs, reads 150 bp x haplotypes 300 bp) that is already resident in HBM.  With N > 1 every
rank sweeps its own 1M-pair batch (different seed): the path shards over independent pairs, there is no collective on
the data path, and the reported value is (cells all ranks processed) / (max over ranks of the wall time) -- weak scaling.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector
FLOPS_PER_CELL = 12          # 8 mul + 4 add


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5", "sw"],
                    help="c1..c5: BASELINE configs of the PairHMM path; sw: the Smith-Waterman realignment next to it (SURVEY 8f next-4)")
    ap.add_argument("--sw-regions", type=int, default=1000, help="--config sw: regions of 48 reads (150 bp) against 400-bp haplotypes")
    ap.add_argument("--scale", type=float, default=1.0, help="multiplies the number of regions of the config")
    ap.add_argument("--chunk-regions", type=int, default=1000, help="regions per resident chunk (c4 streams 12.5 chunks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: min(cores, 16))")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: the PairHMM engine has no CPU path")
    device = local_rank % n_dev
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device)
        if n_dev >= world:
            # one rank per GPU; RCCL carries only the barrier and the max-over-ranks of the wall time
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            # rehearsal on a box with fewer GPUs than ranks (ranks share a card): same code path, gloo for the barrier
            dist.init_process_group(backend="gloo")
    else:
        dist = None
        torch.cuda.set_device(device)

    if args.config == "sw":
        return bench_smith_waterman(args, rank, world, device, dist)

    from rovaca_amd import capi, synth

    # each rank owns its own batch (seed offset by rank): the shard of a WGS-scale pair stream this GPU would see
    cfg = dict(synth.CONFIGS[args.config])
    cfg["seed"] = cfg["seed"] + 1000 * rank
    cfg["n_regions"] = max(1, int(round(cfg["n_regions"] * args.scale)))
    # the batch is generated and kept resident in chunks (one upload each); a step sweeps every chunk once
    chunks = []
    left, k = cfg["n_regions"], 0
    while left > 0:
        c = dict(cfg)
        c["n_regions"] = min(left, args.chunk_regions)
        c["seed"] = cfg["seed"] + 7919 * k
        chunks.append(synth.make_batch(**c))
        left -= c["n_regions"]
        k += 1
    batch = chunks[0]
    cells = sum(b.cells() for b in chunks)
    alg_bytes = sum(b.algorithmic_bytes() for b in chunks)
    n_pairs = sum(b.n_pairs for b in chunks)

    eng = capi.Engine(device)
    residents = [eng.upload(b) for b in chunks]

    class _All:  # the step: every resident chunk, in order, on the engine's stream
        def sweep(self):
            for r in residents:
                r.sweep()

        def sweep_timed(self, n):
            a = c = 0.0
            for r in residents:
                x, y = r.sweep_timed(n)
                a += x
                c += y
            return a, c

    rb = _All()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        rb.sweep()
    eng.synchronize()
    barrier()
    t0 = time.perf_counter()
    # exactly `steps` passes; HIP events on the engine's own stream around the launches of every pass, read after the last
    ms, ms32 = rb.sweep_timed(args.steps)
    kern_ms = [ms]
    fp32_ms = [ms32]
    eng.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        on_gpu = dist.get_backend() == "nccl"
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ct = torch.tensor([float(cells)], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)  # every rank generated its own batch: sum the cells actually swept
        total_cells_all = int(ct.item())
    else:
        total_cells_all = cells

    raw, l10, st = residents[0].download()
    n_rescued = int(st.n_rescued) + sum(int(r.download()[2].n_rescued) for r in residents[1:])
    out = None
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_cells = total_cells_all
        gcups = total_cells * args.steps / elapsed / 1e9
        k_ms = float(np.mean(fp32_ms))  # dominant kernel: the fp32 systolic sweep (all launches of one pass)
        achieved_gbs = alg_bytes / (k_ms * 1e-3) / 1e9
        valu_tflops = FLOPS_PER_CELL * cells / (k_ms * 1e-3) / 1e12
        traffic = None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get(args.config if args.scale == 1.0 else "", None)
            except Exception:
                traffic = None
        out = {
            "metric": "PairHMM GCUPS", "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {n_pairs} pairs/GPU, reads {cfg['read_len']} bp x haps "
                                   f"{cfg['hap_len']} bp, fp32 + fp64 rescue, regions of {cfg['reads']} reads x "
                                   f"{cfg['haps']} haps", "pairs_per_gpu": n_pairs, "resident_chunks": len(chunks), "cells_per_gpu": cells,
                       "parallelism": f"pair-split x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_gbs / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": "pairhmm_systolic_kernel<float,K,UG,G>", "kernel_ms": round(k_ms, 4),
                         "algorithmic_bytes": alg_bytes},
            "roofline_valu": {"bound": "fp32-valu", "achieved": round(valu_tflops, 3), "peak": FP32_VALU_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(valu_tflops / FP32_VALU_PEAK_TFLOPS, 5),
                              "kernel_gcups": round(cells / (k_ms * 1e-3) / 1e9, 2), "flops_per_cell": FLOPS_PER_CELL},
            "rescued_pairs": n_rescued, "pass_kernel_ms": round(float(np.mean(kern_ms)), 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            # the first resident chunk (the whole batch for c1/c2/c3) through the CPU path
            out["cpu_baseline"], delta = cpu_baseline(batch, raw, l10, args.cpu_threads)
            out["max_abs_delta_log10_vs_cpu"] = delta
    for r in residents:
        r.release()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def bench_smith_waterman(args, rank, world, device, dist):
    """--config sw: every read of a region against its best haplotype, the genotyping call site of the reference's
    Smith-Waterman (alignment_utils.cpp:572-579).  A step is one swhip_align call on the batch; value = cells all ranks
    aligned / max wall time of the device launches (HIP events inside the call).  The batch is host memory here (the C ABI
    takes host buffers), so the PCIe-inclusive rate of the whole call is reported beside it."""
    import ctypes as C

    import numpy as np
    import torch

    from rovaca_amd import swapi

    rng = np.random.default_rng(5 + 1000 * rank)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    refs, seqs, pairs = [], [], []
    for region in range(args.sw_regions):
        base = acgt[rng.integers(0, 4, 400)]
        haps = []
        for h in range(4):
            hap = base.copy()
            for _ in range(h):
                hap[int(rng.integers(0, len(hap)))] = acgt[rng.integers(0, 4)]
            haps.append(hap)
        h0 = len(refs)
        refs += haps
        for r in range(48):
            hap = haps[r % 4]
            a = int(rng.integers(0, len(hap) - 150 + 1))
            read = hap[a:a + 150].copy()
            read[rng.random(len(read)) < 0.01] = ord("C")
            if r % 6 == 0:
                k = int(rng.integers(1, len(read) - 1))
                read = np.concatenate([read[:k], acgt[rng.integers(0, 4, 3)], read[k:]])
            seqs.append(read)
            pairs.append((h0 + r % 4, len(seqs) - 1))
    al = swapi.Aligner(device)
    for _ in range(max(1, args.warmup)):
        out, cig, st = al.align(refs, seqs, pairs)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_s, call_s = 0.0, 0.0
    for _ in range(args.steps):
        out, cig, st = al.align(refs, seqs, pairs)
        kernel_s += st.kernel_ms * 1e-3
        call_s += st.total_ms * 1e-3
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        t = torch.tensor([kernel_s, call_s], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernel_s, call_s = float(t[0]), float(t[1])
    if rank != 0:
        return
    cells = int(st.cells)
    gcups = world * cells * args.steps / kernel_s / 1e9
    alg_bytes = sum(len(refs[a]) + len(seqs[b]) + 20 for a, b in pairs) + 4 * len(cig)
    res = {
        "metric": "Smith-Waterman GCUPS", "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(kernel_s / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "sw: %d alignments/GPU, reads 150-153 bp against 400-bp haplotypes, genotype parameters "
                               "(10, -15, -30, -5), traceback and CIGAR on the device" % len(pairs),
                   "alignments_per_gpu": len(pairs), "cells_per_gpu": cells, "parallelism": f"alignment-split x{world}, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(alg_bytes * args.steps / kernel_s / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg_bytes * args.steps / kernel_s / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                     "kernel": "swhip::sw_kernel<K,G>", "kernel_ms": round(kernel_s / args.steps * 1e3, 4),
                     "algorithmic_bytes": alg_bytes,
                     "note": "integer-VALU bound (DESIGN.md 7); the backtrack flags (4 bits per cell) are the HBM traffic"},
        "call_gcups_pcie_inclusive": round(world * cells * args.steps / call_s / 1e9, 2),
        "timing": "value: HIP events around the launches inside swhip_align, summed over the steps (the C ABI takes host buffers, "
                  "there is no resident form); call_gcups_pcie_inclusive: wall clock of the same calls",
    }
    if world == 1 and not args.no_cpu_baseline:
        import oracle  # checker + CPU baseline leg only

        for p in rng.choice(len(pairs), 200, replace=False):
            o = oracle.sw_align(refs[pairs[p][0]], seqs[pairs[p][1]])
            assert (int(out[p]["score"]), int(out[p]["alignment_offset"])) == (o["score"], o["alignment_offset"])
            assert np.array_equal(swapi.cigar_of(out[p], cig), o["cigar"])
        res["parity_sample"] = "200 alignments bit-identical to the CPU restatement (score, offset, CIGAR)"
        if oracle.sw_ref() is not None:
            rb, ro, rl = swapi._tables(refs)
            sb, so, sl = swapi._tables(seqs)
            pr = np.ascontiguousarray([p[0] for p in pairs], np.uint32)
            ps = np.ascontiguousarray([p[1] for p in pairs], np.uint32)
            n = min(len(pairs), 20000)
            chk = C.c_int64()
            sec = oracle.sw_ref().ref_sw_batch(0, rb, ro, rl, sb, so, sl, pr[:n].copy(), ps[:n].copy(), n, 1, C.byref(chk))
            ccells = int(sum(int(rl[pr[k]]) * int(sl[ps[k]]) for k in range(n)))
            res["cpu_baseline"] = {"value": round(ccells / sec / 1e9, 3), "unit": "GCUPS", "cores": 1, "kind": "reference",
                                   "sample": "%d alignments through the reference's sw_avx.c (function chosen by sw_avx_init) on "
                                             "one thread, %.2f s" % (n, sec)}
    print(json.dumps(res))


def cpu_baseline(batch, gpu_raw, gpu_l10, threads):
    """Time the CPU path beside the GPU number (checker code, never the product): the reference's own AVX-512
    16-reads x 1-haplotype kernel compiled from /root/reference (oracle/_ref, kind "reference") when the prebuilt
    library is present and the host has AVX-512, else the scalar C restatement (kind "port") on a bounded sample."""
    import numpy as np

    import oracle
    n_cores = os.cpu_count() or 1
    try:
        n_cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # the box's CPU share for one GPU is 16 cores: more threads than that only measures oversubscription
    threads = threads or min(n_cores, 16)
    cells = batch.cells()
    if oracle.ref() is not None:
        raw, sec = oracle.ref_regions_16x1(batch, n_threads=threads)
        same = bool(np.array_equal(raw.view(np.uint32), gpu_raw.view(np.uint32)))
        acc = raw >= np.float32(1e-28)
        # accepted pairs: identical fp32 sums -> identical log10f on the host; report the measured delta anyway
        L = oracle.lib()
        idx = np.nonzero(acc)[0][:: max(1, int(acc.sum()) // 50000)]
        ref_l10 = np.array([L.phmm_oracle_finalize_f32(float(raw[i])) for i in idx])
        delta = float(np.abs(ref_l10 - gpu_l10[idx]).max()) if idx.size else 0.0
        # the same kernel on one thread, on 100 regions of the 150 x 300 shape (a third of a second), for the per-core figure
        from rovaca_amd import synth
        one = synth.make_batch(seed=77, n_regions=100, reads=40, haps=25, read_len=150, hap_len=300)
        _, sec1 = oracle.ref_regions_16x1(one, n_threads=1)
        return ({"value": round(cells / sec / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": "reference",
                 "value_1_thread_150x300": round(one.cells() / sec1 / 1e9, 3),
                 "sample": f"all {batch.n_pairs} pairs of the benchmark batch through the reference's "
                           "compute_full_prob_avx512_float_1xn (16 reads x 1 hap), kernel + lane packing, "
                           f"{threads} std::threads, {sec:.2f} s", "bit_identical_to_gpu_fp32": same}, delta)
    # scalar port on a bounded sample (~10-30 s)
    n = min(batch.n_pairs, max(1000, int(3e9 * threads / 45000 / 10)))
    sub = batch.subset(np.arange(n))
    t0 = time.perf_counter()
    raw, l10, _ = oracle.batch(sub, n_threads=threads)
    sec = time.perf_counter() - t0
    delta = float(np.abs(l10 - gpu_l10[:n]).max())
    return ({"value": round(sub.cells() / sec / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": "port",
             "sample": f"first {n} pairs of the benchmark batch through oracle/phmm_oracle.c (scalar), {sec:.2f} s"}, delta)


if __name__ == "__main__":
    main()
