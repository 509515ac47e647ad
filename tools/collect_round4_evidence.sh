#!/bin/bash
# Round-4 evidence of the final build beyond tools/profile_round.sh (GPU box, through gpurun; outputs under gpurun_out/r4_evidence): kernel
# rate per rows-per-lane class, the one-shot call on mid-size batches, stage and one-shot latency of region-sized batches, config 5 by
# shape in both orders of passes, PairHMM + realignment through the adapter.
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r4_evidence; rm -rf $OUT; mkdir -p $OUT
export PHMM_ENABLE_TEST_HOOKS=1
timeout -k 5 300 python3 tools/bench_by_k.py > $OUT/by_k.txt 2>&1
python3 - > $OUT/mid_size_one_shot.txt 2>&1 <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from rovaca_amd import capi, synth
eng = capi.Engine(0)
print("# one-shot phmm_compute (host buffers -> log10), median wall time of 15 calls after 3 warm-ups")
for cfg, kw in (("config 2's pairs", dict(read_len=150, hap_len=300)), ("config 3's lengths", dict(read_len=(50, 250), hap_len=(100, 500)))):
    for n_regions in (45, 100, 200, 300, 500):
        b = synth.make_batch(seed=2, n_regions=n_regions, reads=40, haps=25, **kw)
        out = (np.zeros(b.n_pairs, np.float32), np.zeros(b.n_pairs, np.float64))
        for _ in range(3):
            eng.compute(b, out=out)
        w = []
        for _ in range(15):
            t0 = time.perf_counter(); eng.compute(b, out=out); w.append(time.perf_counter() - t0)
        m = float(np.median(w))
        print(f"{cfg}: {b.n_pairs:7d} pairs  {m * 1e3:7.3f} ms  {b.cells() / m / 1e9:7.0f} GCUPS", flush=True)
PY
timeout -k 5 120 python3 tools/measure_stage_latency.py > $OUT/stage_latency.txt 2>&1
timeout -k 5 120 python3 tools/measure_latency.py > $OUT/oneshot_latency.txt 2>&1
{ echo "# default (fp64 first from a batch's second pass on)"; timeout -k 5 200 python3 tools/bench_c5_shapes.py; echo "# the reference's order of passes forced (PHMM_F64_FIRST=0)"; PHMM_F64_FIRST=0 timeout -k 5 200 python3 tools/bench_c5_shapes.py; } > $OUT/c5_shapes.txt 2>&1
{ for cfg in "1 1" "30 1" "128 1" "1 4" "30 4" "128 4"; do set -- $cfg
    echo "== PairHMM + realignment: $1 workers, $2 in flight"
    timeout -k 5 170 rovaca_amd/lib/adapter_bench $1 1000 48 9 1 $2 2>&1 | grep regions_per_s
  done; } > $OUT/adapter_with_realignment.txt 2>&1
ls $OUT
