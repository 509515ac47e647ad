#!/bin/bash
# Round-3 evidence beyond tools/profile_round.sh (run on the GPU box through gpurun; outputs under gpurun_out/r3_evidence):
# the drop-in adapter's throughput and host-time breakdown, the one-shot call's time line, the stage latency table, config 5's shapes.
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r3_evidence; rm -rf $OUT; mkdir -p $OUT
{ echo "nproc $(nproc); cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null); $(grep -m1 'model name' /proc/cpuinfo)"
  for d in 1 4; do for th in 1 8 30 64 128; do
    n=$([ $th -le 8 ] && echo 4000 || echo 2000)
    echo "== PairHMM only: $th workers, $d region(s) in flight per worker, $n regions per worker"
    ADAPTER_BENCH_PHASES=1 timeout -k 5 170 rovaca_amd/lib/adapter_bench $th $n 48 9 0 $d 2>&1
  done; done
  for d in 1 4; do for th in 1 30 128; do
    echo "== PairHMM + realignment: $th workers, $d in flight"
    timeout -k 5 170 rovaca_amd/lib/adapter_bench $th 1000 48 9 1 $d 2>&1
  done; done
  for sw in 0 1; do for d in 1 4; do for th in 1 8 30 64; do
    echo "== regions of mixed shape (ADAPTER_BENCH_MIXED=1), realignment $sw: $th workers, $d in flight"
    ADAPTER_BENCH_MIXED=1 timeout -k 5 170 rovaca_amd/lib/adapter_bench $th 1000 48 9 $sw $d 2>&1
  done; done; done; } > $OUT/adapter.txt 2>&1
{ for cfg in "1 1" "30 1" "30 4"; do set -- $cfg
    echo "== PairHMM + realignment with the host phases of both seams: $1 workers, $2 in flight"
    ADAPTER_BENCH_PHASES=1 timeout -k 5 170 rovaca_amd/lib/adapter_bench $1 1000 48 9 1 $2 2>&1
  done; } > $OUT/sw_host_profile.txt 2>&1
python tools/measure_stage_latency.py > $OUT/stage_latency.txt 2>&1
python tools/measure_e2e_timeline.py > $OUT/e2e_timeline.txt 2>&1
python tools/bench_c5_shapes.py > $OUT/c5_shapes.txt 2>&1
python tools/measure_latency.py > $OUT/oneshot_latency.txt 2>&1
for sc in 1.0 0.26 0.14 0.06; do python bench.py --scale $sc --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-adapter 2>/dev/null | cut -c1-400; done > $OUT/launch_size.txt 2>&1
python bench.py --gpus 4 --oversubscribe --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_gpus4_rehearsal.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_adapter -o adapter -- $GRAFT_REPO_ROOT/rovaca_amd/lib/adapter_bench 30 500 48 9 0 1 > $GRAFT_REPO_ROOT/$OUT/trace_adapter.log 2>&1
cd $GRAFT_REPO_ROOT; ls $OUT
