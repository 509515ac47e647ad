#!/bin/bash
# the stage's fp64 re-run launched by phmm_stage_wait only when a sum underflowed (default) against enqueued behind every sweep
# (PHMM_STAGE_EAGER_RESCUE=1): adapter_bench at 30 workers, synchronous and four in flight, uniform and mixed regions; stage latency
cd "$GRAFT_REPO_ROOT"
export PHMM_ENABLE_TEST_HOOKS=1
one() { timeout -k 5 120 rovaca_amd/lib/adapter_bench 30 2000 48 9 0 $1 2>&1 | grep regions_per_s | sed 's/.*"regions_per_s": \([0-9]*\), "gcups_end_to_end": \([0-9.]*\).*"regions_per_submission": \([0-9.]*\).*/\1 regions\/s \2 GCUPS \3 per submission/'; }
for rep in 1 2 3; do
  for m in deferred eager; do
    if [ $m = eager ]; then export PHMM_STAGE_EAGER_RESCUE=1; else unset PHMM_STAGE_EAGER_RESCUE; fi
    echo "$m sync:      $(one 1)"
    echo "$m depth 4:   $(one 4)"
    echo "$m mixed sync: $(ADAPTER_BENCH_MIXED=1 one 1)"
  done
done
for m in deferred eager; do
  if [ $m = eager ]; then export PHMM_STAGE_EAGER_RESCUE=1; else unset PHMM_STAGE_EAGER_RESCUE; fi
  echo "== stage latency, $m"; timeout -k 5 100 python3 tools/measure_stage_latency.py 2>&1 | tail -8
done
