#!/bin/bash
# A/B of the one-shot call's two-phase form (long-read classes swept after the short-read ones, compute_chunked) against the one-phase
# form (PHMM_CHUNK_ONE_PHASE=1) on configs 3 and 2, alternating on one box.  usage: tools/r4_two_phase.sh [tag]
set -u
cd "$GRAFT_REPO_ROOT"
export PHMM_ENABLE_TEST_HOOKS=1
TAG=${1:-r4_two_phase}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  for mode in two one; do
    if [ $mode = one ]; then export PHMM_CHUNK_ONE_PHASE=1; else unset PHMM_CHUNK_ONE_PHASE; fi
    for c in c3 c2; do
      python3 tools/measure_e2e_timeline.py $c > $OUT/${c}_${mode}_$rep.txt 2>&1
      echo "$c $mode-phase #$rep: $(grep median $OUT/${c}_${mode}_$rep.txt)"
    done
  done
done
