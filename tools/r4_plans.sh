#!/bin/bash
# one-shot call on config 3 (and 2) under different chunk plans (PHMM_CHUNK_WEIGHTS); usage: tools/r4_plans.sh
set -u
cd "$GRAFT_REPO_ROOT"
export PHMM_ENABLE_TEST_HOOKS=1
OUT=gpurun_out/r4_plans; mkdir -p $OUT
for c in c3 c2; do
for w in default 1,3,6,6,6,3 1,2,4,4,4,4,4,2 1,2,4,6,6,4,2 1,4,8,8,4 2,6,14,26,26,18,8; do
  if [ $w = default ]; then unset PHMM_CHUNK_WEIGHTS; else export PHMM_CHUNK_WEIGHTS=$w; fi
  python3 tools/measure_e2e_timeline.py $c > $OUT/${c}_$w.txt 2>&1
  echo "$c $w: $(grep median $OUT/${c}_$w.txt)"
done; done
