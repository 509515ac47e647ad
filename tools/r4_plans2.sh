#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export PHMM_ENABLE_TEST_HOOKS=1
OUT=gpurun_out/r4_plans2; mkdir -p $OUT
for t in 3 5; do
for w in default 2,3,4,5,6,6,6,6,5,4,3 1,2,4,6,6,6,6,6,5,4,3 1,2,3,4,6,8,8,8,6,4 1,1,2,3,4,5,6,6,6,6,5,4,3; do
  export PHMM_CHUNK_PREP_THREADS=$t
  if [ $w = default ]; then unset PHMM_CHUNK_WEIGHTS; else export PHMM_CHUNK_WEIGHTS=$w; fi
  python3 tools/measure_e2e_timeline.py c3 > $OUT/c3_${t}_$w.txt 2>&1
  echo "c3 prep=$t $w: $(grep median $OUT/c3_${t}_$w.txt)"
done; done
