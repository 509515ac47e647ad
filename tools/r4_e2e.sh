#!/bin/bash
# one-shot call on configs 2 and 3: median wall time + host time line (tools/measure_e2e_timeline.py); usage: tools/r4_e2e.sh [tag]
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r4_e2e}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for c in c2 c3; do
  python3 tools/measure_e2e_timeline.py $c > $OUT/$c.txt 2>&1
  grep median $OUT/$c.txt
done
