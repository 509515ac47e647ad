#!/bin/bash
# Device time line of the one-shot call on a config (default c3): rocprofv3 kernel trace of tools/measure_e2e_timeline.py, every kernel
# as a line of gpurun_out/<tag>/timeline.tsv.  usage: tools/trace_e2e.sh [config] [tag]
set -u
CFG=${1:-c3}; TAG=${2:-r4_trace_$CFG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/prof -o t -- python3 tools/measure_e2e_timeline.py $CFG > $OUT/run.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python3 tools/kernel_timeline.py $DB > $OUT/timeline.tsv
wc -l $OUT/timeline.tsv; head -3 $OUT/run.log
rm -rf $OUT/prof
