#!/bin/bash
# PMC passes over bench.py --config c5 (the band kernels): where the waves of the long-read kernels spend their cycles.
# Run on the GPU box through gpurun; outputs under gpurun_out/prof_c5.  Separate passes, program directly after `--`.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_c5; rm -rf $OUT; mkdir -p $OUT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -o pmc -- python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/pmc_$tag.log 2>&1
  echo "pmc $grp done"
done
