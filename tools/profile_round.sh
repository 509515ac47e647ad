#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box through gpurun; outputs under gpurun_out/prof).
# Kernel trace and PMC counters are separate runs; the program follows `--` directly (no wrappers).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
# profile_round.sh sw: only the Smith-Waterman part (into the same directory, next to an earlier full run's files)
if [ "${1:-all}" != sw ]; then
rm -rf $OUT; mkdir -p $OUT
for c in c2 c3 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$c -o $c -- python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-adapter --no-configs > $OUT/bench_$c.log 2>&1
  tail -1 $OUT/bench_$c.log | cut -c1-160
done
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -o pmc -- python3 bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-adapter --no-configs > $OUT/pmc_$tag.log 2>&1
  echo "pmc $grp done"
done
# the clock the chip holds under the hot kernel (MI355X_MICROARCH.md, DVFS give-back: GRBM_GUI_ACTIVE / 8 / kernel wall time), with and
# without the table sweep: one counter next to the kernel trace that gives each dispatch's duration
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/clk_c2 -o clk -- python3 bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-adapter --no-configs > $OUT/clk_c2.log 2>&1
PHMM_ENABLE_TEST_HOOKS=1 PHMM_NO_TAB=1 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/clk_c2_notab -o clk -- python3 bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-adapter --no-configs > $OUT/clk_c2_notab.log 2>&1
echo "clock passes done"
# the same batch through the general kernel (per-base gap-opens: what a run under the default PCR model takes)
export PHMM_ENABLE_TEST_HOOKS=1 PHMM_NO_UNIFORM_GOP=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmcpcr_$tag -o pmc -- python3 bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-adapter --no-configs > $OUT/pmcpcr_$tag.log 2>&1
  echo "pmc (general kernel) $grp done"
done
unset PHMM_NO_UNIFORM_GOP PHMM_ENABLE_TEST_HOOKS
# configs 3 and 5: instruction counts and traffic per pass (bench.py --config c3 / c5 price their lines with them).  Config 5 with the
# reference's order of passes forced (PHMM_F64_FIRST=0: fp32 kernel, then fp64 for what underflowed -- every pair), so that "per pass"
# means one real launch of each kernel; by default the passes after the first run the fp64 kernel first and an empty fp32 launch (the
# kernel traces above are of the default).  The fp64 kernel does the same work either way.
export PHMM_ENABLE_TEST_HOOKS=1 PHMM_F64_FIRST=0
for c in c3 c5; do
for grp in "SQ_INSTS_VALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc${c}_$tag -o pmc -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-adapter --no-configs > $OUT/pmc${c}_$tag.log 2>&1
  echo "pmc $c $grp done"
done; done
unset PHMM_ENABLE_TEST_HOOKS PHMM_F64_FIRST
find $OUT -name "*.csv" | head -30
fi
mkdir -p $OUT; rm -rf $OUT/trace_sw $OUT/pmcsw_*
# Smith-Waterman aligner: kernel trace of bench.py --config sw (without the leg of four concurrent callers, whose
# overlapping launches would blur the lone kernel's duration)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sw -o sw -- python3 bench.py --config sw --no-cpu-baseline --no-sustained > $OUT/bench_sw.log 2>&1
tail -1 $OUT/bench_sw.log | cut -c1-200
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmcsw_$tag -o pmc -- python3 bench.py --config sw --steps 2 --warmup 1 --no-cpu-baseline --no-sustained > $OUT/pmcsw_$tag.log 2>&1
done
echo "pmc sw done"
