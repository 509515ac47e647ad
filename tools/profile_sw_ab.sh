#!/bin/bash
# profile_sw_ab.sh [VARIANT_LIB] -- PMC passes over bench.py --config sw for the regular Smith-Waterman library and, if given, a variant
# (tools/build_sw_variant.sh): issue/wait cycles, instruction counts and the clock.  Run on the GPU box through gpurun.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/swprof; rm -rf $OUT; mkdir -p $OUT
run() {  # tag
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $grp --output-format csv -d $OUT/$1_$tag -o pmc -- python3 bench.py --config sw --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$1_$tag.log 2>&1
  done
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$1_clk -o clk -- python3 bench.py --config sw --steps 5 --warmup 1 --no-cpu-baseline > $OUT/$1_clk.log 2>&1
  echo "$1 done"
}
run regular
if [ $# -ge 1 ]; then export ROVACA_AMD_SW_LIB=$(readlink -f $1); run variant; fi
python3 - <<'PY'
import csv, glob, collections
for tag in ("regular", "variant"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/swprof/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sw_kernel<10" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(f"gpurun_out/swprof/{tag}_clk/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sw_kernel<10" in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not acc: continue
    print(tag, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}, "dur_ns", round(sum(dur) / max(1, len(dur))))
PY
