#!/bin/bash
# time_sw_variants.sh NAME... -- average duration of the Smith-Waterman kernel of bench.py --config sw for the regular library ("regular")
# and variants built by tools/build_sw_variant.sh, from a kernel trace each (under the profiler bench.py skips its parity sample, so
# timing-only ablations that compute wrong results can be timed).  Run on the GPU box through gpurun.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/swvar; mkdir -p $OUT
for v in "$@"; do
  if [ $v = regular ]; then unset ROVACA_AMD_SW_LIB; else export ROVACA_AMD_SW_LIB=$PWD/rovaca_amd/lib/variants/libsw_$v.so; fi
  rm -rf $OUT/$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o t -- python3 bench.py --config sw --steps 10 --warmup 2 --no-cpu-baseline --no-sustained > $OUT/$v.log 2>&1
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(python3 -c "
import csv, sys
for r in csv.DictReader(open('$f')):
    if 'sw_kernel' in r['Name']: print(r['Name'].split('swhip::')[1].split('(')[0], r['Calls'], 'calls', round(float(r['AverageNs']) / 1e3, 1), 'us', end=' | ')
")"
done
