#!/bin/bash
# adapter A/B on one box: current host library against rovaca_amd/lib/oldhost (mutex-guarded reservation), mixed and uniform regions
cd "$GRAFT_REPO_ROOT"
run() { # dir mixed threads depth
  local out=$( ( [ "$2" = 1 ] && export ADAPTER_BENCH_MIXED=1; timeout -k 5 120 $1/adapter_bench $3 1000 48 9 0 $4 2>&1 ) | grep -o '"regions_per_s": [0-9]*\|"gcups_end_to_end": [0-9.]*\|"regions_per_submission": [0-9.]*' | tr '\n' ' ')
  echo "$1 mixed=$2 threads=$3 depth=$4: $out"
}
for rep in 1 2; do
for mixed in 1 0; do for depth in 1 4; do
  run rovaca_amd/lib $mixed 30 $depth
  run rovaca_amd/lib/oldhost $mixed 30 $depth
done; done; done
