#!/bin/bash
run() { timeout -k 5 150 python bench.py --config c5 --steps 4 --warmup 2 --no-cpu-baseline --no-e2e 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*\|"pass_kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
for v in rovaca_amd/lib/variants/lib_*.so; do
  for w in 0 1; do
    echo "=== $v waves_forced=$w"
    if [ $w = 1 ]; then ROVACA_AMD_HIP_LIB=$PWD/$v PHMM_FORCE_TEAM_WAVES=1 run; else ROVACA_AMD_HIP_LIB=$PWD/$v run; fi
  done
done
