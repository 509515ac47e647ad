#!/bin/bash
# tuning aid: fp64 re-run rate of alternative builds of the HIP library (rovaca_amd/lib/variants/lib_*.so)
echo "=== baseline"; timeout -k 5 100 python tools/bench_rescue.py | tail -2
for v in rovaca_amd/lib/variants/lib_*.so; do
  echo "=== $v"
  ROVACA_AMD_HIP_LIB=$PWD/$v timeout -k 5 100 python tools/bench_rescue.py | tail -2
done
