#!/bin/bash
# tuning aid: run the per-K benchmark against alternative builds of the HIP library
for v in rovaca_amd/lib/variants/lib_*.so; do
  echo "=== $v"
  ROVACA_AMD_HIP_LIB=$PWD/$v timeout -k 5 200 python tools/bench_by_k.py ${1:-8,9,10,11}
done
