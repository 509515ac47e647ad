#!/bin/bash
# tuning aid: headline configuration (or $CONFIG) against alternative builds of the HIP library (rovaca_amd/lib/variants/lib_*.so)
run() { timeout -k 5 150 python bench.py --config ${CONFIG:-c2} --no-cpu-baseline --no-e2e 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
echo "=== baseline"; run
for v in rovaca_amd/lib/variants/lib_*.so; do
  echo "=== $v"
  ROVACA_AMD_HIP_LIB=$PWD/$v run
done
echo "=== baseline again"; run
