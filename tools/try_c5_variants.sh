#!/bin/bash
# tuning aid: long-read configuration against alternative builds of the HIP library (rovaca_amd/lib/variants/lib_*.so),
# optionally with the fp64 rows per lane forced (second argument list)
run() { timeout -k 5 150 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*\|"pass_kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
echo "=== baseline"; run
export PHMM_ENABLE_TEST_HOOKS=1
for r in 4 5; do echo "=== baseline rows=$r"; PHMM_FORCE_F64_ROWS=$r run; done
for v in rovaca_amd/lib/variants/lib_*.so; do
  for r in 4 5 6; do
    echo "=== $v rows=$r"
    ROVACA_AMD_HIP_LIB=$PWD/$v PHMM_FORCE_F64_ROWS=$r run
  done
done
