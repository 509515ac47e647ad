#!/bin/bash
# sweep_matrix.sh OUT VARIANT... -- resident sweep rate of every rows-per-lane class (reads of 16 K bases against 300-base haplotypes), both
# arithmetic forms, for the regular library ("default") and tab-kernel variants (tools/build_tab_variant.sh).  One line per run.
OUT=$1; shift
mkdir -p $(dirname $OUT); : > $OUT
for v in "$@"; do
  if [ $v = default ]; then unset ROVACA_AMD_HIP_LIB; else export ROVACA_AMD_HIP_LIB=$PWD/rovaca_amd/lib/variants/lib_$v.so; fi
  for form in ugo general; do
    if [ $form = general ]; then export PHMM_ENABLE_TEST_HOOKS=1 PHMM_NO_UNIFORM_GOP=1; else unset PHMM_ENABLE_TEST_HOOKS PHMM_NO_UNIFORM_GOP; fi
    for K in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
      echo "$v $form K=$K $(timeout -k 10 100 python tools/bench_sweep.py $((16 * K)) 300 2>&1 | tail -1 | cut -c1-66)" >> $OUT
    done
  done
  echo "$v done"
done
