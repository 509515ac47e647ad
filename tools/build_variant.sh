#!/bin/bash
# build_variant.sh NAME "EXTRA HIPCC FLAGS" -- an alternative build of the PairHMM kernels (tuning experiments): compiles
# pairhmm_kernels.hip and pairhmm_kernels_ragged.hip with the extra flags and links rovaca_amd/lib/variants/lib_NAME.so from it and the regular objects.
# Run the regular build first (make -C rovaca_amd/csrc).  Use: ROVACA_AMD_HIP_LIB=$PWD/rovaca_amd/lib/variants/lib_NAME.so python bench.py ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rovaca_amd/csrc"
OUT=../lib; V=$OUT/variants; mkdir -p $V
FLAGS=$(make -s print-hipflags)  # the regular build's flags, from the Makefile
/opt/rocm/bin/hipcc $FLAGS "$@" -c pairhmm_kernels.hip -o $V/kernels_$NAME.o
/opt/rocm/bin/hipcc $FLAGS "$@" -c pairhmm_kernels_ragged.hip -o $V/kernels_ragged_$NAME.o
# its own build id: bench.py refuses the committed PMC counters (collected on the regular build) for a variant library
g++ -O2 -fPIC -DPHMM_KERNEL_BUILD_ID="\"variant:$NAME\"" -c pairhmm_buildid.cpp -o $V/buildid_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib_$NAME.so $V/kernels_$NAME.o $V/kernels_ragged_$NAME.o $OUT/pairhmm_kernels_tab.o $OUT/pairhmm_kernels_tab_s1.o $OUT/pairhmm_kernels_tab_s2.o $V/buildid_$NAME.o $OUT/pairhmm_kernels_ragged_tab_ug.o $OUT/pairhmm_kernels_ragged_tab_ugo.o $OUT/pairhmm_setup.o $OUT/pairhmm_capi.o $OUT/pairhmm_sweep.o $OUT/pairhmm_chunked.o $OUT/pairhmm_multi.o $OUT/pairhmm_stage.o \
    $OUT/pairhmm_binning.o $OUT/pairhmm_tables.o $OUT/pairhmm_marginal.o $OUT/pairhmm_cpu.o $OUT/pairhmm_host_logic.o -lpthread
echo "$V/lib_$NAME.so"
