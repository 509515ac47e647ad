#!/bin/bash
# build_tab_variant.sh NAME "EXTRA HIPCC FLAGS" -- an alternative build of the table kernels (tuning experiments): compiles
# pairhmm_kernels_tab.hip and both builds of pairhmm_kernels_ragged_tab.hip with the extra flags and links rovaca_amd/lib/variants/lib_NAME.so from it and the regular objects.
# Run the regular build first.  Use: ROVACA_AMD_HIP_LIB=$PWD/rovaca_amd/lib/variants/lib_NAME.so python tools/bench_sweep.py ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rovaca_amd/csrc"
OUT=../lib; V=$OUT/variants; mkdir -p $V
FLAGS=$(make -s print-hipflags)
# the one-class kernels are three builds, a scheduler each (pairhmm_tab_sched.mk); with -DPHMM_TAB_ONE_BUILD every class is in the first
# (scheduler and loop-form experiments: tools/sweep_matrix.sh) and the regular objects stand in for the other two
S1=$(sed -n 's/^TAB_SCHED_1 := //p' pairhmm_tab_sched.mk); S2=$(sed -n 's/^TAB_SCHED_2 := //p' pairhmm_tab_sched.mk)
/opt/rocm/bin/hipcc $FLAGS "$@" -DPHMM_TAB_SCHED=0 -c pairhmm_kernels_tab.hip -o $V/kernels_tab_$NAME.o &
T1=$OUT/pairhmm_kernels_tab_s1.o; T2=$OUT/pairhmm_kernels_tab_s2.o
case " $* " in
  *PHMM_TAB_ONE_BUILD*) ;;
  *) T1=$V/kernels_tab_s1_$NAME.o; T2=$V/kernels_tab_s2_$NAME.o
     /opt/rocm/bin/hipcc $FLAGS $S1 "$@" -DPHMM_TAB_SCHED=1 -c pairhmm_kernels_tab.hip -o $T1 &
     /opt/rocm/bin/hipcc $FLAGS $S2 "$@" -DPHMM_TAB_SCHED=2 -c pairhmm_kernels_tab.hip -o $T2 & ;;
esac
/opt/rocm/bin/hipcc $FLAGS "$@" -DPHMM_RAGGED_TAB_MODE=1 -c pairhmm_kernels_ragged_tab.hip -o $V/kernels_ragged_tab_ug_$NAME.o &
/opt/rocm/bin/hipcc $FLAGS "$@" -DPHMM_RAGGED_TAB_MODE=2 -c pairhmm_kernels_ragged_tab.hip -o $V/kernels_ragged_tab_ugo_$NAME.o &
wait
g++ -O2 -fPIC -DPHMM_KERNEL_BUILD_ID="\"variant:$NAME\"" -c pairhmm_buildid.cpp -o $V/buildid_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib_$NAME.so $OUT/pairhmm_kernels.o $OUT/pairhmm_kernels_ragged.o $V/kernels_tab_$NAME.o $T1 $T2 $V/buildid_$NAME.o $V/kernels_ragged_tab_ug_$NAME.o $V/kernels_ragged_tab_ugo_$NAME.o $OUT/pairhmm_setup.o $OUT/pairhmm_capi.o $OUT/pairhmm_sweep.o $OUT/pairhmm_chunked.o $OUT/pairhmm_multi.o $OUT/pairhmm_stage.o \
    $OUT/pairhmm_binning.o $OUT/pairhmm_tables.o $OUT/pairhmm_marginal.o $OUT/pairhmm_cpu.o $OUT/pairhmm_host_logic.o -lpthread
echo "$V/lib_$NAME.so"
