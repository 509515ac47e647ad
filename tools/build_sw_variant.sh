#!/bin/bash
# build_sw_variant.sh NAME [SOURCE.hip] ["EXTRA HIPCC FLAGS"] -- an alternative build of the Smith-Waterman kernels (tuning experiments):
# compiles SOURCE.hip (default: sw_kernels.hip; e.g. a copy of an earlier revision) and links rovaca_amd/lib/variants/libsw_NAME.so.
# Use: ROVACA_AMD_SW_LIB=$PWD/rovaca_amd/lib/variants/libsw_NAME.so python bench.py --config sw ...
set -e
NAME=$1; SRC=${2:-sw_kernels.hip}; shift; shift || true
[ -f "$SRC" ] && SRC=$(readlink -f "$SRC")
cd "$(dirname "$0")/../rovaca_amd/csrc"
OUT=../lib; V=$OUT/variants; mkdir -p $V
FLAGS=$(make -s print-hipflags)
/opt/rocm/bin/hipcc $FLAGS -I. "$@" -c $SRC -o $V/sw_kernels_$NAME.o
g++ -O2 -fPIC -DSWHIP_KERNEL_BUILD_ID="\"variant:$NAME\"" -c pairhmm_buildid.cpp -o $V/sw_buildid_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libsw_$NAME.so $V/sw_kernels_$NAME.o $OUT/sw_capi.o $V/sw_buildid_$NAME.o -lpthread
rm -f $V/sw_kernels_$NAME.o $V/sw_buildid_$NAME.o
echo "$V/libsw_$NAME.so"
