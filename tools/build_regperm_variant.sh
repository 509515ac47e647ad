#!/bin/bash
# build_regperm_variant.sh NAME SPEC.json -- the table kernels with the VGPRs of chosen kernels renumbered (tools/regperm.py) in hipcc's
# device assembly, linked with the regular objects into rovaca_amd/lib/variants/lib_NAME.so.  The steps are hipcc's own (hipcc -###):
# device asm -> (renumber) -> assemble -> lld -> offload bundle -> host compile with the bundle included.
set -e
NAME=$1; SPEC=$(readlink -f $2)
ROOT=$(readlink -f "$(dirname "$0")/..")
cd $ROOT/rovaca_amd/csrc
OUT=../lib; V=$OUT/variants; mkdir -p $V
LLVM=/opt/rocm/lib/llvm/bin
FLAGS=$(make -s print-hipflags)
[ -f $V/tab_dev.s ] || /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o $V/tab_dev.s pairhmm_kernels_tab.hip 2>/dev/null
python3 $ROOT/tools/regperm.py $V/tab_dev.s $V/tab_dev_$NAME.s $SPEC
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $V/tab_dev_$NAME.s -o $V/tab_dev_$NAME.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $V/tab_dev_$NAME.out $V/tab_dev_$NAME.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$V/tab_dev_$NAME.out -output=$V/tab_$NAME.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $V/tab_$NAME.hipfb -c pairhmm_kernels_tab.hip -o $V/kernels_tab_$NAME.o 2>/dev/null
g++ -O2 -fPIC -DPHMM_KERNEL_BUILD_ID="\"variant:$NAME\"" -c pairhmm_buildid.cpp -o $V/buildid_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib_$NAME.so $OUT/pairhmm_kernels.o $OUT/pairhmm_kernels_ragged.o $V/kernels_tab_$NAME.o $V/buildid_$NAME.o $OUT/pairhmm_kernels_ragged_tab_ug.o $OUT/pairhmm_kernels_ragged_tab_ugo.o $OUT/pairhmm_setup.o $OUT/pairhmm_capi.o $OUT/pairhmm_sweep.o $OUT/pairhmm_chunked.o $OUT/pairhmm_multi.o $OUT/pairhmm_stage.o \
    $OUT/pairhmm_binning.o $OUT/pairhmm_tables.o $OUT/pairhmm_marginal.o $OUT/pairhmm_cpu.o $OUT/pairhmm_host_logic.o -lpthread
for f in tab_dev_$NAME.s tab_dev_$NAME.o tab_dev_$NAME.out tab_$NAME.hipfb kernels_tab_$NAME.o buildid_$NAME.o; do rm -f "$V/$f"; done
echo "$V/lib_$NAME.so"
