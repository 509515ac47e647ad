export PHMM_ENABLE_TEST_HOOKS=1
V=$PWD/rovaca_amd/lib/variants
for shape in "32 100" "48 100" "80 160" "96 192" "144 288" "240 480"; do
for lib in loopall peelall; do
ROVACA_AMD_HIP_LIB=$V/lib_$lib.so python tools/bench_sweep.py $shape | cut -c1-90
PHMM_NO_UNIFORM_GOP=1 ROVACA_AMD_HIP_LIB=$V/lib_$lib.so python tools/bench_sweep.py $shape | cut -c1-90 | sed 's/^/   ug: /'
done; done
