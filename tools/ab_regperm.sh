export PHMM_ENABLE_TEST_HOOKS=1
V=$PWD/rovaca_amd/lib/variants
python tools/bench_sweep.py 150 300 | cut -c1-100
PHMM_NO_UNIFORM_GOP=1 python tools/bench_sweep.py 150 300 | cut -c1-100 | sed 's/^/   ug: /'
for lib in $V/lib_rp*.so; do
ROVACA_AMD_HIP_LIB=$lib python tools/bench_sweep.py 150 300 | cut -c1-100
PHMM_NO_UNIFORM_GOP=1 ROVACA_AMD_HIP_LIB=$lib python tools/bench_sweep.py 150 300 | cut -c1-100 | sed 's/^/   ug: /'
done
