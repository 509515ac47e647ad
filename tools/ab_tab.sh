export PHMM_ENABLE_TEST_HOOKS=1
V=$PWD/rovaca_amd/lib/variants
python tools/bench_sweep.py 150 300
ROVACA_AMD_HIP_LIB=$V/lib_w5.so python tools/bench_sweep.py 150 300
ROVACA_AMD_HIP_LIB=$V/lib_w3.so python tools/bench_sweep.py 150 300
python tools/bench_sweep.py 100 200
ROVACA_AMD_HIP_LIB=$V/lib_w5.so python tools/bench_sweep.py 100 200
python tools/bench_sweep.py 250 500
ROVACA_AMD_HIP_LIB=$V/lib_w3.so python tools/bench_sweep.py 250 500
