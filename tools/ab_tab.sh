export PHMM_ENABLE_TEST_HOOKS=1
V=$PWD/rovaca_amd/lib/variants
PHMM_NO_TAB=1 python tools/bench_sweep.py 150 300 | cut -c1-100
python tools/bench_sweep.py 150 300 | cut -c1-100
ROVACA_AMD_HIP_LIB=$V/lib_loopall.so python tools/bench_sweep.py 150 300 | cut -c1-100
ROVACA_AMD_HIP_LIB=$V/lib_peelall.so python tools/bench_sweep.py 150 300 | cut -c1-100
python tools/bench_sweep.py 150 300 | cut -c1-100
export PHMM_NO_UNIFORM_GOP=1
PHMM_NO_TAB=1 python tools/bench_sweep.py 150 300 | cut -c1-100
python tools/bench_sweep.py 150 300 | cut -c1-100
ROVACA_AMD_HIP_LIB=$V/lib_loopall.so python tools/bench_sweep.py 150 300 | cut -c1-100
ROVACA_AMD_HIP_LIB=$V/lib_peelall.so python tools/bench_sweep.py 150 300 | cut -c1-100
