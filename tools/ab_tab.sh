mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_bench_contract.py::test_device_line_carries_the_measured_traffic > gpurun_out/r5/gputests.log 2>&1
tail -5 gpurun_out/r5/gputests.log
timeout -k 10 400 python tests/debug/fuzz_gpu.py 150 2025 > gpurun_out/r5/fuzz1.log 2>&1; tail -3 gpurun_out/r5/fuzz1.log
timeout -k 10 400 python tests/debug/fuzz_gpu.py 120 77 > gpurun_out/r5/fuzz2.log 2>&1; tail -3 gpurun_out/r5/fuzz2.log
export PHMM_ENABLE_TEST_HOOKS=1
python tools/bench_sweep.py 150 300
PHMM_NO_UNIFORM_GOP=1 python tools/bench_sweep.py 150 300
