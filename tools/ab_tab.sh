mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_contract.py::test_device_line_carries_the_measured_traffic > gpurun_out/r5/gputests.log 2>&1
tail -25 gpurun_out/r5/gputests.log
