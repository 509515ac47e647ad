mkdir -p gpurun_out/r5
export PHMM_ENABLE_TEST_HOOKS=1
PHMM_TAB=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r5/gputests_tab.log 2>&1
tail -30 gpurun_out/r5/gputests_tab.log
