mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_shards.py tests/test_rescue_f128.py tests/test_gpu_fuzz.py -q -x -k "config5 or f128 or long_read" > gpurun_out/r5/c5tests.log 2>&1; tail -4 gpurun_out/r5/c5tests.log
python bench.py --config c5 --no-adapter --steps 5 --warmup 2 > gpurun_out/r5/c5_tab.json 2> gpurun_out/r5/c5_tab.err; tail -c 300 gpurun_out/r5/c5_tab.err
PHMM_ENABLE_TEST_HOOKS=1 PHMM_NO_BAND_TAB=1 python bench.py --config c5 --no-adapter --steps 5 --warmup 2 > gpurun_out/r5/c5_notab.json 2>/dev/null
python - <<'PY'
import json
for n in ("tab","notab"):
    d=json.loads(open(f"gpurun_out/r5/c5_{n}.json").read().strip().splitlines()[-1])
    print(n, "value", d["value"], "ms", d["ms_per_step"], "fp32_first", d.get("fp32_first",{}).get("value"), d.get("fp32_first",{}).get("identical_log10"))
PY
