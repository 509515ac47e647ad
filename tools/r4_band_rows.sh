#!/bin/bash
# config 5 under band-kernel builds with other stripe heights (tools/build_variant.sh NAME -DPHMM_BAND_ROWS_F32=n / _F64=n): the corner
# parity test, then bench.py --config c5 (value, pass ms, fp32 ms), the regular build before and after.  usage: tools/r4_band_rows.sh [tag]
set -u
cd "$GRAFT_REPO_ROOT"
export PHMM_ENABLE_TEST_HOOKS=1
TAG=${1:-r4_band_rows}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {
  timeout -k 5 200 python3 bench.py --config c5 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python3 - "$OUT/bench_$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], "GCUPS  pass", d["ms_per_step"], "ms  fp32", d.get("fp32_kernel_ms", d.get("roofline", {}).get("kernel_ms")), flush=True)
except Exception as e:
    print(sys.argv[2], "no line:", e, flush=True)
PY
}
run regular_a
for v in rovaca_amd/lib/variants/lib_band*.so; do
  n=$(basename $v .so)
  export ROVACA_AMD_HIP_LIB=$PWD/$v
  timeout -k 5 300 python3 -m pytest tests/test_gpu_shards.py -x -q -m gpu -k "config5_corner" > $OUT/test_$n.log 2>&1
  echo "$n parity: $(tail -1 $OUT/test_$n.log)"
  run $n
  unset ROVACA_AMD_HIP_LIB
done
run regular_b
