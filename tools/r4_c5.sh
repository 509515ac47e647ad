#!/bin/bash
# config 5 A/B: the default library against a variant (rovaca_amd/lib/variants/lib_$1.so); usage: tools/r4_c5.sh VARIANT [tag]
set -u
cd "$GRAFT_REPO_ROOT"
V=$PWD/rovaca_amd/lib/variants/lib_${1:-pf2}.so; TAG=${2:-r4_c5}; OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python3 bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e"
for r in 1 2; do
  $B > $OUT/new_$r.json 2>/dev/null; python3 -c "import json; d=json.load(open('$OUT/new_$r.json')); print('new', d['value'], d['pass_kernel_ms'], d['roofline']['kernel_ms'])"
  ROVACA_AMD_HIP_LIB=$V $B > $OUT/var_$r.json 2>/dev/null; python3 -c "import json; d=json.load(open('$OUT/var_$r.json')); print('${1:-pf2}', d['value'], d['pass_kernel_ms'], d['roofline']['kernel_ms'])"
done
python3 tools/bench_c5_latency.py > $OUT/lat_new.txt 2>&1 &
wait
ROVACA_AMD_HIP_LIB=$V python3 tools/bench_c5_latency.py > $OUT/lat_var.txt 2>&1
paste -d'|' <(cut -c1-70 $OUT/lat_new.txt) <(cut -c34-70 $OUT/lat_var.txt) | head -32
