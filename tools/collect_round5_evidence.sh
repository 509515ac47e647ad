#!/bin/bash
# collect_round5_evidence.sh -- the default bench line three times in a row on one box, the c3 / c5 lines, and the per-K rates of the
# table sweep beside the bit-plane sweep (PHMM_NO_TAB=1) on the same box.  Run on the GPU box through gpurun; writes gpurun_out/r5/.
mkdir -p gpurun_out/r5
for i in 1 2 3; do python bench.py > gpurun_out/r5/bench_default_$i.json 2> gpurun_out/r5/bench_default_$i.err; done
python bench.py --config c3 --no-adapter > gpurun_out/r5/bench_c3.json 2> gpurun_out/r5/bench_c3.err
python bench.py --config c5 --no-adapter > gpurun_out/r5/bench_c5.json 2> gpurun_out/r5/bench_c5.err
export PHMM_ENABLE_TEST_HOOKS=1
{
echo "## kernel rate per rows-per-lane class, table sweep (tools/bench_by_k.py: reads of exactly 16 K bases, resident)"
python tools/bench_by_k.py
echo "## the same through the bit-plane kernels (PHMM_NO_TAB=1), same box"
PHMM_NO_TAB=1 python tools/bench_by_k.py
echo "## per-base gap-opens (PHMM_NO_UNIFORM_GOP=1), table sweep / bit-plane"
PHMM_NO_UNIFORM_GOP=1 python tools/bench_by_k.py 4,7,10,12,14,16
PHMM_NO_UNIFORM_GOP=1 PHMM_NO_TAB=1 python tools/bench_by_k.py 4,7,10,12,14,16
} > gpurun_out/r5/evidence_tables.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5/bench_default_*.json"))+["gpurun_out/r5/bench_c3.json","gpurun_out/r5/bench_c5.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    line=[f.split("/")[-1], "value", d["value"], "ms", d["ms_per_step"]]
    if "value_e2e_pcie_inclusive" in d: line+=["e2e", d["value_e2e_pcie_inclusive"]]
    if "pcr_capped_gap_open" in d: line+=["pcr", d["pcr_capped_gap_open"]["kernel_gcups"]]
    if "configs" in d and "c3" in d["configs"]:
        c=d["configs"]; line+=["c3", c["c3"]["value_resident"], c["c3"]["value_e2e"], "c5", c["c5"]["value"], c["c5"]["fp32_first"]["value"], "sw", c["sw"]["value"]]
    if "adapter" in d:
        a=d["adapter"]; line+=["adapter", a["value"], a["synchronous_seam"]["value"], a["mixed_shapes"]["value"], a["mixed_shapes"]["synchronous_seam"]["value"]]
    if "fp32_first" in d: line+=["fp32_first", d["fp32_first"]["value"]]
    line+=["roofline", d["roofline"]["frac"], "valu", d.get("roofline_valu",{}).get("frac"), "issue", d.get("roofline_issue",{}).get("frac"), "cpu", d.get("cpu_baseline",{}).get("value")]
    print(*line)
PY
