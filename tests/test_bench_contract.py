"""bench.py prints ONE JSON line with the keys the driver reads; checked on the CPU backend here and on the device under -m gpu."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config"}


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # one line, nothing else on stdout
    return json.loads(lines[0])


def test_cpu_backend_line_has_the_contract_keys():
    d = _run("--backend", "cpu", "--config", "c1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert REQUIRED <= set(d)
    assert d["metric"] == "PairHMM GCUPS" and d["unit"] == "GCUPS" and d["n_gpus"] == 0 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 0 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.gpu
def test_device_line_has_roofline_and_cpu_baseline():
    d = _run("--scale", "0.05", "--steps", "3", "--warmup", "1", "--cpu-threads", "4")
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak" and d["data"] == "synthetic"
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-5 and roof["kernel_ms"] > 0
    # algorithmic bytes: 3R + H + 4 per pair = 754 B at 150 x 300
    assert roof["algorithmic_bytes"] == 754 * d["config"]["pairs_per_gpu"]
    base = d["cpu_baseline"]
    assert base["kind"] in ("reference", "port") and base["cores"] == 4 and base["value"] > 0 and base["host_cores"] >= 1
    assert base["bit_identical_to_gpu_fp32"] is True and d["max_abs_delta_log10_vs_cpu"] == 0.0
    assert d["e2e"]["identical_to_resident_path"] is True and d["e2e"]["value"] > 0
    assert d["pcr_capped_gap_open"]["kernel_gcups"] > 0
    # the drop-in adapter leg: workers -> rovaca::call_pairhmm -> batcher -> device -> H8, checked against the resident path
    ad = d["adapter"]
    assert "error" not in ad, ad
    assert ad["bit_identical_to_resident_path"] is True and ad["value"] > 0 and ad["synchronous_seam"]["value"] > 0
    assert ad["workers"] >= 1 and ad["regions_per_submission"] >= 1.0 and ad["regions"] == ad["workers"] * 2000
    assert ad["mixed_shapes"]["value"] > 0 and ad["mixed_shapes"]["synchronous_seam"]["value"] > 0  # regions that are not all alike


def test_committed_pmc_counters_belong_to_the_committed_kernels():
    # profiles/hbm_traffic.json feeds roofline.traffic and roofline_issue on the bench line: it must have been collected from the
    # kernel sources in the tree (tools/profile_round.sh + tools/summarize_profiles.py after the last kernel change)
    from rovaca_amd import kernel_source_id, loaded_kernel_build_id
    pj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    assert loaded_kernel_build_id() == kernel_source_id(), "the built library is older than the kernel sources (or their flags): make -C rovaca_amd/csrc"
    assert pj.get("kernel_source_id") == kernel_source_id(), "stale PMC counters: re-run tools/profile_round.sh on the GPU box and tools/summarize_profiles.py"
    assert pj["c2"] > 0 and pj["c2_valu_wave_instructions"] > 0 and "pairhmm_systolic" in pj["c2_kernel"]


def test_committed_smith_waterman_counters_belong_to_the_committed_kernel():
    from rovaca_amd import loaded_sw_kernel_build_id, sw_kernel_source_id
    pj = json.load(open(os.path.join(ROOT, "profiles", "sw_counters.json")))
    assert loaded_sw_kernel_build_id() == sw_kernel_source_id(), "the built library is older than the kernel sources (or their flags)"
    assert pj.get("kernel_source_id") == sw_kernel_source_id(), "stale PMC counters: re-run tools/profile_round.sh and tools/summarize_profiles.py"
    assert pj["valu_wave_instructions"] > 0 and "sw_kernel" in pj["kernel"]


@pytest.mark.gpu
def test_smith_waterman_line_has_its_issue_roofline():
    d = _run("--config", "sw", "--no-cpu-baseline")
    assert d["metric"] == "Smith-Waterman GCUPS" and d["value"] > 0
    assert 0.2 < d["roofline_issue"]["frac"] <= 1.0 and d["roofline"]["traffic"] > d["roofline"]["algorithmic_bytes"]  # the backtrack flags


@pytest.mark.gpu
def test_device_line_carries_the_measured_traffic():
    d = _run("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-adapter")
    assert d["roofline"]["traffic"] and d["roofline"]["traffic"] < d["roofline"]["algorithmic_bytes"]  # reads shared 25x / 40x inside the L2s
    assert d["kernel_build_id"] and d["kernel_build_id"] == d["counters_kernel_source_id"]  # the counters were taken on these kernels
    assert 0.5 < d["roofline_issue"]["frac"] <= 1.0
    # the other configs ride on the default line, each checked against the CPU leg on a sample
    cfg = d["configs"]
    assert "error" not in cfg, cfg
    c3, c5, sw = cfg["c3"], cfg["c5"], cfg["sw"]
    assert c3["value_resident"] > 0 and c3["value_e2e"] > 0 and c3["pairs"] == 1000000 and 1 <= c3["launches_per_pass"] <= 4
    assert c3["bit_identical_to_cpu_sample"] is True and c3["e2e_identical_to_resident_path"] is True
    assert c5["value"] > 0 and c5["fp64_ms"] > 0 and c5["rescued_pairs"] == c5["pairs"] == 1000 and c5["identical_to_cpu_sample"] is True
    assert sw["value"] > 0 and "bit-identical" in sw["parity_sample"]
