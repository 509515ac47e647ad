"""Turn gpurun_out/prof (tools/profile_round.sh) into the committed summaries under profiles/: per-config kernel stats,
the PMC summary of the headline kernel and profiles/hbm_traffic.json (read by bench.py for roofline.traffic)."""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r1_final"
dst = os.path.join(ROOT, "profiles")

stats = {}
for c in ("c2", "c3", "c5"):
    f = os.path.join(src, "trace_" + c, c + "_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, c)))
        stats[c] = list(csv.DictReader(open(f)))

hot = next(r for r in stats["c2"] if "pairhmm_systolic" in r["Name"])  # rows are sorted by total time
bench_ms = None
m = re.search(r'"kernel_ms": ([0-9.]+)', open(os.path.join(src, "bench_c2.log")).read())
if m:
    bench_ms = float(m.group(1))

def collect(pattern, kernel_substring):
    """mean per launch of every counter over the PMC passes matching `pattern`, and the kernel's name"""
    counters, name = {}, None
    for f in glob.glob(os.path.join(src, pattern, "pmc_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel_substring not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"]
            counters.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            counters[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return counters, {k: sum(v.values()) / len(v) for k, v in counters.items()}, name


counters, mean, _ = collect("pmc_*", "pairhmm_systolic")
# the same batch through the general kernel (per-base gap-opens, what a run under the default PCR model takes): PHMM_NO_UNIFORM_GOP=1 passes
_, mean_pcr, pcr_kernel = collect("pmcpcr_*", "pairhmm_systolic")
avg_ms = float(hot["AverageNs"]) / 1e6
lines = [
    "rocprofv3 --pmc passes (separate runs, one counter group each; tools/profile_round.sh) of",
    "`python3 bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline` on MI355X.",
    "kernel: %s (config c2: 1,000,000 pairs, 150 x 300)" % hot["Name"],
    "FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x",
    "(MI355X_MICROARCH.md, HBM section), so HBM read bytes are bracketed by [1x, 2x] FETCH_SIZE * 1024.",
    "kernel-trace average duration of the same kernel (profiles/%s_c2_kernel_stats.csv): %.4f ms (min %.4f); bench.py HIP events in that run: %s ms"
    % (tag, avg_ms, float(hot["MinNs"]) / 1e6, bench_ms),
    "",
]
for k in sorted(mean):
    lines.append("%-24s launches=%d mean_per_launch=%g" % (k, len(counters[k]), mean[k]))
if "SQ_INSTS_VALU" in mean and "SQ_WAVES" in mean:
    per_wave = mean["SQ_INSTS_VALU"] / mean["SQ_WAVES"]
    lines += ["", "derived: VALU wave-instructions per launch %.3g; per wave %.0f; per sweep step (315 steps/pair) %.1f"
              % (mean["SQ_INSTS_VALU"], per_wave, per_wave / 315.0)]
    # 1024 SIMDs; cycles available per SIMD = duration * clock
    for ghz in (2.05,):
        lines.append("         cycles per VALU instruction per SIMD at %.2f GHz over %.3f ms: %.2f"
                     % (ghz, avg_ms, avg_ms * 1e-3 * ghz * 1e9 * 1024 / mean["SQ_INSTS_VALU"]))
def effective_clock(sub):
    """median over the sweep dispatches of GRBM_GUI_ACTIVE / 8 / duration (GHz), and the kernel's name"""
    dur, gui, name = {}, {}, None
    for f in glob.glob(os.path.join(src, sub, "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            if "pairhmm_systolic" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    for f in glob.glob(os.path.join(src, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                gui[r["Dispatch_Id"]] = gui.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    q = sorted(gui[k] / 8.0 / dur[k] for k in gui if dur[k] > 0)
    return (q[len(q) // 2] if q else None), name, (sorted(dur.values())[len(dur) // 2] / 1e6 if dur else None)


clk, clk_name, clk_ms = effective_clock("clk_c2")
clk0, clk0_name, clk0_ms = effective_clock("clk_c2_notab")
if clk:
    lines += ["", "effective clock under the hot kernel (GRBM_GUI_ACTIVE / 8 / dispatch duration, median over the dispatches of a counter pass):",
              "  %s: %.3f GHz (%.3f ms per dispatch in that pass)" % (clk_name, clk, clk_ms)]
    if clk0:
        lines.append("  %s (PHMM_NO_TAB=1, the bit-plane kernel of rounds 1-4): %.3f GHz (%.3f ms)" % (clk0_name, clk0, clk0_ms))
    if "SQ_INSTS_VALU" in mean:
        lines.append("  -> %.2f cycles per VALU wave-instruction per SIMD at that clock" % (clk_ms * 1e-3 * clk * 1e9 * 1024 / mean["SQ_INSTS_VALU"]))
open(os.path.join(dst, "%s_c2_pmc_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    sys.path.insert(0, ROOT)
    from rovaca_amd import kernel_source_id
    fetch, write = mean["FETCH_SIZE"] * 1024, mean["WRITE_SIZE"] * 1024
    extra = {}
    if mean_pcr.get("FETCH_SIZE") and mean_pcr.get("WRITE_SIZE"):
        extra = {"c2_pcr": int(2 * mean_pcr["FETCH_SIZE"] * 1024 + mean_pcr["WRITE_SIZE"] * 1024),
                 "c2_pcr_valu_wave_instructions": int(mean_pcr.get("SQ_INSTS_VALU", 0)),
                 "c2_pcr_kernel": (pcr_kernel or "").replace("void ", "").split("(")[0]}
    # configs 3 and 5: totals per PASS (bench.py --steps 3 --warmup 1 = 4 passes) over every sweep kernel of the pass
    def per_pass(cfg, name_part, passes=4):
        tot, names = {}, set()
        for f in glob.glob(os.path.join(src, "pmc%s_*" % cfg, "pmc_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if name_part in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    names.add(r["Kernel_Name"].replace("void ", "").split("(")[0])
        return {k: v / passes for k, v in tot.items()}, sorted(names)

    c3, c3_names = per_pass("c3", "pairhmm_systolic")
    if c3.get("SQ_INSTS_VALU"):
        extra.update({"c3": int(2 * c3.get("FETCH_SIZE", 0) * 1024 + c3.get("WRITE_SIZE", 0) * 1024),
                      "c3_valu_wave_instructions": int(c3["SQ_INSTS_VALU"]), "c3_waves": int(c3.get("SQ_WAVES", 0)),
                      "c3_kernel": " + ".join(c3_names)})
    for prec, part in (("fp32", "pairhmm_band_kernel<float"), ("fp64", "pairhmm_band_kernel<double")):
        c5, c5_names = per_pass("c5", part)
        if c5.get("SQ_INSTS_VALU"):
            extra.update({"c5_%s" % prec: int(2 * c5.get("FETCH_SIZE", 0) * 1024 + c5.get("WRITE_SIZE", 0) * 1024),
                          "c5_%s_valu_wave_instructions" % prec: int(c5["SQ_INSTS_VALU"]), "c5_%s_waves" % prec: int(c5.get("SQ_WAVES", 0)),
                          "c5_%s_wave_cycles" % prec: int(c5.get("SQ_WAVE_CYCLES", 0)), "c5_%s_busy_cycles" % prec: int(c5.get("SQ_BUSY_CYCLES", 0)),
                          "c5_%s_active_inst_valu" % prec: int(c5.get("SQ_ACTIVE_INST_VALU", 0)), "c5_%s_wait_inst_any" % prec: int(c5.get("SQ_WAIT_INST_ANY", 0)),
                          "c5_%s_kernel" % prec: " + ".join(c5_names)})
    json.dump({"kernel_source_id": kernel_source_id(), **extra, "c2": int(2 * fetch + write),
               "_note": "bytes per launch of %s on config c2: 2 x FETCH_SIZE (gfx950 correction, upper bound) + WRITE_SIZE, "
                        "from profiles/%s_c2_pmc_summary.txt" % (hot["Name"].replace("void ", "").split("(")[0], tag),
               "c2_fetch_raw_bytes": int(fetch), "c2_write_bytes": int(write),
               "c2_valu_wave_instructions": int(mean.get("SQ_INSTS_VALU", 0)), "c2_kernel": hot["Name"].replace("void ", "").split("(")[0],
               "c2_lds_wave_instructions": int(mean.get("SQ_INSTS_LDS", 0)), "c2_effective_clock_ghz": round(clk, 3) if clk else None,
               "c2_effective_clock_ghz_bit_plane_kernel": round(clk0, 3) if clk0 else None},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print("\n".join(lines))
for c in stats:
    print(c, [(r["Name"][:60], r["Calls"], r["AverageNs"]) for r in stats[c][:4]])


# ---- Smith-Waterman aligner (bench.py --config sw) ----
f = os.path.join(src, "trace_sw", "sw_kernel_stats.csv")
if os.path.exists(f):
    shutil.copy(f, os.path.join(dst, "%s_kernel_stats.csv" % tag.replace("final", "sw")))
    rows = list(csv.DictReader(open(f)))
    hot = next(r for r in rows if "sw_kernel" in r["Name"])
    mean = {}
    for g in glob.glob(os.path.join(src, "pmcsw_*", "pmc_counter_collection.csv")):
        c = {}
        for r in csv.DictReader(open(g)):
            if "sw_kernel" in r["Kernel_Name"]:
                c.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                c[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, v in c.items():
            mean[k] = sum(v.values()) / len(v)
    ms = float(hot["AverageNs"]) / 1e6
    m = re.search(r'"kernel_ms": ([0-9.]+)', open(os.path.join(src, "bench_sw.log")).read())
    valu, waves = mean.get("SQ_INSTS_VALU", 0.0), mean.get("SQ_WAVES", 1.0)
    txt = [
        "rocprofv3 on `python3 bench.py --config sw --no-cpu-baseline --no-sustained` (48 000 read-vs-haplotype alignments, 150-153 x 400, genotype",
        "parameters) on MI355X (tools/profile_round.sh).  kernel: %s (16 lanes per alignment, 10 columns per lane)." % hot["Name"],
        "kernel-trace (profiles/%s_kernel_stats.csv): average %.3f ms, min %.3f ms; HIP events inside swhip_align in that run: %s ms."
        % (tag.replace("final", "sw"), ms, float(hot["MinNs"]) / 1e6, m.group(1) if m else "?"),
        "PMC passes (separate runs), per launch:",
        "  ".join("%s %.4g" % (k, mean[k]) for k in sorted(mean)),
    ]
    if valu:
        txt += [
            "derived: %.0f VALU wave-instructions per wave = %.0f per sweep step (400 + 15 steps, 10 cells per lane per step: %.1f per cell"
            % (valu / waves, valu / waves / 415, valu / waves / 4150),
            "         including end-cell search and traceback); over %.2f ms at 2.05 GHz the 1 024 SIMDs spend %.2f cycles per VALU"
            % (ms, ms * 1e-3 * 2.05e9 * 1024 / valu),
            "         instruction; tools/ubench_valu.hip measures 2.4-2.7 for v_add_u32 / v_sub_co_u32 / v_cndmask and 4.2-4.4 for v_cmp to VCC and v_addc_co_u32.",
        ]
    if "WRITE_SIZE" in mean:
        txt += ["bound: integer VALU.  HBM traffic is the backtrack flags (4 bits per cell): %.2f GB written per launch = %.2f TB/s, %.0f %% of 8 TB/s;"
                % (mean["WRITE_SIZE"] * 1024 / 1e9, mean["WRITE_SIZE"] * 1024 / ms / 1e9, 100 * mean["WRITE_SIZE"] * 1024 / ms / 1e9 / 8),
                "       the algorithmic input/output (len1 + len2 in, 20 B + CIGAR out, ~0.6 KB per alignment) is 0.2 % of HBM."]
    open(os.path.join(dst, "%s_pmc_summary.txt" % tag.replace("final", "sw")), "w").write("\n".join(txt) + "\n")
    if valu:
        # read by bench.py --config sw for roofline_issue / roofline.traffic (gated on the kernel sources, like hbm_traffic.json)
        from rovaca_amd import sw_kernel_source_id
        json.dump({"kernel_source_id": sw_kernel_source_id(), "kernel": hot["Name"].replace("void ", "").split("(")[0],
                   "valu_wave_instructions": int(valu), "waves": int(waves),
                   "fetch_raw_bytes": int(mean.get("FETCH_SIZE", 0.0) * 1024), "write_bytes": int(mean.get("WRITE_SIZE", 0.0) * 1024),
                   "_note": "per launch of bench.py --config sw (48 000 alignments, 150-153 x 400), from profiles/%s_pmc_summary.txt" % tag.replace("final", "sw")},
                  open(os.path.join(dst, "sw_counters.json"), "w"), indent=1)
    print("\n".join(txt))
