#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_<round>/) into the tracked files under profiles/:

    python tools/pmc_traffic.py r02

  profiles/<round>_bench_kernel_stats.csv, <round>_bench.json       kernel statistics + JSON line of the same bench command
  profiles/<round>_pmc_<config>_<counter>.csv                       rows of the CG kernels of every PMC pass
  profiles/<round>_traffic.json                                     HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> B; the
                                                                    gfx950 correction of MI355X_MICROARCH.md), median over the launches
  profiles/<round>_valu_issue.json                                  the VALU issue microbenchmark
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CG_KERNELS = ("k_cg_resident", "k_apply_march", "k_cg_update_r", "k_cg_flush")


def rows_of(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
    dst = os.path.join(ROOT, "profiles")
    # gpurun merges every call's output into the same directory: the newest file of a kind is the one of the last run
    newest = lambda pattern: sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime, reverse=True)
    stats = newest(os.path.join(src, "bench", "**", "*kernel_stats.csv"))
    # (bench.py's legs start child processes -- tools/hbm_ceiling_bench.bin -- which the profiler follows and gives statistics files
    # of their own: the bench's is the newest one that holds the library's kernels)
    stats = [f for f in stats if "srps::" in open(f).read()]
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{rnd}_bench_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "bench.json")):
        line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")]
        if line:
            open(os.path.join(dst, f"{rnd}_bench.json"), "w").write(line[-1])
    if os.path.exists(os.path.join(src, "valu_issue.json")):
        shutil.copy(os.path.join(src, "valu_issue.json"), os.path.join(dst, f"{rnd}_valu_issue.json"))
    traffic = {"note": "HBM-side bytes per launch from rocprofv3 PMC (separate passes): 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction of "
                       "MI355X_MICROARCH.md section HBM); Infinity-Cache hits are included in these counters; medians over the launches of a pass"}
    keymap = {"resident_2048": ("2048x2048_sf4", "resident"), "streaming_2048": ("2048x2048_sf4", "apply"), "streaming_4096": ("4096x4096_sf2", "apply")}
    for cfg, (size_key, leg) in keymap.items():
        per_counter = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            files = newest(os.path.join(src, f"pmc_{cfg}_{counter}", "**", "*counter_collection.csv"))
            if not files:
                continue
            rows = [r for r in rows_of(files[0]) if any(k in r["Kernel_Name"] for k in CG_KERNELS)]
            with open(os.path.join(dst, f"{rnd}_pmc_{cfg}_{counter.lower()}.csv"), "w", newline="") as f:
                if rows:
                    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
            main_kernel = "k_cg_resident" if leg == "resident" else "k_apply_march"
            vals = [float(r["Counter_Value"]) for r in rows if main_kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
            durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if main_kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
            if leg != "resident":          # the step launches: skip the residual launch of every solve (MODE 1) by taking the median
                pass
            if vals:
                per_counter[counter] = (statistics.median(vals), statistics.median(durs), len(vals))
        if "FETCH_SIZE" in per_counter and "WRITE_SIZE" in per_counter:
            d = traffic.setdefault(size_key, {})
            d[leg] = 2 * per_counter["FETCH_SIZE"][0] * 1024 + per_counter["WRITE_SIZE"][0] * 1024
            d[leg + "_detail"] = {"fetch_KB": per_counter["FETCH_SIZE"][0], "write_KB": per_counter["WRITE_SIZE"][0],
                                  "median_us_under_pmc": per_counter["FETCH_SIZE"][1], "launches": per_counter["FETCH_SIZE"][2]}
    for sq in ("SQ1", "SQ2"):
        files = newest(os.path.join(src, f"pmc_resident_2048_{sq}", "**", "*counter_collection.csv"))
        if files:
            rows = [r for r in rows_of(files[0]) if "k_cg_resident" in r["Kernel_Name"]]
            with open(os.path.join(dst, f"{rnd}_pmc_resident_2048_{sq.lower()}.csv"), "w", newline="") as f:
                if rows:
                    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
            agg = {}
            for r in rows:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            traffic.setdefault("2048x2048_sf4", {}).setdefault("resident_sq", {}).update({k: statistics.median(v) for k, v in agg.items()})
    json.dump(traffic, open(os.path.join(dst, f"{rnd}_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
