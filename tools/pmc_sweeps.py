#!/usr/bin/env python3
"""Counter bytes of the image sweeps against their algorithmic bytes.
    python tools/pmc_sweeps.py <dir with fetch/ and write/ rocprofv3 outputs of tools/pass_prof.py> P N C [out.json]
HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB; the gfx950 correction of MI355X_MICROARCH.md section HBM: FETCH_SIZE reads
half the bytes of a 16-B-per-lane stream), median over the launches of a kernel."""
import csv, glob, json, os, statistics, sys

def rows_of(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))

def main():
    src = sys.argv[1]; P = int(sys.argv[2]); N = int(sys.argv[3]); C = int(sys.argv[4])
    out = {}
    for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        files = sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
        if not files:
            continue
        for r in rows_of(files[0]):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void srps::", "").replace("srps::", "")
            d = out.setdefault(name, {"FETCH_SIZE": [], "WRITE_SIZE": [], "us": []})
            d[counter].append(float(r["Counter_Value"]))
            d["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    img = 4.0 * P * N * C
    # algorithmic bytes of the sweeps: the images once + the per-pixel planes each reads / writes once
    alg = {"k_light_fused_tile": img + 4.0 * P * (6 + C) + 4.0 * P * 4,                # I; dz xx yy z zx zy (6), rho (C); N0 N1 N2 + dz out (4)
           "k_light_fused_mfw": img + 4.0 * P * (6 + C) + 4.0 * P * 4,                 # the same sweep on the matrix pipe (round 5)
           "k_albedo_numden": img + 4.0 * P * 4 + 4.0 * P * (2 * C + 3 * C),          # I; N (4); num, den [C] + the three image sums [C][3]
           "k_albedo_fused": img + 4.0 * P * (3 + 3 + 1) + 4.0 * P * (C + C + 3),       # I; N0..2 (N3 == 1 is not read since round 6), dz xx yy, gofp (the old rho only where a denominator is zero); rho [C], g [C], q [3] out
           "k_depth_from_sums": 4.0 * P * (3 * C + C + 1 + 2) + 4.0 * P * (3 + 3)}
    res = {}
    for name, d in sorted(out.items()):
        if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
            continue
        f = statistics.median(d["FETCH_SIZE"]) * 1024; w = statistics.median(d["WRITE_SIZE"]) * 1024
        us = statistics.median(d["us"])
        key = next((k for k in alg if name.startswith(k)), None)
        res[name] = {"launches": len(d["FETCH_SIZE"]), "median_us_under_pmc": us, "fetch_bytes_x2": 2 * f, "write_bytes": w, "traffic_bytes": 2 * f + w,
                     "algorithmic_bytes": alg.get(key), "traffic_over_algorithmic": (2 * f + w) / alg[key] if key else None,
                     "traffic_GBs": (2 * f + w) / us * 1e-3}
    js = json.dumps(res, indent=1)
    print(js)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(js)

if __name__ == "__main__":
    main()
