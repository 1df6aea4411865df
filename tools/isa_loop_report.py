#!/usr/bin/env python3
"""Where the scratch accesses and the instruction mix of one kernel's main loops sit in its ISA.

    python tools/isa_loop_report.py <file.hip> <mangled-name substring> [--json out.json --key KEY] [-- extra hipcc flags]

Prints per main loop (the CG loop of a persistent kernel): the scratch instructions inside it with their position, the
instruction histogram, and the VALU issue clocks per wave and pass priced with the measured issue costs
(tools/_isa_loops.py: issue_clocks).  --json adds {KEY: {...}} to a profile file that bench.py reads for roofline.issue."""
import collections
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _isa_loops import find_main_loops, issue_clocks      # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    out_json = key = None
    if "--json" in args:
        i = args.index("--json"); out_json = args[i + 1]; del args[i:i + 2]
    if "--key" in args:
        i = args.index("--key"); key = args[i + 1]; del args[i:i + 2]
    src, pat = args[0], args[1]
    asm = "/tmp/isa_report.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include",
                    f"-I{os.path.dirname(os.path.abspath(src))}", "-S", "--cuda-device-only", src, "-o", asm] + extra, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(rf"^_Z\S*{pat}\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    print(f"{lines[start][:110]}  {len(body)} lines")
    result = []
    for a, b in find_main_loops(body):
        hist = collections.Counter(); classes = collections.Counter(); clk = 0.0
        scratch = []
        for i in range(a, b + 1):
            t = body[i].strip()
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            ins = t.split()[0]
            c = issue_clocks(t)
            clk += c
            if "scratch_" in ins:
                scratch.append(i - a)
            cls = ("valu_half_rate_or_scalar_operand" if c > 4 else "valu_full_rate") if ins.startswith("v_") else \
                  "salu" if ins.startswith("s_") else "lds" if ins.startswith("ds_") else "vmem" if re.match(r"(global|buffer|scratch|flat)_", ins) else "other"
            classes[cls] += 1
            if ins.startswith("v_"):
                hist[ins + (" dpp" if "dpp" in t or "row_" in t or "wave_s" in t else "")] += 1
        print(f"loop {a}-{b} ({b - a} lines): " + ", ".join(f"{k} {v}" for k, v in sorted(classes.items())) + f"; VALU issue clocks per wave and pass {clk:.0f}")
        print("   scratch accesses at", scratch)
        print("   " + ", ".join(f"{k} {v}" for k, v in hist.most_common(18)))
        result.append({"lines": b - a, "classes": dict(classes), "valu_issue_clocks_per_wave_step": clk, "scratch_accesses_in_loop": len(scratch),
                       "top_valu": dict(hist.most_common(12))})
    if out_json and key and result:
        try:
            doc = json.load(open(out_json))
        except Exception:
            doc = {"note": "static instruction mix of the CG loop of k_cg_resident (tools/isa_loop_report.py) priced with the measured issue "
                           "clocks per instruction class (tools/valu_issue_bench.hip, profiles/r02_valu_issue.json); both branches of the rare "
                           "direct-sum path are counted, so this is an upper bound of the executed stream by a few per cent"}
        r = result[0]
        doc[key] = dict(r, waves_per_simd=2, GHz=2.4, kernel=pat,
                        source="tools/isa_loop_report.py on srmeetsps-cuda_amd/csrc/kernels_resident.hip; issue clocks: profiles/r02_valu_issue.json")
        json.dump(doc, open(out_json, "w"), indent=1)
        print("wrote", out_json, key)


if __name__ == "__main__":
    main()
