#!/usr/bin/env python3
"""VALU issue clocks of a kernel's main loops by SOURCE LINE (hipcc -S -gline-tables-only + the measured issue costs of
tools/valu_issue_bench.hip: 2.3 clocks per wave instruction at full rate; 4.5 for packed fp32, v_bfe, DPP, fp64, lane
reads -- and for ANY instruction with a scalar-register source operand).

    python tools/isa_line_costs.py <file.hip> <mangled-name substring> [top N] [-- extra hipcc flags]
"""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _isa_loops import find_main_loops, issue_clocks as cost      # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    src, pat = args[0], args[1]
    top = int(args[2]) if len(args) > 2 else 30
    asm = "/tmp/isa_lines.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include",
                    f"-I{os.path.dirname(os.path.abspath(src))}", "-S", "--cuda-device-only", "-gline-tables-only", src, "-o", asm] + extra,
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if m.group(3) and not m.group(3).startswith("/") else (m.group(3) or m.group(2))
    start = next(i for i, l in enumerate(lines) if re.match(rf"^_Z\S*{pat}\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    srcs = {}
    for (a, b) in find_main_loops(body):
        cur = (0, 0)
        clk = collections.Counter(); cnt = collections.Counter()
        for l in body[a:b + 1]:
            t = l.strip()
            m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
            if m:
                cur = (int(m.group(1)), int(m.group(2))); continue
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            c = cost(t)
            if c:
                clk[cur] += c; cnt[cur] += 1
        print(f"loop {a}-{b}: {sum(clk.values()):.0f} VALU issue clocks per wave and pass")
        for (f, ln), c in sorted(clk.items(), key=lambda kv: -kv[1])[:top]:
            fn = files.get(f, "?")
            if fn not in srcs:
                try:
                    srcs[fn] = open(fn).read().split("\n")
                except Exception:
                    srcs[fn] = []
            text = srcs[fn][ln - 1].strip()[:100] if 0 < ln <= len(srcs[fn]) else ""
            print(f"  {c:6.0f} clk {cnt[(f, ln)]:4d} ins  {os.path.basename(fn)}:{ln}  {text}")


if __name__ == "__main__":
    main()
