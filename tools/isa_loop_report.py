#!/usr/bin/env python3
"""Where the scratch accesses and the instruction mix of one kernel sit in its ISA.

    python tools/isa_loop_report.py <file.hip> <mangled-name substring> [-- extra hipcc flags]

Prints every scratch instruction with its line number, the backward branches (loops) with their spans, and per loop the
instruction histogram by class with the measured issue cost (profiles/r02_valu_issue.json: full-rate VALU 2.3 clocks per
wave instruction, packed fp32 / v_bfe / DPP 4.4-4.6)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF_RATE = ("v_pk_", "v_bfe_", "_dpp", "v_mul_f64", "v_add_f64", "v_fma_f64", "v_mul_lo", "v_mul_hi", "v_mad_u64", "v_cvt_f64", "v_cvt_f32_f64",
             "v_rcp", "v_rsq", "v_sqrt", "v_readlane", "v_writelane", "v_readfirstlane")


def cost(ins, text):
    if not ins.startswith("v_"):
        return 0.0
    if any(k in text for k in HALF_RATE):
        return 4.5
    return 2.3


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    src, pat = args[0], args[1]
    asm = "/tmp/isa_report.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include",
                    f"-I{os.path.dirname(os.path.abspath(src))}", "-S", "--cuda-device-only", src, "-o", asm] + extra, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if pat in l and l.endswith(":") or (pat in l and re.match(r"^_Z\S+:", l)))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    print(f"{lines[start][:100]}  {len(body)} lines")
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            loops.append((labels[m.group(2)], i))
    for i, l in enumerate(body):
        if "scratch_" in l:
            inside = [f"{a}-{b}" for a, b in loops if a <= i <= b]
            print(f"  scratch @{i}: {l.strip()[:70]}   loops: {' '.join(inside)}")
    for a, b in sorted(loops, key=lambda t: t[0] - t[1])[:6]:
        hist = collections.Counter(); clk = collections.Counter()
        for l in body[a:b + 1]:
            t = l.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            ins = t.split()[0]
            cls = ("valu_half" if cost(ins, t) > 4 else "valu_full") if ins.startswith("v_") else \
                  "salu" if ins.startswith("s_") else "lds" if ins.startswith("ds_") else "vmem" if re.match(r"(global|buffer|scratch|flat)_", ins) else "other"
            hist[cls] += 1; clk[cls] += cost(ins, t)
        tot = sum(clk.values())
        print(f"loop {a}-{b} ({b - a} lines): " + ", ".join(f"{k} {v}" for k, v in sorted(hist.items())) + f"; VALU issue clocks per wave {tot:.0f}")


if __name__ == "__main__":
    main()
