#!/usr/bin/env python3
"""Registers, scratch and occupancy of the kernels of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py srmeetsps-cuda_amd/csrc/kernels_resident.hip [name filter] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    src = args[0]
    filt = args[1] if len(args) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include",
           f"-I{os.path.dirname(os.path.abspath(src))}", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", "")).replace("void srps::", "")}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        if filt and filt not in r["name"]:
            continue
        print(f'{r["name"]:60s} VGPR {r.get("VGPRs","?"):>4s} AGPR {r.get("AGPRs","?"):>3s} SGPR {r.get("TotalSGPRs", r.get("SGPRs","?")):>4s} '
              f'scratch {r.get("ScratchSize [bytes/lane]","?"):>5s} occ {r.get("Occupancy [waves/SIMD]","?"):>2s} LDS {r.get("LDS Size [bytes/block]","?")}')


if __name__ == "__main__":
    main()
