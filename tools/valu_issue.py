#!/usr/bin/env python3
"""Runs tools/valu_issue_bench.bin and converts its times into SIMD clocks per wave64 instruction with the shader clock
sampled from sysfs (pp_dpm_sclk, the line marked '*') while the kernels run; 2.4 GHz nominal when sysfs is not readable.

    python tools/valu_issue.py [out.json]
"""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def read_sclk_mhz():
    best = None
    for path in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(path):
                m = re.search(r"(\d+)\s*[Mm][Hh]z\s*\*", line)
                if m:
                    v = int(m.group(1))
                    best = v if best is None else max(best, v)
        except OSError:
            pass
    return best


def main():
    exe = os.path.join(HERE, "valu_issue_bench.bin")
    samples = []
    stop = threading.Event()

    def poll():
        while not stop.is_set():
            v = read_sclk_mhz()
            if v:
                samples.append(v)
            time.sleep(0.02)
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    res = subprocess.run([exe], capture_output=True, text=True, check=True)
    stop.set(); th.join()
    busy = sorted(s for s in samples if s > 500)
    ghz = (busy[len(busy) // 2] / 1000.0) if busy else 2.4
    rows = [json.loads(line) for line in res.stdout.splitlines() if line.startswith("{")]
    for r in rows:
        r["GHz"] = ghz
        r["GHz_source"] = "sysfs pp_dpm_sclk, median while running" if busy else "nominal (sysfs not readable)"
        r["simd_clocks_per_wave_instruction"] = r["ns_per_wave_instruction_per_simd"] * ghz
        del r["assumed_GHz"]
    out = {"device_clock_samples_MHz": {"n": len(busy), "min": busy[0] if busy else None, "max": busy[-1] if busy else None}, "rows": rows}
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")
    for r in rows:
        print(f'{r["instruction"]:32s} {r["waves_per_simd"]} waves/SIMD  {r["ns_per_wave_instruction_per_simd"]:.3f} ns  '
              f'{r["simd_clocks_per_wave_instruction"]:.2f} clk @ {ghz:.2f} GHz')


if __name__ == "__main__":
    main()
