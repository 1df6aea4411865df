#!/usr/bin/env python3
"""How much of an alternating pass the GPU spends inside kernels: reads a rocprofv3 --kernel-trace CSV of `bench.py` (the headline's
passes are the stretches between two launches of the resident depth CG, k_cg_resident<4, 3, true, true>) and prints, per pass, the
time from the end of one resident launch to the end of the next, the sum of the kernel durations in between, and the idle remainder --
the figure behind "the pass is kernel-bound: a graph has no gaps to remove" (DESIGN.md section 8).

    python tools/pass_gaps.py gpurun_out/prof_r03/bench/runc/<pid>_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
marks = [i for i, e in enumerate(ev) if "k_cg_resident<4, 3, true, true>" in e[2]]
print(f"{len(ev)} dispatches, {len(marks)} launches of the headline's resident CG")
out = []
for a, b in zip(marks[:-1], marks[1:]):
    span = ev[b][1] - ev[a][1]
    inside = ev[a + 1:b + 1]
    if span <= 0 or span > 3_000_000 or len(inside) > 60:      # another leg, a set-up or the host in between
        continue
    busy = sum(e[1] - e[0] for e in inside)
    # overlapping kernels (two streams) would make busy > span; they do not occur inside a pass
    out.append((span, busy, len(inside)))
for span, busy, n in out[:12]:
    print(f"pass {span / 1e3:8.1f} us   in kernels {busy / 1e3:8.1f} us ({100.0 * busy / span:5.1f} %)   idle {(span - busy) / 1e3:6.1f} us over {n} launches")
if out:
    s = sum(o[0] for o in out); b = sum(o[1] for o in out)
    print(f"{len(out)} passes: {100.0 * b / s:.1f} % of the time inside kernels, {(s - b) / len(out) / 1e3:.1f} us idle per pass, {sum(o[2] for o in out) / len(out):.1f} launches per pass")
