#!/usr/bin/env python3
"""Development aid: what a resident CG solve costs besides its steps -- solves of 1, 11, 51 and 101 steps timed (host clock around 20 solves each,
plain launches), a line fitted: slope = us per step, intercept = launch + prologue (loads, ring, census) + pass 0 + epilogue per solve.
python tools/cg_intercept.py [size] [sf]"""
import importlib, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc = pkg.synth.make_scene(size, size, sf, 2, seed=1237, mask_kind="full")
ctx = pkg.Context(device_id=0)
ctx.set_option("exclusive_device", 1)
ctx.setup(pkg.DataHandler.from_scene(sc))
pkg.alternating_loop(ctx, None, max_outer=1)
ctx.bench_cg(solves=5, iters=101)
pts = []
for it in (1, 11, 51, 101):
    r = []
    for _ in range(5):
        b = ctx.bench_cg(solves=20, iters=it)
        r.append(1e6 * b["seconds"] / 20)
    pts.append((it, statistics.median(r)))
n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts); sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
print({"us_per_solve_by_steps": {k: round(v, 1) for k, v in pts}, "us_per_step": round(slope, 3), "intercept_us": round(icpt, 1)})
ctx.close()
