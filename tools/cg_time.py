#!/usr/bin/env python3
"""Development aid: the resident CG step at 2048 x 2048, sf 4 timed carefully -- plain launches (exclusive_device), several rounds
of 20 solves, minimum and median of the rounds.  python tools/cg_time.py [size] [sf] [mask_kind] [NAME=INT options]"""
import importlib, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
opts = [a for a in sys.argv[1:] if "=" in a]            # NAME=INT: srps_set_option before the set-up
sys.argv = [a for a in sys.argv if "=" not in a]
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind = sys.argv[3] if len(sys.argv) > 3 else "full"
sc = pkg.synth.make_scene(size, size, sf, 2, seed=1237, mask_kind=kind)
ctx = pkg.Context(device_id=0)
ctx.set_option("exclusive_device", 1)
for kv in opts:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.setup(pkg.DataHandler.from_scene(sc))
pkg.alternating_loop(ctx, None, max_outer=1)
ctx.bench_cg(solves=5, iters=101)
rounds = []
for _ in range(7):
    b = ctx.bench_cg(solves=20, iters=101)
    rounds.append(1e6 * b["seconds"] / b["iterations"])
print({"options": opts, "min_us": round(min(rounds), 3), "median_us": round(statistics.median(rounds), 3), "resident": ctx.get_option("cg_resident_active"),
       "rect": ctx.get_option("cg_resident_rect_active")})
ctx.close()
