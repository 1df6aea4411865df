#!/usr/bin/env python3
"""Development aid: run a few CG solves (for rocprofv3 --pmc / --kernel-trace).  python tools/cg_prof.py HxW sf recompute strip [iters] [resident] [resident_debug]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
size = sys.argv[1]; hh, ww = (int(v) for v in size.split("x")) if "x" in size else (int(size), int(size))
sf = int(sys.argv[2]); rec = int(sys.argv[3]); strip = int(sys.argv[4])
sc = pkg.synth.make_scene(hh, ww, sf, 2, seed=int(os.environ.get("SRPS_SEED", "1237")), mask_kind="full")
ctx = pkg.Context(device_id=0)
ctx.set_option("tensor_recompute", rec); ctx.set_option("march_strip", strip)
if len(sys.argv) > 6: ctx.set_option("cg_resident", int(sys.argv[6]))
if len(sys.argv) > 7: ctx.set_option("cg_resident_debug", int(sys.argv[7]))
ctx.setup(pkg.DataHandler.from_scene(sc))
pkg.alternating_loop(ctx, None, max_outer=1)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
b = ctx.bench_cg(solves=2, iters=iters)
print(b)
ctx.close()
