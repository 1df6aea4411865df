#!/usr/bin/env python3
"""Development aid: is the streaming CG step's fast / slow mode a property of the ALLOCATION?  N contexts of the 4096 x 4096 (sf 2) grid
alive at the same time (N different arenas), each timed, then each timed AGAIN in reverse order.   python tools/alloc_lottery.py [N=6]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sc = pkg.synth.make_scene(4096, 4096, 2, 2, seed=1237, mask_kind="full")
dh = pkg.DataHandler.from_scene(sc)
ctxs = []
def t(ctx):
    r = []
    for _ in range(3):
        b = ctx.bench_cg(solves=5, iters=101)
        r.append(1e6 * b["seconds"] / b["iterations"])
    return round(min(r), 1)
for it in range(n):
    ctx = pkg.Context(device_id=0)
    ctx.set_option("exclusive_device", 1)
    ctx.setup(dh)
    pkg.alternating_loop(ctx, None, max_outer=1)
    ctx.bench_cg(solves=2, iters=101)
    ctxs.append(ctx)
    print({"context": it, "us_first": t(ctx), "z_ptr": hex(ctx.device_ptr("z")[0])}, flush=True)
for it in reversed(range(n)):
    print({"context": it, "us_again": t(ctxs[it])}, flush=True)
for c in ctxs:
    c.close()
