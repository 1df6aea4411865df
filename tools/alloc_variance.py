#!/usr/bin/env python3
"""Development aid: does the streaming CG step's time depend on WHERE its planes were allocated?  One process, the 4096 x 4096 (sf 2) grid
set up again and again (a fresh context = a fresh arena each time, with a dummy allocation of a varying size in between to move it),
20 solves of 101 steps timed per set-up.   python tools/alloc_variance.py [size=4096] [sf=2] [setups=8]"""
import importlib, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("srmeetsps-cuda_amd")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sf = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc = pkg.synth.make_scene(size, size, sf, 2, seed=1237, mask_kind="full")
dh = pkg.DataHandler.from_scene(sc)
keep = []
for it in range(n):
    if it % 2 == 1:
        keep.append(torch.empty((37 + 61 * it) * (1 << 20), dtype=torch.uint8, device="cuda"))      # moves the next arena
    ctx = pkg.Context(device_id=0)
    ctx.set_option("exclusive_device", 1)
    for kv in sys.argv[4:]:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.setup(dh)
    pkg.alternating_loop(ctx, None, max_outer=1)
    ctx.bench_cg(solves=3, iters=101)
    r = []
    for _ in range(4):
        b = ctx.bench_cg(solves=10, iters=101)
        r.append(1e6 * b["seconds"] / b["iterations"])
    ptr = ctx.device_ptr("z")[0]
    print({"setup": it, "min_us": round(min(r), 2), "max_us": round(max(r), 2), "z_ptr_mod_2MB": hex(ptr % (1 << 21)), "z_ptr": hex(ptr)}, flush=True)
    ctx.close()
