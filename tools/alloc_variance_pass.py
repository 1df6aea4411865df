#!/usr/bin/env python3
"""Development aid: do the phases of a pass (image sweeps, resident CG) depend on where the context's arenas were allocated?  One process,
the headline workload set up again and again on fresh contexts (dummy allocations of varying size in between), phase medians per set-up.
    python tools/alloc_variance_pass.py [setups=6]"""
import importlib, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("srmeetsps-cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sc = pkg.synth.make_scene(2048, 2048, 4, 20, seed=1237, mask_kind="full")
dh = pkg.DataHandler.from_scene(sc)
keep = []
for it in range(n):
    if it % 2 == 1:
        keep.append(torch.empty((37 + 61 * it) * (1 << 20), dtype=torch.uint8, device="cuda"))
    ctx = pkg.Context(device_id=0)
    ctx.set_option("exclusive_device", 1)
    for kv in sys.argv[2:]:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.setup(dh)
    pkg.alternating_loop(ctx, None, max_outer=2)
    ctx.set_option("phase_timing", 1)
    rows = []
    for _ in range(10):
        pkg.alternating_loop(ctx, None, max_outer=1)
        rows.append(ctx.timings())
    ph = {k: round(statistics.median(r[k] for r in rows if k in r), 4) for k in rows[0]}
    print({"setup": it, **ph}, flush=True)
    ctx.close()
