#!/usr/bin/env python3
"""Occupied tiles against microseconds per CG step: full-frame square grids, sf 4, from 196 to 1024 tiles of
256 x 64 -- the resident kernel up to one tile per CU, the streaming step beyond.   python tools/cliff_curve.py -> JSON lines"""
import importlib, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
for size in (1792, 2048, 2112, 2176, 2304, 2560, 2816, 3072, 3584, 4096):
    sc = pkg.synth.make_scene(size, size, 4, 2, seed=1237, mask_kind="full")
    ctx = pkg.Context(device_id=0)
    ctx.set_option("exclusive_device", 1)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    pkg.alternating_loop(ctx, None, max_outer=1)
    ctx.bench_cg(solves=3, iters=101)
    rounds = [1e6 * b["seconds"] / b["iterations"] for b in (ctx.bench_cg(solves=10, iters=101) for _ in range(3))]
    d = ctx.dims()
    tiles = -(-d["grid_h"] // 256) * -(-d["grid_w"] // 64)
    us = min(rounds)
    print(json.dumps({"grid": [d["grid_h"], d["grid_w"]], "unknowns": d["npix"], "tiles_256x64": tiles, "resident": ctx.get_option("cg_resident_active"),
                      "us_per_step": round(us, 2), "ns_per_unknown_step": round(1e3 * us / d["npix"], 4),
                      "streaming_bytes_GBs": None if ctx.get_option("cg_resident_active") else round(45.0 * d["npix"] / us * 1e-3)}), flush=True)
    ctx.close()
