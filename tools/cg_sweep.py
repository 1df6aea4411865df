#!/usr/bin/env python3
"""Development aid: time the depth-CG kernels for a few kernel options on one synthetic grid.
   python tools/cg_sweep.py [size] [sf]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
size = sys.argv[1] if len(sys.argv) > 1 else "2048"
hh, ww = (int(v) for v in size.split("x")) if "x" in size else (int(size), int(size))
sf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc = pkg.synth.make_scene(hh, ww, sf, 2, seed=1237, mask_kind="full")
ctx = pkg.Context(device_id=0)
ctx.set_option("cg_resident", 0)
ctx.setup(pkg.DataHandler.from_scene(sc))
pkg.alternating_loop(ctx, None, max_outer=1)
opts = [("march_snake", 0), ("march_snake", 1)] + [("march_strip", int(v)) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,16,20".split(","))]
for name, v in opts:
    ctx.set_option(name, v)
    if name == 'tensor_recompute':
        pkg.alternating_loop(ctx, None, max_outer=1)     # re-assemble in the new form
    for rep in range(2):
        b = ctx.bench_cg(solves=5, iters=101)
    us = 1e6 * b["seconds"] / b["iterations"]
    P = hh * ww
    upd = f"update {b['update_us']:6.2f} us ({b['update_bytes']/b['update_us']/1e3:6.0f} GB/s)  " if b['update_us'] > 0 else ""
    print(f"[{hh}x{ww}] {1e3*us/P:6.3f} ps/unknown/iter | {name}={v} (strip {ctx.get_option('march_strip')}): {us:7.2f} us/iter  apply {b['apply_us']:6.2f} us ({b['apply_bytes']/b['apply_us']/1e3:6.0f} GB/s)  "
          f"{upd}loop {(b['apply_bytes']+b['update_bytes'])/us/1e3:6.0f} GB/s", flush=True)
ctx.close()
