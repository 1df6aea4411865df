#!/usr/bin/env python3
"""Development aid: wall time of full solves (set-up + alternating loop to the reference's stop rule) at a given shape.
python tools/solve_time.py H W sf images [mask_kind]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
H, W, sf, n = (int(v) for v in sys.argv[1:5])
kind = sys.argv[5] if len(sys.argv) > 5 else "disc"
sc = pkg.synth.make_scene(H, W, sf, n, seed=77, mask_kind=kind)
dh = pkg.DataHandler.from_scene(sc)
ctx = pkg.Context(device_id=0)
for rep in range(3):
    t0 = time.perf_counter()
    ctx.setup(dh)
    ctx.synchronize()
    t1 = time.perf_counter()
    en = pkg.alternating_loop(ctx, None)
    ctx.synchronize()
    t2 = time.perf_counter()
    d = ctx.dims()
    print(f"{H}x{W} sf{sf} {n} images, mask {kind}: P={d['npix']}  set-up {1e3*(t1-t0):.1f} ms, loop {1e3*(t2-t1):.2f} ms, {len(en)} passes, "
          f"{1e3*(t2-t1)/len(en):.3f} ms/pass, resident CG: {bool(ctx.get_option('cg_resident_active'))}")
ctx.close()
