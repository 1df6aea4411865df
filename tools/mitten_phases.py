#!/usr/bin/env python3
"""Development aid: per-phase device times (srps_get_timings) and the CG step time of the full-frame Mitten solve
(tests/golden/mitten_full.npz).  python tools/mitten_phases.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
M = np.load(os.path.join(ROOT, "tests/golden/mitten_full.npz"))
h, w, sf = int(M["h"]), int(M["w"]), int(M["sf"])
mask = np.unpackbits(M["mask_bits"])[: h * w].astype(np.float32)
mi = np.flatnonzero(mask == 1)
I = np.zeros((8, 3, h * w), np.float32); I[:, :, mi] = M["I_u8"].astype(np.float32) / np.float32(255)
zs = np.zeros((h // sf) * (w // sf), np.float32); zs[M["imasks"]] = M["zs_lr_masked"]
zf = np.zeros(h * w, np.float32); zf[mi] = M["z_full_masked"]
dh = pkg.DataHandler(I=I, mask=mask, K=M["K"], sf=sf, z0=zs.reshape(1, -1), I_h=h, I_w=w, I_c=3, I_n=8, I_n_total=8, zs_lr=zs, z_full=zf)
ctx = pkg.Context(device_id=0)
ctx.set_option("exclusive_device", 1)
ctx.setup(dh); pkg.alternating_loop(ctx, None)
ctx.set_option("phase_timing", 1)
ctx.setup(dh)
for it in range(4):
    pkg.alternating_loop(ctx, None, max_outer=1)
    print(it, {k: round(v * 1e3, 1) for k, v in ctx.timings().items()}, "us")
d = ctx.dims(); print(d, "resident", ctx.get_option("cg_resident_active"), "rect tiles", ctx.get_option("cg_resident_rect_tiles_256"), ctx.get_option("cg_resident_rect_active"))
b = ctx.bench_cg(solves=5, iters=101); print("cg us/step", 1e6 * b["seconds"] / b["iterations"])
