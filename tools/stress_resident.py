#!/usr/bin/env python3
"""Development aid: repeated solves at several shapes (persistent kernels: looks for rare hangs / non-determinism).
python tools/stress_resident.py [seconds]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd"); pkg.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
shapes = [(2048, 2048, 4, 4, "full"), (300, 200, 2, 3, "ragged"), (1024, 1536, 4, 2, "ellipse"), (260, 130, 1, 2, "ragged"), (64, 48, 4, 3, "full")]
t0 = time.time(); n = 0
ref = {}
while time.time() - t0 < budget:
    for (h, w, sf, nimg, kind) in shapes:
        key = (h, w, sf)
        if key not in ref:
            sc = pkg.synth.make_scene(h, w, sf, nimg, seed=h, mask_kind=kind)
            ref[key] = [pkg.DataHandler.from_scene(sc), None]
        dh = ref[key][0]
        ctx = pkg.Context(device_id=0)
        ctx.setup(dh)
        en = pkg.alternating_loop(ctx, None, max_outer=2)
        z = ctx.get("z")
        ctx.close()
        if ref[key][1] is None: ref[key][1] = (en, z)
        else:
            assert en == ref[key][1][0], (key, en, ref[key][1][0])
            assert np.array_equal(z, ref[key][1][1]), key
        n += 1
print("stress ok:", n, "solves of 2 passes in", round(time.time() - t0, 1), "s, all bit-identical per shape")
