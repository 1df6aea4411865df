"""Development aid: depth phase with the resident CG against the streaming CG on a few shapes (python tools/resident_check.py, on a GPU box)."""
import importlib, sys, time, numpy as np
sys.path.insert(0, ".")
pkg = importlib.import_module("srmeetsps-cuda_amd"); pkg.load()
import torch
def run(h, w, sf, n, res, kind="full", nch=3):
    sc = pkg.synth.make_scene(h, w, sf, n, seed=5, mask_kind=kind, n_ch=nch)
    ctx = pkg.Context(device_id=0)
    ctx.set_option("cg_resident", res)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    ctx.lighting(); ctx.albedo()
    e = ctx.depth()
    z = ctx.get("z"); it = ctx.last_cg_iterations()["depth"]
    ctx.close()
    return e, z, it
for (h, w, sf, kind, nch) in [(64, 48, 4, "full", 3), (300, 200, 2, "ragged", 3), (512, 640, 4, "ellipse", 3), (260, 130, 1, "ragged", 1), (1024, 1024, 4, "full", 3)]:
    e0, z0, i0 = run(h, w, sf, 4, 0, kind, nch)
    e1, z1, i1 = run(h, w, sf, 4, 1, kind, nch)
    print(h, w, sf, kind, nch, "energy", e0, e1, "iters", i0, i1, "max|dz|", float(np.abs(z0 - z1).max()), "rmse", float(np.sqrt(np.mean((z0 - z1) ** 2))), flush=True)
