import importlib, sys, numpy as np
sys.path.insert(0, ".")
pkg = importlib.import_module("srmeetsps-cuda_amd"); pkg.load()
f32=np.float32
h, w = 24, 28
m = np.zeros((h, w)); m[10:12, 14:16] = 1
sc = pkg.synth.make_scene(h, w, 2, 3, seed=5, n_ch=3, mask_kind="full")
sc.mask = pkg.synth.to_cm(m).astype(f32)
for one in (0, 1):
  for it in (0, 1, 2, 3, 5, 100):
    ctx = pkg.Context(device_id=0)
    ctx.set_option("cg_one_sync", one); ctx.set_option("cg_max_iter", it)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    ctx.lighting(); ctx.albedo()
    e = ctx.depth()
    print("one_sync", one, "max_iter", it, "energy", e, "iters", ctx.last_cg_iterations()["depth"], "z", ctx.get("z"), "rho", ctx.get("rho")[:4], "s", ctx.get("s")[:4])
    ctx.close()
