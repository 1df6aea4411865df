#!/usr/bin/env python3
"""Development aid for tools/ab_variants.sh: a depth phase at 2048 x 2048, sf 4 (full mask and ellipse) with the library in place --
prints a digest of z and the energy, so that two builds can be compared for bit-identity."""
import hashlib, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
for kind in ("full", "ellipse"):
    sc = pkg.synth.make_scene(2048, 2048, 4, 2, seed=1237, mask_kind=kind)
    ctx = pkg.Context(device_id=0)
    ctx.set_option("exclusive_device", 1)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    ctx.lighting(); ctx.albedo()
    e = ctx.depth()
    z = ctx.get("z")
    print(kind, "energy", repr(e), "z digest", hashlib.sha1(z.tobytes()).hexdigest()[:16], "iters", ctx.last_cg_iterations()["depth"], "fallbacks", ctx.get_option("persistent_fallbacks"))
    ctx.close()
