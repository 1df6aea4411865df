#!/usr/bin/env python3
"""Development aid for tools/ab_variants.sh: a depth phase (default: 2048 x 2048, sf 4, full mask and ellipse; or `H W sf kind ...` quadruples) with
the library in place -- prints a digest of z and the energy, so that two builds can be compared for bit-identity."""
import hashlib, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
args = sys.argv[1:]
cases = [(2048, 2048, 4, "full"), (2048, 2048, 4, "ellipse")] if not args else [(int(args[i]), int(args[i + 1]), int(args[i + 2]), args[i + 3]) for i in range(0, len(args), 4)]
for h, w, sf, kind in cases:
    sc = pkg.synth.make_scene(h, w, sf, 2, seed=1237, mask_kind=kind)
    for one_sync in (1, 0):
        ctx = pkg.Context(device_id=0)
        ctx.set_option("exclusive_device", 1)
        ctx.set_option("cg_one_sync", one_sync)
        ctx.setup(pkg.DataHandler.from_scene(sc))
        ctx.lighting(); ctx.albedo()
        e = ctx.depth()
        z = ctx.get("z")
        print(h, w, sf, kind, "one_sync", one_sync, "energy", repr(e), "z digest", hashlib.sha1(z.tobytes()).hexdigest()[:16], "iters", ctx.last_cg_iterations()["depth"],
              "resident", ctx.get_option("cg_resident_active"), "fallbacks", ctx.get_option("persistent_fallbacks"))
        ctx.close()
