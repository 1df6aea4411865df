#!/usr/bin/env python3
"""Development aid: a few whole alternating passes of the headline workload (for rocprofv3 --kernel-trace / --pmc on the image sweeps).
    python tools/pass_prof.py [size=2048] [sf=4] [images=20] [passes=3] [NAME=INT options ...]      e.g. albedo_mode=2"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
pos = [a for a in sys.argv[1:] if "=" not in a]
opts = [a for a in sys.argv[1:] if "=" in a]
size = int(pos[0]) if len(pos) > 0 else 2048
sf = int(pos[1]) if len(pos) > 1 else 4
n_img = int(pos[2]) if len(pos) > 2 else 20
passes = int(pos[3]) if len(pos) > 3 else 3
sc = pkg.synth.make_scene(size, size, sf, n_img, seed=1234 + 3, mask_kind=os.environ.get("SRPS_MASK", "full"))
ctx = pkg.Context(device_id=0)
ctx.set_option("exclusive_device", 1)
for kv in opts:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
ctx.setup(pkg.DataHandler.from_scene(sc))
en = pkg.alternating_loop(ctx, None, max_outer=passes)
print(en, ctx.last_cg_iterations())
ctx.close()
