#!/usr/bin/env python3
"""Development aid: where a pass of the headline workload goes, per phase (HIP events of option "phase_timing"), and the pass as a
whole (host clock around K passes, like bench.py).
    python tools/pass_time.py [size=2048] [sf=4] [images=20] [passes=8] [NAME=INT options ...]
Prints one JSON line: median ms per phase, ms per pass."""
import importlib, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("srmeetsps-cuda_amd")
pos = [a for a in sys.argv[1:] if "=" not in a]
opts = [a for a in sys.argv[1:] if "=" in a]
size = int(pos[0]) if len(pos) > 0 else 2048
sf = int(pos[1]) if len(pos) > 1 else 4
n_img = int(pos[2]) if len(pos) > 2 else 20
passes = int(pos[3]) if len(pos) > 3 else 8
sc = pkg.synth.make_scene(size, size, sf, n_img, seed=1234 + 3, mask_kind=os.environ.get("SRPS_MASK", "full"))
if os.environ.get("SRPS_BYTES") == "1":
    import numpy as np
    sc.I = (np.rint(np.clip(sc.I, 0, 1) * 255).astype(np.float32) / np.float32(255)).astype(np.float32)
ctx = pkg.Context(device_id=0)
ctx.set_option("exclusive_device", 1)
for kv in opts:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
ctx.setup(pkg.DataHandler.from_scene(sc))
pkg.alternating_loop(ctx, None, max_outer=2)
torch.cuda.synchronize()
t0 = time.perf_counter()
en = [pkg.alternating_loop(ctx, None, max_outer=1)[0] for _ in range(passes)]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ctx.set_option("phase_timing", 1)
rows = []
for _ in range(passes):
    pkg.alternating_loop(ctx, None, max_outer=1)
    rows.append(ctx.timings())
ph = {k: round(statistics.median(r[k] for r in rows if k in r), 4) for k in rows[0]}
print(json.dumps({"lib": os.path.basename(os.environ.get("SRPS_LIB_PATH", "libsrps_hip.so")), "options": opts, "ms_per_pass": round(1e3 * dt / passes, 4),
                  "phase_ms": ph, "phase_sum_ms": round(sum(ph.values()), 4), "energy": en[-1], "depth_steps": ctx.last_cg_iterations()["depth"]}))
ctx.close()
