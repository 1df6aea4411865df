#!/usr/bin/env python3
"""Development aid: wall time of the depth solve as ONE resident launch against the resident kernel on column strips of several
contexts of this process on this one device (srps_strip_group_solve_resident) -- what the group form costs on top of the single launch
when the "peer" memory is the same device's (system-scope stores and polls, the second store of the border tiles, the host's events
and gathers around the launches).   python tools/resident_strips_time.py [h=2048] [w=2048] [sf=4] [ranks=2]"""
import importlib, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("srmeetsps-cuda_amd")
h = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
sf = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ranks = int(sys.argv[4]) if len(sys.argv) > 4 else 2
sc = pkg.synth.make_scene(h, w, sf, 2, seed=1237, mask_kind="full")
dh = pkg.DataHandler.from_scene(sc)

def prep():
    c = pkg.Context(device_id=0)
    c.set_option("cg_resident_tile", 512); c.set_option("exclusive_device", 1)
    c.setup(dh); c.lighting(); c.albedo(); c.depth_partial(); c.synchronize()
    return c

one = prep()
t = []
for _ in range(6):
    one.depth_partial(); one.synchronize()
    t0 = time.perf_counter(); one.depth_solve(); one.synchronize(); t.append(time.perf_counter() - t0)
single_ms = 1e3 * statistics.median(t[1:])
z1 = one.get("z")
grp = [prep() for _ in range(ranks)]
t = []
for _ in range(6):
    for c in grp:
        c.depth_partial(); c.synchronize()
    t0 = time.perf_counter(); pkg.Context.strip_group_solve_resident(grp); t.append(time.perf_counter() - t0)
group_ms = 1e3 * statistics.median(t[1:])
print(json.dumps({"grid": [h, w], "sf": sf, "ranks_on_one_device": ranks, "single_launch_solve_ms": round(single_ms, 4), "group_solve_ms": round(group_ms, 4),
                  "single_us_per_step": round(1e3 * single_ms / 102, 3), "group_us_per_step_upper_bound": round(1e3 * group_ms / 102, 3),
                  "note": "host clock around srps_depth_solve / srps_strip_group_solve_resident (102 passes: residual + 101 steps); the group figure includes the host's tile-list copy, events, three synchronisations and the gather of the strips"}))
one.close()
for c in grp: c.close()
