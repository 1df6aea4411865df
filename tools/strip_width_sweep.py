#!/usr/bin/env python3
"""Development aid: one rank's compute per step of the strip-partitioned CG at 4096 x 4096, sf 2, as a function of the marching
kernel's strip width (option march_strip), with the ranks as contexts of this process on one device.
python tools/strip_width_sweep.py [ranks=8] [widths...]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("srmeetsps-cuda_amd")
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
widths = [int(v) for v in sys.argv[2:]] or [0, 4, 8, 12, 16, 24, 32]
sc = pkg.synth.make_scene(4096, 4096, 2, 2, seed=1238, mask_kind="full")
dh = pkg.DataHandler.from_scene(sc)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    group = []
    for _ in range(ranks):
        c = pkg.Context(device_id=0)
        c.set_stream(stream.cuda_stream)
        c.set_option("exclusive_device", 1)
        c.setup(dh)
        c.lighting(); c.albedo(); c.depth_partial()
        group.append(c)
    for wd in widths:
        for c in group:
            c.set_option("march_strip", wd)
        pkg.Context.strip_group_solve(group)
        for c in group:
            c.depth_partial()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pkg.Context.strip_group_solve(group)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for c in group:
            c.depth_partial()
        print(f"march_strip {wd:3d}: {1e6 * dt / (101 * ranks):7.2f} us per step and rank ({ranks} ranks)", flush=True)
    for c in group:
        c.close()
