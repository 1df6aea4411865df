#!/usr/bin/env python3
"""Development aid: milliseconds per alternating pass (2048 x 2048, sf 4, 20 images) with the host sleeping on the stream at the
pass's one wait (host_wait_spin 0) and polling it (1, the default), alternately, three rounds."""
import importlib, sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("srmeetsps-cuda_amd")
sc = pkg.synth.make_scene(2048, 2048, 4, 20, seed=1235, mask_kind="full")
dh = pkg.DataHandler.from_scene(sc)
for rep in range(3):
    for spin in (0, 1):
        ctx = pkg.Context(device_id=0)
        ctx.set_option("exclusive_device", 1)
        ctx.set_option("host_wait_spin", spin)
        ctx.setup(dh)
        pkg.alternating_loop(ctx, None, max_outer=2)
        import torch; torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): pkg.alternating_loop(ctx, None, max_outer=1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("host_wait_spin", spin, "ms per pass", round(dt * 1e3, 4))
        ctx.close()
