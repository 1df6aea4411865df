"""Does what a process did before (allocations, other contexts) change the 4096^2 streaming CG step?  Run by hand on the GPU
box (python tools/fragmentation_check.py); nothing here runs at import."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def leg(pkg, tag):
    sc4 = pkg.synth.make_scene(4096, 4096, 2, 2, seed=1239, mask_kind="full")
    c4 = pkg.Context(device_id=0)
    c4.setup(pkg.DataHandler.from_scene(sc4))
    pkg.alternating_loop(c4, None, max_outer=1)
    b = c4.bench_cg(solves=3, iters=101)
    print(tag, "4096^2 us/step", 1e6 * b["seconds"] / b["iterations"], flush=True)
    c4.close()


def main():
    import torch
    pkg = importlib.import_module("srmeetsps-cuda_amd")
    leg(pkg, "fresh process")
    # now what bench.py does before its 4096 leg: a 2048^2 context with 20 images, a second context, torch allocations
    sc = pkg.synth.make_scene(2048, 2048, 4, 20, seed=1237, mask_kind="full")
    dh = pkg.DataHandler.from_scene(sc)
    ctx = pkg.Context(device_id=0); ctx.setup(dh); pkg.alternating_loop(ctx, None, max_outer=2)
    leg(pkg, "after a 2048^2 x 20 context (still alive)")
    c2 = pkg.Context(device_id=0); c2.set_option("cg_resident", 0); c2.setup(dh); pkg.alternating_loop(c2, None, max_outer=1); c2.close()
    leg(pkg, "after a second context was created and closed")
    ctx.close()
    leg(pkg, "after closing everything")
    t = torch.empty(256 * 1024 * 1024, device="cuda"); del t
    leg(pkg, "after a 1 GiB torch allocation")


if __name__ == "__main__":
    main()
