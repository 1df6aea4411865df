#!/usr/bin/env python3
"""Development aid: where the eight waves of a block are at the phase boundaries of a resident CG step (library built with
`make -C srmeetsps-cuda_amd/csrc EXTRA=-DSRPS_STAMPS`).  Prints, per stamp, the mean arrival time of every wave relative to the
block's wave 0 at stamp 0 of the same step, and the phase durations per wave."""
import importlib, os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
kind = sys.argv[1] if len(sys.argv) > 1 else "full"
sc = pkg.synth.make_scene(2048, 2048, 4, 2, seed=1237, mask_kind=kind)
ctx = pkg.Context(device_id=0)
ctx.set_option("exclusive_device", 1)
ctx.setup(pkg.DataHandler.from_scene(sc))
pkg.alternating_loop(ctx, None, max_outer=1)
b = ctx.bench_cg(solves=2, iters=101)
print("us/step", round(b["seconds"] * 1e6 / b["iterations"], 2))
buf = np.zeros(64 * 64 * 8 * 16, dtype=np.uint64)
assert ctx.lib.srps_debug_read_wave_stamps(ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(64, 64, 8, 16).astype(np.int64)[8:60] * 10.0          # ns; [step][block][wave][stamp]
order = [0, 1, 2, 3, 14, 15, 4, 11, 5, 6, 12, 13, 7]
names = {0: "step start", 1: "after p-update barrier", 2: "columns done", 3: "ring done (at barrier A)", 14: "after barrier A", 15: "after barrier B (u exchanged)",
         4: "finalize done", 11: "edges published", 5: "sums published + ring requested", 6: "totals in", 12: "x, r updated", 13: "ring awaited", 7: "step end"}
ref = t[:, :, 0:1, 0:1]
rel = t - ref
np.set_printoptions(linewidth=200, precision=0, suppress=True)
print("mean time of each wave at each stamp, ns after wave 0's step start (waves 0..7):")
for sid in order:
    print(f"  {names[sid]:34s}", rel[:, :, :, sid].mean(axis=(0, 1)))
print("phase durations per wave (mean, ns):")
for a_, b_ in zip(order[:-1], order[1:]):
    d = (t[:, :, :, b_] - t[:, :, :, a_]).mean(axis=(0, 1))
    print(f"  {names[a_][:22]:22s} -> {names[b_][:26]:26s}", d)
nxt = t[1:, :, :, 0] - t[:-1, :, :, 0]
print("step (start to next start), mean per wave:", nxt.mean(axis=(0, 1)))
