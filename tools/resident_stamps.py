#!/usr/bin/env python3
"""Development aid: where one step of the resident CG kernel spends its time.  Needs a library built with
`make -C srmeetsps-cuda_amd/csrc EXTRA=-DSRPS_STAMPS` (thread 0 of every block stamps s_memrealtime at the phase
boundaries of every step; kernels_resident.hip); prints the mean / slowest-block / fastest-block duration of every phase
at 2048 x 2048, sf 4, and the latencies of the grid-wide reduction.  Rebuild without EXTRA afterwards."""
import importlib, os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("srmeetsps-cuda_amd")
opts = [a for a in sys.argv[1:] if "=" in a]            # NAME=INT options; a bare word: the mask kind
kind = next((a for a in sys.argv[1:] if "=" not in a), "full")
sc = pkg.synth.make_scene(2048, 2048, 4, 2, seed=1237, mask_kind=kind)
ctx = pkg.Context(device_id=0)
for kv in opts:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.setup(pkg.DataHandler.from_scene(sc))
pkg.alternating_loop(ctx, None, max_outer=1)
b = ctx.bench_cg(solves=2, iters=101)
print("us/step", round(b["seconds"] * 1e6 / b["iterations"], 2))
lib = ctx.lib
buf = np.zeros(128 * 256 * 16, dtype=np.uint64)
rc = lib.srps_debug_read_stamps(ctypes.c_void_p(buf.ctypes.data))
assert rc == 0, rc
t = buf.reshape(128, 256, 16).astype(np.int64)[10:100] * 10.0      # ns
names = ["p-update+barrier (0->1)", "columns (1->2)", "ring rows/cols (2->3)", "u exchange+finalize (3->4)", "publish (4->5)", "collect wait (5->6)", "update+ring await (6->7)"]
for i, n in enumerate(names):
    d = t[:, :, i + 1] - t[:, :, i]
    print(f"{n:32s} mean {d.mean():8.0f} ns   mean-of-max-over-blocks {d.max(axis=1).mean():8.0f}   min {d.min(axis=1).mean():8.0f}")
step = t[1:, :, 0] - t[:-1, :, 0]
print("step (0->0 next)", step.mean())
# compute end spread and publish->collect latency
t4 = t[:, :, 4]; t5 = t[:, :, 5]; t6 = t[:, :, 6]
print("spread of compute end over blocks (max-min)", (t4.max(axis=1) - t4.min(axis=1)).mean(), " (max-median)", (t4.max(axis=1) - np.median(t4, axis=1)).mean())
print("last publish -> first collect", (t6.min(axis=1) - t5.max(axis=1)).mean(), " -> median collect", (np.median(t6, axis=1) - t5.max(axis=1)).mean(), " -> last collect", (t6.max(axis=1) - t5.max(axis=1)).mean())
# which blocks are last
last = t4.argmax(axis=1)
print("blocks finishing last (top):", np.bincount(last, minlength=256).argsort()[::-1][:8], np.sort(np.bincount(last, minlength=256))[::-1][:8])
own = (t4 - t[:, :, 0])
print("compute per block 0->4: mean", own.mean(), "max-block mean", own.mean(axis=0).max(), "min-block mean", own.mean(axis=0).min())
def seg(a, b, name):
    d = t[:, :, b] - t[:, :, a]
    print(f"{name:40s} mean {d.mean():8.0f}  max-over-blocks {d.max(axis=1).mean():8.0f}  min {d.min(axis=1).mean():8.0f}")
if t[:, :, 8].any():
    seg(4, 11, "edges published (4->11)")
    seg(11, 8, "sums: totals+barrier+store (11->8)")
    seg(8, 5, "ring requested (8->5)")
    seg(5, 9, "first poll round returned (5->9)")
    seg(9, 10, "retries (9->10)")
    seg(10, 6, "totals + barrier (10->6)")
    seg(6, 12, "x, r update (6->12)")
    seg(12, 13, "ring await, thread 0 (12->13)")
    seg(13, 7, "r.r, beta, next p, ring p (13->7)")
    if t[:, :, 14].any():
        seg(3, 14, "barrier behind the ring (general body only) (3->14)"); seg(14, 15, "u to LDS + barrier (14->15)"); seg(15, 4, "u in, finalize (15->4)")
    t8 = t[:, :, 8]; t9 = t[:, :, 9]; t10 = t[:, :, 10]
    print("last store issue -> median retries-done", (np.median(t10, axis=1) - t8.max(axis=1)).mean(), " -> first", (t10.min(axis=1) - t8.max(axis=1)).mean(), " -> last", (t10.max(axis=1) - t8.max(axis=1)).mean())
    print("spread of store issue (max-min)", (t8.max(axis=1) - t8.min(axis=1)).mean(), "(max-median)", (t8.max(axis=1) - np.median(t8, axis=1)).mean())
ctx.close()
# round 5: which blocks are systematically slow?  per-block mean of the compute span (stamps 0 -> 4), the ten slowest and fastest with their
# XCD (block & 7), tile and the tile's position (row br of 8, column bc of 32 at 2048 x 2048)
import numpy as _np
ownm = (t[:, :, 4] - t[:, :, 0]).mean(axis=0)
order = _np.argsort(ownm)
def _where(b):
    nwg = 256; q = nwg >> 3; xcd = b & 7; kk = b >> 3; tile = xcd * q + kk
    return f"block {b:3d} xcd {xcd} tile {tile:3d} (br {tile % 8}, bc {tile // 8}) {ownm[b]:6.0f} ns"
print("slowest:", "; ".join(_where(int(b)) for b in order[::-1][:10]))
print("fastest:", "; ".join(_where(int(b)) for b in order[:10]))
print("per-XCD mean of the compute span:", [round(float(ownm[_np.arange(256) % 8 == x].mean())) for x in range(8)])
print("by tile row br:", [round(float(_np.mean([ownm[b] for b in range(256) if (((b & 7) * 32 + (b >> 3)) % 8) == r]))) for r in range(8)])
print("by tile column bc (groups of 4):", [round(float(_np.mean([ownm[b] for b in range(256) if (((b & 7) * 32 + (b >> 3)) // 8) // 4 == g]))) for g in range(8)])
