"""Where srps_setup's time goes at the metric's configuration (2048 x 2048, sf 4, 20 images): floats and bytes, first and repeated
set-ups on one context (SRPS_SETUP_TIMING=1 prints the library's own breakdown to stderr).
python tools/setup_time.py [size] [images] [NAME=INT options, e.g. pin_uploads=0]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SRPS_SETUP_TIMING", "1")


def main():
    import torch
    opts = [a for a in sys.argv[1:] if "=" in a]
    sys.argv = [a for a in sys.argv if "=" not in a]
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    pkg = importlib.import_module("srmeetsps-cuda_amd")
    sc = pkg.synth.make_scene(size, size, 4, n_img, seed=1237, mask_kind="full")
    dh = pkg.DataHandler.from_scene(sc)
    k = np.rint(np.clip(sc.I, 0, 1) * 255).astype(np.uint8)
    dh8 = pkg.DataHandler.from_scene(sc); dh8.I = None; dh8.I_u8 = k
    for name, d in (("floats", dh), ("bytes", dh8)):
        ctx = pkg.Context(device_id=0)
        ctx.set_option("exclusive_device", 1)
        for kv in opts:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.setup(d)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            en = pkg.alternating_loop(ctx, None)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"{name} set-up {rep}: {1e3 * (t1 - t0):.2f} ms, solve {1e3 * (t2 - t1):.2f} ms ({len(en)} passes), bytes store {ctx.get_option('image_store_bytes_active')}", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
