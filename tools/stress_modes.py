"""Repeats whole solves at 1024 x 1024 (resident CG with 256 tiles, persistent albedo CG, cooperative launches) in the three
albedo modes, many contexts in one process: looks for intermittent failures.  python tools/stress_modes.py [repeats]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    pkg = importlib.import_module("srmeetsps-cuda_amd")
    sc = pkg.synth.make_scene(1024, 1024, 4, 6, seed=1024 + 1024 + 6, mask_kind="full")
    dh = pkg.DataHandler.from_scene(sc)
    ref = {}
    for i in range(reps):
        for mode in (2, 1, 0):
            ctx = pkg.Context(device_id=0)
            ctx.set_option("albedo_mode", mode)
            srps = pkg.SRPS(dh, ctx=ctx)
            en = srps.execute(max_outer=3)
            z = srps.z(); rho = srps.rho()
            fb = ctx.get_option("persistent_fallbacks")
            ctx.close()
            key = (mode,)
            if key not in ref:
                ref[key] = (en, z, rho)
            else:
                same = en == ref[key][0] and np.array_equal(z, ref[key][1]) and np.array_equal(rho, ref[key][2])
                if not same or fb:
                    print(f"rep {i} mode {mode}: same bits {same}, fallbacks {fb}", flush=True)
        if i % 5 == 0:
            print(f"rep {i} done", flush=True)
    print("stress_modes: finished", flush=True)


if __name__ == "__main__":
    main()
