#!/usr/bin/env python3
"""Generates tests/golden/srps_small.npz: inputs and per-phase expected outputs of the oracle's
*faithful* restatement (oracle/srps_oracle.py) on a small ragged-mask scene.

The reference ships no golden vectors and cannot run here, so this fixture pins the
ORACLE (regression) and gives the GPU tests a committed input/output pair; it is not a reference
run.  Re-run:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import srps_oracle as O  # noqa: E402

synth = importlib.import_module("srmeetsps-cuda_amd.synth")


def main():
    sc = synth.make_scene(24, 32, 2, 4, seed=77, mask_kind="ragged")
    prob = O.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init)
    st = O.setup(prob)
    out = dict(h=sc.h, w=sc.w, sf=sc.sf, mask=sc.mask, K=sc.K, I_full=sc.I.astype(np.float16).astype(np.float32),
               zs_lr=sc.zs_lr, z_full=sc.z_init)
    # images are stored as float16-rounded values to keep the fixture small; recompute from those
    prob = O.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, out["I_full"], sc.zs_lr, sc.z_init)
    st = O.setup(prob)
    out.update(imask=st.geo.imask, imasks=st.geo.imasks, z_init=st.z.copy(), N_init=st.N.copy(), dz_init=st.dz.copy(),
               xx=st.xx, yy=st.yy, z0s=st.z0s)
    li = []; O.lighting_estimation(st.s, st.rho, st.N, st.I, cg_iters=li); out["s_after_lighting"] = st.s.copy()
    ai = []; O.albedo_estimation(st.s, st.rho, st.N, st.I, cg_iters=ai); out["rho_after_albedo"] = st.rho.copy()
    tr = []
    e = O.depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, st.z, st.fx, st.fy, cg_trace=tr)
    out["z_after_depth"] = st.z.copy(); out["energy_1"] = np.float64(e)
    out["cg_trace_k"] = np.array([t[0] for t in tr]); out["cg_trace_r1"] = np.array([t[1] for t in tr]); out["cg_trace_alpha"] = np.array([t[2] for t in tr])
    out["lighting_cg_iters"] = np.array(li); out["albedo_cg_iters"] = np.array(ai)
    full = O.execute(prob, depth="faithful")
    out.update(final_z=full.z, final_rho=full.rho, final_s=full.s, final_N=full.N, energies=np.array(full.energies), n_outer=full.iterations)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "srps_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; energies", full.energies)


if __name__ == "__main__":
    main()
