"""BASELINE.json config 2 at its true size ("mitten_sf2 (sf=2, ~8 images) on 1 x MI355X, full alternating solve to
convergence"): the WHOLE frame of the reference's bundled Mitten data set (960 x 1280, 148 600 masked pixels, the first 8 images,
real 16-bit depth, real mask), committed as tests/golden/mitten_full.npz by tests/golden/make_mitten_full.py -- masked samples
only, the images as the bytes the PNGs hold.  Expected outputs: the oracle's faithful restatement.  Depth is in the data's
units (~700), so the north_star tolerance applies to the RELATIVE RMSE.

Round 6: parametrised over 8 and ALL 20 images.  `srps -t images -d dataset/Images/Mitten` -- the reference's own CLI on its bundled folder --
globs the whole RGB/ directory (Utilities.cpp:349-352): 20 images, in cv::glob's lexicographic order (I_1, I_10, ..., I_19, I_2, I_20,
I_3, ...).  tests/golden/mitten_full_20.npz adds the masked bytes of images 9 - 20 of that order and the oracle's outputs of the
20-image solve (5 passes)."""
import os
import time

import numpy as np
import pytest

f32 = np.float32
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mitten_full.npz")
PATH20 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mitten_full_20.npz")
G8 = np.load(PATH)
G20 = np.load(PATH20)
G = G8


def _expected(n):
    return G8 if n == 8 else G20


def _inputs(oracle, n=8):
    h, w, sf = int(G8["h"]), int(G8["w"]), int(G8["sf"])
    mask = np.unpackbits(G8["mask_bits"])[: h * w].astype(f32)
    geo = oracle.build_geometry(h, w, sf, mask)
    bytes_ = G8["I_u8"] if n == 8 else np.concatenate([G8["I_u8"], G20["I_u8_9_to_20"]])
    assert bytes_.shape[0] == n
    I = np.zeros((n, 3, h * w), f32)
    I[:, :, geo.imask] = bytes_.astype(f32) / f32(255)                     # what the image loader produces (Utilities.cpp:343)
    assert np.array_equal(geo.imasks, G8["imasks"])
    zs = np.zeros((h // sf) * (w // sf), f32); zs[geo.imasks] = G8["zs_lr_masked"]
    zf = np.zeros(h * w, f32); zf[geo.imask] = G8["z_full_masked"]
    return h, w, sf, mask, G8["K"], I, zs, zf, geo


def rel_rmse(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


def test_fixture_is_self_consistent(oracle):
    h, w, sf, mask, K, I, zs, zf, geo = _inputs(oracle)
    assert (h, w, sf) == (960, 1280, 2)
    assert (geo.npix, geo.npixs) == (int(G["npix"]), int(G["npixs"])) == (148600, 36915)
    assert G["I_u8"].shape == (8, 3, geo.npix)
    assert G["final_z"].shape == (geo.npix,) and G["final_rho"].shape == (3, geo.npix)
    assert 400 < float(G["z_full_masked"].mean()) < 9870 and int(G["n_outer"]) == len(G["energies"]) == 4
    assert np.all(np.diff(G["energies"]) < 0)                              # the reference's stop rule ended a converging run
    assert G20["I_u8_9_to_20"].shape == (12, 3, geo.npix) and G20["final_z"].shape == (geo.npix,)
    assert int(G20["n_outer"]) == len(G20["energies"]) == 5 and np.all(np.diff(G20["energies"]) < 0)


def test_c_oracle_agrees_with_the_20_image_fixture(oracle):
    """the fixture's expected outputs are the numpy oracle's; the C oracle's assembled depth step (written separately) from the same start
    reproduces the first pass's energy -- a regression pin of the 20-image inputs as much as of either oracle"""
    import c_oracle as CO
    h, w, sf, mask, K, I, zs, zf, geo = _inputs(oracle, 20)
    st = oracle.setup(oracle.Problem(h, w, sf, mask, K, I, zs, zf))
    oracle.lighting_estimation(st.s, st.rho, st.N, st.I)
    oracle.albedo_estimation(st.s, st.rho, st.N, st.I)
    cs = CO.Structure(h, w, sf, mask)
    z = st.z.copy()
    e, it = CO.depth_estimation(cs, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z, st.fx, st.fy, True)
    assert it == 101 and abs(e - float(G20["energies"][0])) <= 1e-3 * float(G20["energies"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("n_images", [8, 20])
def test_hip_full_mitten_solve_to_convergence(pkg, oracle, n_images):
    G = _expected(n_images)
    h, w, sf, mask, K, I, zs, zf, geo = _inputs(oracle, n_images)
    dh = pkg.DataHandler(I=I, mask=mask, K=K, sf=sf, z0=zs.reshape(1, -1), I_h=h, I_w=w, I_c=3, I_n=n_images, I_n_total=n_images, zs_lr=zs, z_full=zf)
    ctx = pkg.Context(device_id=0)
    srps = pkg.SRPS(dh, ctx=ctx)
    en = srps.execute()                                                    # includes the set-up (upload of the images)
    t0 = time.perf_counter(); en = srps.execute(); dt = time.perf_counter() - t0
    print(f"Mitten, full frame, {n_images} images: {len(en)} passes, {1e3 * dt:.1f} ms with set-up; resident CG {ctx.get_option('cg_resident_active')}, "
          f"bytes {ctx.get_option('image_store_bytes_active')}")
    assert ctx.get_option("cg_resident_active") == 1
    occ, tot = ctx.get_option("cg_resident_tiles_occupied_16"), ctx.get_option("cg_resident_tiles_16")
    print(f"Mitten: {occ} of the bounding box's {tot} tiles of 256 x 16 hold a masked pixel: that many blocks")
    assert 0 < occ < tot
    assert ctx.get_option("image_store_bytes_active") == (1 if geo.npix % 4 == 0 else 0)
    assert len(en) == int(G["n_outer"]), (en, G["energies"])
    print("Mitten: energies relative deviation per pass", [abs(a - b) / b for a, b in zip(en, G["energies"])], "depth rel. RMSE", rel_rmse(srps.z(), G["final_z"]),
          "albedo max abs", float(np.abs(srps.rho() - G["final_rho"]).max()))
    # measured (round 3): every pass's energy within 2.4e-5, depth 1.8e-6 relative, albedo 1.0e-3 -- the asserts allow three times that
    np.testing.assert_allclose(en, G["energies"], rtol=8e-5)
    assert rel_rmse(srps.z(), G["final_z"]) < 1e-4                        # north_star's bar; relative: depth ~ 700, ulp(700) = 6e-5
    assert rel_rmse(srps.z(), G["final_z"]) < 6e-6                        # ... and what a regression would have to stay under
    # WHY the albedo is only good to 1e-3 here when the depth agrees to 2e-6 (round-3 review, weak 1b) -- found in round 4
    # (tools/albedo_mode_compare.py): not the albedo solve (the diagonal system's condition number is 2 - 3 on this data; CG and
    # fixed point agree to 4e-7 per step), and not a scale exchanged between albedo and lighting (the fitted scale is 1 to 1e-7).
    # It is the DEPTH's rounding reaching the normals: N = (fx zx, fy zy, ...) / dz takes finite differences of z (~ 550 here,
    # 1 ulp = 6e-5) and multiplies them by the focal length (1217): two runs whose depths differ by d differ in the normals by up to
    # 2 fx d / dz, in the shading by about as much, and in the albedo by that times rho.  So the albedo's tolerance FOLLOWS from the
    # depth's deviation:
    dzn = ctx.get("dz")
    d_abs = float(np.abs(srps.z().astype(np.float64) - G["final_z"]).max())
    amp = 2.0 * float(K[0]) * d_abs / float(dzn.min())
    alb = float(np.abs(srps.rho() - G["final_rho"]).max())
    print(f"Mitten: depth max-abs deviation {d_abs:.3e} (z ~ {float(np.median(srps.z())):.0f}), fx {float(K[0]):.1f}, min dz {float(dzn.min()):.1f}: normals may differ by "
          f"{amp:.3e}; albedo max-abs {alb:.3e} = {alb / (amp * float(srps.rho().max())):.2f} x that bound x max rho, RMSE {float(np.sqrt(np.mean((srps.rho() - G['final_rho']) ** 2))):.2e}")
    assert alb < 3e-3
    # (measured, round 4: depth max-abs 6.8e-3 -> the normals may differ by 3.1e-2; the albedo's 1.15e-3 is 3 % of that worst case)
    assert alb <= 0.25 * amp * float(srps.rho().max())                     # the albedo deviates no more than the depth's deviation explains
    assert float(np.sqrt(np.mean((srps.rho() - G["final_rho"]) ** 2))) < 2e-4
    ctx.close()
