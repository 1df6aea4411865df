"""BASELINE.json config 2 ("mitten_sf2"): a 256 x 192 window of the reference's bundled Mitten data set
(8 images, sf 2, real 16-bit depth with invalid samples, real mask edge), committed as
tests/golden/mitten_crop.npz by tests/golden/make_mitten_crop.py.  Depth is in the data's units
(~700), so the north_star tolerance is applied to the RELATIVE RMSE."""
import os
import numpy as np
import pytest

f32 = np.float32
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mitten_crop.npz")
G = np.load(PATH)


def _inputs():
    I = G["I_u8"].astype(f32) / f32(255)                                   # what the image loader produces (Utilities.cpp:343)
    return int(G["h"]), int(G["w"]), int(G["sf"]), G["mask"].astype(f32), G["K"], I, G["zs_lr"], G["z_full"]


def rel_rmse(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


def test_fixture_is_self_consistent(oracle):
    h, w, sf, mask, K, I, zs, zf = _inputs()
    geo = oracle.build_geometry(h, w, sf, mask)
    assert (geo.npix, geo.npixs) == (int(G["npix"]), int(G["npixs"]))
    assert 0 < geo.npix < h * w                                           # the window straddles the mask edge
    assert I.shape == (8, 3, h * w) and zs.size == (h // sf) * (w // sf) and zf.size == h * w
    assert (G["z0_u16"] == 0).any()                                       # invalid depth samples are present (inpainted)
    assert 400 < float(zf[mask == 1].mean()) < 9870


@pytest.mark.gpu
def test_hip_on_mitten_crop(pkg, oracle):
    h, w, sf, mask, K, I, zs, zf = _inputs()
    dh = pkg.DataHandler(I=I, mask=mask, K=K, sf=sf, z0=zs.reshape(1, -1), I_h=h, I_w=w, I_c=3, I_n=8, I_n_total=8, zs_lr=zs, z_full=zf)
    ctx = pkg.Context(device_id=0)
    srps = pkg.SRPS(dh, ctx=ctx)
    en = srps.execute()
    assert len(en) == int(G["n_outer"]), (en, G["energies"])
    np.testing.assert_allclose(en, G["energies"], rtol=1e-2)
    assert abs(en[-1] - G["energies"][-1]) / G["energies"][-1] < 2e-3
    assert rel_rmse(srps.z(), G["final_z"]) < 1e-4                        # relative: depth ~ 700, ulp(700) = 6e-5
    assert np.abs(srps.rho() - G["final_rho"]).max() < 5e-3
    # the stored-tensor operator (what the operator-level API runs) gives the same
    ctx.set_option("tensor_recompute", 0)
    srps2 = pkg.SRPS(dh, ctx=ctx)
    en2 = srps2.execute()
    assert len(en2) == len(en) and rel_rmse(srps2.z(), srps.z()) < 1e-4
    ctx.close()
