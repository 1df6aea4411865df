"""The 8-bit image store (srps_api.hip: image_store_prepare): when every sample of I is k / 255.f for a byte k -- what the
reference's image loader produces (Utilities.cpp:343) -- the albedo sweep reads the images as bytes.  The floats it forms, and
with them every result of a solve, must be the SAME BITS as with the float store; images that are not of that form must leave
the float store in use."""
import numpy as np
import pytest

f32 = np.float32


def _quantised_scene(pkg, h, w, sf, n_img, seed, mask_kind="ragged"):
    sc = pkg.synth.make_scene(h, w, sf, n_img, seed=seed, mask_kind=mask_kind)
    k = np.rint(np.clip(sc.I, 0, 1) * 255).astype(np.uint8)
    # every byte value occurs, the extremes included
    flat = k.reshape(-1)
    flat[:256] = np.arange(256, dtype=np.uint8)
    flat[-256:] = np.arange(256, dtype=np.uint8)[::-1]
    sc.I = (k.astype(f32) / f32(255)).reshape(sc.I.shape)                     # Utilities.cpp:343
    return sc


def test_byte_to_float_recipe_is_the_division_for_all_256_values():
    """fma(k, R_hi, fl(k R_lo)) with R_hi + R_lo = 1/255 to 48 bits, as device_utils.h: unit_from_byte forms it, in exact rational
    arithmetic: the value the fused multiply-add rounds has k / 255.f as its nearest float for every byte"""
    from fractions import Fraction
    k = np.arange(256, dtype=f32)
    hi = f32(1) / f32(255)
    lo = f32(1.0 / 255.0 - float(hi))
    small = (k * lo).astype(f32)                                             # the rounded second product
    want = k / f32(255)
    for i in range(256):
        exact = Fraction(int(i)) * Fraction(float(hi)) + Fraction(float(small[i]))      # what the fused multiply-add rounds
        c = want[i]
        lo_n, hi_n = np.nextafter(c, f32(-1)), np.nextafter(c, f32(2))
        d = abs(exact - Fraction(float(c)))
        assert d < abs(exact - Fraction(float(lo_n))) and d < abs(exact - Fraction(float(hi_n))), i      # c is THE nearest float
    assert ((k * hi).astype(f32) != want).sum() > 100                        # the second term is what makes it exact
    # MATLAB's im2double followed by single() gives the same floats (no double rounding at these values)
    assert np.array_equal((np.arange(256) / 255.0).astype(f32), k / f32(255))


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,sf,n_img,kind", [(96, 64, 2, 7, "ragged"), (256, 128, 4, 20, "full"), (60, 44, 1, 3, "ellipse")])
def test_byte_store_gives_the_same_bits_as_the_float_store(pkg, h, w, sf, n_img, kind):
    sc = _quantised_scene(pkg, h, w, sf, n_img, seed=401 + n_img, mask_kind=kind)
    out = []
    for store in (1, 0):
        ctx = pkg.Context(device_id=0)
        ctx.set_option("image_store", store)
        srps = pkg.SRPS(pkg.DataHandler.from_scene(sc), ctx=ctx)
        en = srps.execute(max_outer=3)
        P = ctx.get("z").size
        assert ctx.get_option("image_store_bytes_active") == (1 if store and P % 4 == 0 else 0)
        out.append((np.array(en, f32), srps.z().copy(), srps.rho().copy(), ctx.get("s").copy(), P))
        ctx.close()
    (e1, z1, r1, s1, P), (e0, z0, r0, s0, _) = out
    if P % 4 != 0:
        pytest.skip("mask with P % 4 != 0: the float store is used (asserted above)")
    assert np.array_equal(e1.view(np.uint32), e0.view(np.uint32))
    assert np.array_equal(z1.view(np.uint32), z0.view(np.uint32))
    assert np.array_equal(r1.view(np.uint32), r0.view(np.uint32))
    assert np.array_equal(s1.view(np.uint32), s0.view(np.uint32))


@pytest.mark.gpu
def test_byte_store_is_not_used_for_other_images_and_follows_changes(pkg, oracle):
    sc = _quantised_scene(pkg, 64, 64, 2, 4, seed=411, mask_kind="full")
    ctx = pkg.Context(device_id=0)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    one_pass = lambda: pkg.alternating_loop(ctx, None, max_outer=1)
    one_pass()
    assert ctx.get_option("image_store") == 1 and ctx.get_option("image_store_bytes_active") == 1
    # one sample off the byte lattice (and, separately, a negative zero): the float store is used again after the next sweep
    I = ctx.get("I")
    assert ctx.get_option("image_store_bytes_active") == 1                   # reading the state back changes nothing
    for bad in (np.nextafter(I.flat[7], f32(2)), f32(-0.0)):
        J = I.copy(); J.flat[7] = bad
        ctx.set("I", J)
        one_pass()
        assert ctx.get_option("image_store_bytes_active") == 0
    ctx.set("I", I)
    one_pass()
    assert ctx.get_option("image_store_bytes_active") == 1
    # srps_upload_image replaces one image: the store is looked at again (a float-valued image ends it, the byte image restores it)
    h_w = sc.h * sc.w
    img0 = sc.I[0].reshape(3, h_w)
    ctx.upload_image(0, (img0 * f32(0.999)).astype(f32))
    one_pass()
    assert ctx.get_option("image_store_bytes_active") == 0
    ctx.upload_image(0, img0)
    one_pass()
    assert ctx.get_option("image_store_bytes_active") == 1
    # a device pointer to the images lets the caller write them at any time: no byte copy from then on
    ctx.device_ptr("I")
    one_pass()
    assert ctx.get_option("image_store_bytes_active") == 0
    ctx.close()
    # float-valued images (the synthetic scenes of the other tests): never active, results against the oracle as ever
    sc2 = pkg.synth.make_scene(64, 64, 2, 4, seed=412, mask_kind="full")
    ctx = pkg.Context(device_id=0)
    srps = pkg.SRPS(pkg.DataHandler.from_scene(sc2), ctx=ctx)
    srps.execute(max_outer=1)
    assert ctx.get_option("image_store_bytes_active") == 0
    ctx.close()
    # and the quantised scene against the oracle (the oracle reads floats)
    ctx = pkg.Context(device_id=0)
    srps = pkg.SRPS(pkg.DataHandler.from_scene(sc), ctx=ctx)
    en = srps.execute(max_outer=2)
    ref = oracle.execute(oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init), depth="faithful", max_outer=2)
    assert ctx.get_option("image_store_bytes_active") == 1
    assert float(np.sqrt(np.mean((srps.z() - ref.z) ** 2))) < 1e-4
    np.testing.assert_allclose(en, ref.energies, rtol=2e-2)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,sf,n_img,kind", [(96, 64, 2, 7, "ragged"), (256, 128, 4, 20, "full"), (60, 44, 1, 3, "ellipse")])
def test_images_handed_over_as_bytes_equal_their_floats(pkg, h, w, sf, n_img, kind):
    """srps_problem.I_u8 (the bytes the reference's image-folder loader read, Utilities.cpp:343) against the same images handed
    over as the floats byte / 255.f: the floats formed on the device, the byte store, and every result of a solve are the same
    bits; a quarter of the bytes crossed PCIe.  Also srps_upload_image_u8 for one image at a time."""
    sc = _quantised_scene(pkg, h, w, sf, n_img, seed=431 + n_img, mask_kind=kind)
    k = np.rint(sc.I * f32(255)).astype(np.uint8)
    assert np.array_equal(k.astype(f32) / f32(255), sc.I)
    out = []
    for how in ("floats", "bytes", "bytes_one_by_one"):
        ctx = pkg.Context(device_id=0)
        dh = pkg.DataHandler.from_scene(sc)
        if how == "bytes":
            dh.I_u8 = k; dh.I = None
        elif how == "bytes_one_by_one":
            dh.I = None                                    # set-up without images, then srps_upload_image_u8 per image
        ctx.setup(dh)
        if how == "bytes_one_by_one":
            for i in range(n_img):
                ctx.upload_image(i, k[i])
        P = ctx.dims()["npix"]
        I_dev = ctx.get("I")
        np.testing.assert_array_equal(I_dev.view(np.uint32), np.ascontiguousarray(sc.I[:, :, sc.mask == 1]).reshape(-1).view(np.uint32))
        en = pkg.alternating_loop(ctx, None, max_outer=3)
        if how == "bytes":
            assert ctx.get_option("image_store_bytes_active") == (1 if P % 4 == 0 else 0)
        out.append((np.array(en, f32), ctx.get("z"), ctx.get("rho"), ctx.get("s")))
        ctx.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
def test_setup_refuses_floats_and_bytes_together(pkg):
    sc = _quantised_scene(pkg, 40, 32, 2, 2, seed=440)
    dh = pkg.DataHandler.from_scene(sc)
    import ctypes as C
    from importlib import import_module
    lib = pkg.load()
    api = import_module("srmeetsps-cuda_amd.api")
    k = np.rint(sc.I * f32(255)).astype(np.uint8)
    mask = np.ascontiguousarray(dh.mask, f32); K = np.ascontiguousarray(dh.K, f32); zs = np.ascontiguousarray(dh.zs_lr, f32); zf = np.ascontiguousarray(dh.z_full, f32)
    I = np.ascontiguousarray(dh.I, f32)
    pr = api.Problem(dh.I_h, dh.I_w, dh.I_c, 2, 2, 0, 2, api._fptr(mask), api._fptr(K), api._fptr(I), api._fptr(zs), api._fptr(zf),
                     k.ctypes.data_as(C.POINTER(C.c_ubyte)))
    ctx = pkg.Context(device_id=0)
    assert lib.srps_setup(ctx.h, C.byref(pr)) == 1 and b"not both" in lib.srps_last_error()
    ctx.close()
