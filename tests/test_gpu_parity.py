"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Tolerances (all data is unit scale: depth ~ 1, intensities in [0,1]):
  * element-wise kernels (normals, gradients, init)            : 1e-6 absolute
  * one lighting / albedo phase                                : 2e-4 on s, 1e-4 on rho
  * operator A_ x                                              : 2e-5 relative l2
  * one depth step (101 truncated CG steps, fp32)              : depth RMSE < 1e-4  (north_star)
  * whole alternating loop (<= 11 outer passes)                : depth RMSE < 1e-4, energy 1e-3 rel
The reference's own results are only defined up to fp32 summation order (cuBLAS / cuSPARSE),
which is what these tolerances express; see DESIGN.md section 6.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def _t(a, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.pin_memory().cuda().contiguous()      # through pinned memory: the device does not map pageable pages (csrc/srps_xfer.hip)


def _scene(pkg, h=48, w=40, sf=2, n=5, kind="ragged", seed=3):
    return pkg.synth.make_scene(h, w, sf, n, seed=seed, mask_kind=kind)


def _state(oracle, sc):
    prob = oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init)
    return oracle.setup(prob), prob


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(np.linalg.norm(np.asarray(b, np.float64)), 1e-30))


def _shading_rel(s_a, s_b, rho, N):
    """max over (image, channel) of || A_c (s_a - s_b) || / || A_c s_b ||, A_c = rho_c (.) N"""
    worst = 0.0
    for c in range(rho.shape[0]):
        A = (rho[c][None, :] * N).astype(np.float64)            # [4][P]
        pa = s_a[:, c, :].astype(np.float64) @ A; pb = s_b[:, c, :].astype(np.float64) @ A
        worst = max(worst, float(np.max(np.linalg.norm(pa - pb, axis=1) / np.linalg.norm(pb, axis=1))))
    return worst


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def assert_same_stop(en, en_ref):
    """the stop rule (relative change < 5e-3, SRPS.cu:299) may fire one pass apart when the decisive
    relative change sits within 10 % of the threshold; otherwise the pass counts must be equal"""
    if len(en) == len(en_ref):
        return
    assert abs(len(en) - len(en_ref)) == 1, (en, en_ref)
    longer = en if len(en) > len(en_ref) else en_ref
    k = min(len(en), len(en_ref)) - 1
    rel_k = abs(longer[k - 1] - longer[k]) / abs(longer[k])
    assert 0.9 * 5e-3 < rel_k < 1.1 * 5e-3, (rel_k, en, en_ref)


# ------------------------------------------------------------------------------------------
def test_init_kernels(gpu_ctx, oracle):
    import torch
    rng = np.random.default_rng(0)
    h, w, nc = 12, 20, 3
    z0 = rng.uniform(0.5, 2.0, size=(nc, h * w)).astype(f32)
    z0[0, 5] = 0; z0[2, 17] = 0; z0[:, 40] = 0
    mean = torch.empty(h * w, device="cuda"); flag = torch.empty(h * w, dtype=torch.uint8, device="cuda")
    gpu_ctx.mean_across_channels(z0, h, w, nc, mean, flag)
    gpu_ctx.synchronize()
    m_ref, f_ref = oracle.mean_across_channels(z0, h, w, nc)
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref, rtol=1e-6)
    np.testing.assert_array_equal(flag.cpu().numpy(), f_ref)
    rho = torch.zeros(3 * 100, device="cuda")
    gpu_ctx.rho_init(rho, 100, 3)
    gpu_ctx.synchronize()
    assert torch.all(rho == 0.5)
    # landscape grid (w > h): the reference kernel would leave columns unwritten; we do not
    xx = torch.full((h * w,), -1e9, device="cuda"); yy = torch.full((h * w,), -1e9, device="cuda")
    gpu_ctx.meshgrid_create(w, h, 9.5, 5.5, xx, yy)
    gpu_ctx.synchronize()
    jj, ii = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
    np.testing.assert_allclose(xx.cpu().numpy(), (jj.reshape(-1) - 9.5).astype(f32))
    np.testing.assert_allclose(yy.cpu().numpy(), (ii.reshape(-1) - 5.5).astype(f32))


def test_normals(gpu_ctx, oracle):
    import torch
    rng = np.random.default_rng(1)
    P = 1003
    z = rng.uniform(0.8, 1.2, P).astype(f32); zx = rng.normal(0, 1e-3, P).astype(f32); zy = rng.normal(0, 1e-3, P).astype(f32)
    xx = rng.uniform(-500, 500, P).astype(f32); yy = rng.uniform(-500, 500, P).astype(f32)
    N = torch.empty(4 * P, device="cuda"); dz = torch.empty(P, device="cuda")
    gpu_ctx.normal_init(_t(z), _t(zx), _t(zy), _t(xx), _t(yy), P, 1200.0, 1200.0, N, dz)
    gpu_ctx.synchronize()
    N_ref, dz_ref = oracle.normal_init(z, zx, zy, xx, yy, 1200.0, 1200.0)
    np.testing.assert_allclose(N.cpu().numpy().reshape(4, P), N_ref, atol=2e-6)
    np.testing.assert_allclose(dz.cpu().numpy(), dz_ref, rtol=2e-6)


@pytest.mark.parametrize("kind,sf", [("ragged", 2), ("full", 4), ("ellipse", 2), ("ragged", 1), ("ragged", 3)])
def test_gradient_matches_make_gradient(gpu_ctx, oracle, pkg, kind, sf):
    import torch
    h, w = (36, 30) if sf == 3 else (40, 32)
    m = pkg.synth.make_mask(h, w, sf, kind)
    mask = pkg.synth.to_cm(m).astype(f32)
    geo = oracle.build_geometry(h, w, sf, mask)
    gpu_ctx.bind_grid(h, w, sf, mask)
    rng = np.random.default_rng(2)
    z = rng.normal(size=geo.npix).astype(f32)
    zx = torch.empty(geo.npix, device="cuda"); zy = torch.empty(geo.npix, device="cuda")
    gpu_ctx.gradient(_t(z), geo.npix, zx, zy)
    gpu_ctx.synchronize()
    np.testing.assert_allclose(zx.cpu().numpy(), geo.Dx @ z, atol=1e-6)
    np.testing.assert_allclose(zy.cpu().numpy(), geo.Dy @ z, atol=1e-6)


def test_lighting_phase(gpu_ctx, oracle, pkg):
    sc = _scene(pkg)
    st, _ = _state(oracle, sc)
    s_dev = _t(st.s)
    gpu_ctx.lightning_estimation(s_dev, _t(st.rho), _t(st.N), _t(st.I), st.geo.npix, sc.n_img, sc.n_ch)
    gpu_ctx.synchronize()
    s_ref = oracle.lighting_estimation(st.s.copy(), st.rho, st.N, st.I)
    got = s_dev.cpu().numpy()
    # The 4x4 Gram of [rho N0, rho N1, rho N2, rho] is close to singular while the normals are still
    # nearly constant (N2 ~ -1): s is only determined up to that flat direction in fp32, so the
    # comparison is made on what s predicts (the shading A s) and, loosely, on s itself.
    assert _shading_rel(got, s_ref, st.rho, st.N) < 1e-4, np.abs(got - s_ref).max()
    assert np.abs(got - s_ref).max() < 5e-2


@pytest.mark.parametrize("mode", [0, 1])
def test_albedo_phase(gpu_ctx, oracle, pkg, mode):
    sc = _scene(pkg)
    st, _ = _state(oracle, sc)
    oracle.lighting_estimation(st.s, st.rho, st.N, st.I)
    rho_dev = _t(st.rho)
    gpu_ctx.set_option("albedo_mode", mode)
    try:
        gpu_ctx.albedo_estimation(_t(st.s), rho_dev, _t(st.N), _t(st.I), st.geo.npix, sc.n_img, sc.n_ch)
        gpu_ctx.synchronize()
    finally:
        gpu_ctx.set_option("albedo_mode", 0)
    its = []
    rho_ref = oracle.albedo_estimation(st.s, st.rho.copy(), st.N, st.I, cg_iters=its)
    got = rho_dev.cpu().numpy()
    assert np.abs(got - rho_ref).max() < 1e-4
    if mode == 0:      # the CG variant also reproduces the reference's iteration counts (+-2)
        gi = gpu_ctx.last_cg_iterations()["albedo"][: sc.n_ch]
        assert all(abs(a - b) <= 2 for a, b in zip(gi, its)), (gi, its)


@pytest.mark.parametrize("kind,sf,h,w", [("ragged", 2, 48, 40), ("full", 4, 32, 48), ("ellipse", 4, 64, 48),
                                          ("ragged", 1, 24, 20), ("ragged", 3, 36, 30)])
def test_depth_operator_matches_assembled_matrix(gpu_ctx, oracle, pkg, kind, sf, h, w):
    """matrix-free A_ x  ==  (KT'KT + A'A) x with the reference's assembled matrix (dc.cu:668-736)"""
    import torch
    sc = pkg.synth.make_scene(h, w, sf, 3, seed=5, mask_kind=kind)
    st, _ = _state(oracle, sc)
    oracle.lighting_estimation(st.s, st.rho, st.N, st.I)
    oracle.albedo_estimation(st.s, st.rho, st.N, st.I)
    P = st.geo.npix
    gpu_ctx.bind_grid(h, w, sf, sc.mask)
    z_dev = _t(st.z)
    e = gpu_ctx.depth_estimation(_t(st.s), _t(st.rho), _t(st.N), _t(st.I), _t(st.xx), _t(st.yy), _t(st.dz), _t(st.z0s),
                                 z_dev, st.fx, st.fy, P, sc.n_img, sc.n_ch)
    assert np.isfinite(e)
    A, A_, B = oracle.assemble_depth_system(st.geo, st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    rng = np.random.default_rng(7)
    for _ in range(3):
        x = rng.normal(size=P).astype(f32)
        y = torch.empty(P, device="cuda")
        gpu_ctx.depth_operator_apply(_t(x), P, y)
        gpu_ctx.synchronize()
        ref = A_.astype(np.float64) @ x.astype(np.float64)
        assert rel(y.cpu().numpy(), ref) < 2e-5


@pytest.mark.parametrize("kind,sf,h,w,n", [("ragged", 2, 48, 40, 5), ("full", 4, 32, 48, 4), ("ellipse", 2, 64, 64, 6)])
def test_depth_phase(gpu_ctx, oracle, pkg, kind, sf, h, w, n):
    sc = pkg.synth.make_scene(h, w, sf, n, seed=11, mask_kind=kind)
    st, _ = _state(oracle, sc)
    oracle.lighting_estimation(st.s, st.rho, st.N, st.I)
    oracle.albedo_estimation(st.s, st.rho, st.N, st.I)
    P = st.geo.npix
    gpu_ctx.bind_grid(h, w, sf, sc.mask)
    z_dev = _t(st.z)
    e = gpu_ctx.depth_estimation(_t(st.s), _t(st.rho), _t(st.N), _t(st.I), _t(st.xx), _t(st.yy), _t(st.dz), _t(st.z0s),
                                 z_dev, st.fx, st.fy, P, sc.n_img, sc.n_ch)
    assert gpu_ctx.last_cg_iterations()["depth"] == 101           # truncated CG, dc.cu:252
    z_ref = st.z.copy()
    e_ref = oracle.depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z_ref, st.fx, st.fy)
    z64 = st.z.copy()
    e64 = oracle.mf_depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z64, st.fx, st.fy)
    got = z_dev.cpu().numpy()
    # calibration: the two oracle orderings (fp32 assembled vs fp64 matrix-free) bound the fp32 noise
    noise = rmse(z_ref, z64)
    assert rmse(got, z64) < max(1e-4, 3 * noise)
    assert rmse(got, z_ref) < 1e-4
    # energy: against the fp32 faithful restatement; the fp64 run only bounds the truncated-CG drift
    assert abs(e - e_ref) / abs(e_ref) < 1e-3 and abs(e - e64) / abs(e64) < 2e-2, (e, e_ref, e64)


@pytest.mark.parametrize("kind,sf,h,w", [("ragged", 2, 48, 40), ("full", 4, 64, 48), ("ellipse", 1, 36, 44)])
def test_depth_estimation_with_the_references_csr_arguments(gpu_ctx, oracle, pkg, kind, sf, h, w):
    """srps_depth_estimation_csr = cuda_based_depth_estimation's 35-argument list (devicecalls.cuh:36, call site SRPS.cu:293):
    Dx, Dy, KT built the reference's way (make_gradient / KT filter -> COO -> cuda_based_host_COO_to_device_CSR) are checked
    against the bound mask and the call gives the same depth and energy as srps_depth_estimation; a matrix that does not
    belong to the mask is refused with a message naming it"""
    import torch
    sc = pkg.synth.make_scene(h, w, sf, 4, seed=13, mask_kind=kind)
    st, _ = _state(oracle, sc)
    oracle.lighting_estimation(st.s, st.rho, st.N, st.I)
    oracle.albedo_estimation(st.s, st.rho, st.N, st.I)
    P, Ps = st.geo.npix, st.geo.npixs
    gpu_ctx.bind_grid(h, w, sf, sc.mask)

    def upload(M, shuffle_seed):
        M = M.tocoo()
        perm = np.random.default_rng(shuffle_seed).permutation(M.nnz)      # COO in arbitrary order, as push_back leaves it
        rp = torch.empty(M.shape[0] + 1, dtype=torch.int32, device="cuda"); ci = torch.empty(max(M.nnz, 1), dtype=torch.int32, device="cuda")
        vv = torch.empty(max(M.nnz, 1), device="cuda")
        gpu_ctx.host_COO_to_device_CSR(M.row[perm], M.col[perm], M.data[perm].astype(f32), M.shape[0], M.shape[1], rp, ci, vv)
        return (rp, ci, vv, M.shape[0], M.shape[1], M.nnz)
    Dx, Dy, KT = upload(st.geo.Dx, 1), upload(st.geo.Dy, 2), upload(st.geo.KT, 3)
    args = (_t(st.s), _t(st.rho), _t(st.N), _t(st.I), _t(st.xx), _t(st.yy), _t(st.dz))
    z_a = _t(st.z); z_b = _t(st.z)
    e_a = gpu_ctx.depth_estimation(*args, _t(st.z0s), z_a, st.fx, st.fy, P, sc.n_img, sc.n_ch)
    e_b = gpu_ctx.depth_estimation_csr(*args, Dx, Dy, KT, _t(st.z0s), z_b, st.fx, st.fy, P, sc.n_img, sc.n_ch)
    assert e_a == e_b
    np.testing.assert_array_equal(z_a.cpu().numpy(), z_b.cpu().numpy())
    # wrong matrices: Dy in place of Dx, a flipped sign, a KT weight that is not 1/sf^2, a wrong size
    for bad, name in (((Dy, Dy, KT), "Dx"), ((Dx, Dx, KT), "Dy")):
        with pytest.raises(pkg.SRPSError) as ei:
            gpu_ctx.depth_estimation_csr(*args, *bad, _t(st.z0s), _t(st.z), st.fx, st.fy, P, sc.n_img, sc.n_ch)
        assert name in str(ei.value)
    v_bad = Dx[2].clone(); v_bad[0] = -v_bad[0]
    with pytest.raises(pkg.SRPSError):
        gpu_ctx.depth_estimation_csr(*args, (Dx[0], Dx[1], v_bad) + Dx[3:], Dy, KT, _t(st.z0s), _t(st.z), st.fx, st.fy, P, sc.n_img, sc.n_ch)
    if Ps > 0:
        k_bad = KT[2].clone(); k_bad[KT[5] // 2] *= 2.0
        with pytest.raises(pkg.SRPSError) as ei:
            gpu_ctx.depth_estimation_csr(*args, Dx, Dy, (KT[0], KT[1], k_bad) + KT[3:], _t(st.z0s), _t(st.z), st.fx, st.fy, P, sc.n_img, sc.n_ch)
        assert "KT" in str(ei.value)
    with pytest.raises(pkg.SRPSError):
        gpu_ctx.depth_estimation_csr(*args, Dx[:3] + (P - 1, P, Dx[5]), Dy, KT, _t(st.z0s), _t(st.z), st.fx, st.fy, P, sc.n_img, sc.n_ch)


@pytest.mark.parametrize("kind,sf,h,w,n", [("ragged", 2, 48, 40, 5), ("full", 4, 64, 48, 6)])
def test_full_alternating_loop(gpu_ctx, oracle, pkg, kind, sf, h, w, n):
    """SRPS::execute end to end: same number of outer passes, energies and final z/rho/s"""
    sc = pkg.synth.make_scene(h, w, sf, n, seed=21, mask_kind=kind)
    prob = oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init)
    ref = oracle.execute(prob, depth="faithful")
    dh = pkg.DataHandler.from_scene(sc)
    srps = pkg.SRPS(dh, ctx=gpu_ctx)
    energies = srps.execute()
    assert_same_stop(energies, ref.energies)
    n_common = min(len(energies), len(ref.energies))
    # the first passes start from constant normals, where the lighting Gram is near-singular and the
    # truncated CG is far from converged: their energies are reproducible to ~1e-3..1e-2 only
    np.testing.assert_allclose(energies[:n_common], ref.energies[:n_common], rtol=1e-2)
    assert abs(energies[n_common - 1] - ref.energies[n_common - 1]) / ref.energies[n_common - 1] < 1e-3
    if len(energies) != len(ref.energies):       # compare at the same pass count
        ref = oracle.execute(prob, depth="faithful", max_outer=len(energies))
    assert rmse(srps.z(), ref.z) < 1e-4
    assert np.abs(srps.rho() - ref.rho).max() < 2e-3
    assert _shading_rel(srps.s(), ref.s, ref.rho, ref.N) < 2e-3, np.abs(srps.s() - ref.s).max()
    # and the python-driven loop (phase-split entry points) gives the same as the in-library loop
    srps2 = pkg.SRPS(dh, ctx=gpu_ctx)
    e2 = srps2.execute(max_outer=len(energies))
    np.testing.assert_allclose(e2, energies, rtol=1e-6)
    np.testing.assert_array_equal(srps2.z(), srps.z())


@pytest.mark.parametrize("sf,h,w", [(1, 300, 40), (2, 520, 72), (4, 1040, 48), (4, 248, 16), (2, 250, 18)])
def test_marching_kernel_equals_simple_kernel(gpu_ctx, oracle, pkg, sf, h, w):
    """the register-marching operator (several segments / strips, halo lanes, DPP row exchange) against
    the one-thread-per-pixel form, on a ragged mask: operator, residual and a whole CG solve"""
    import torch
    sc = pkg.synth.make_scene(h, w, sf, 2, seed=9, mask_kind="ragged")
    st, _ = _state(oracle, sc)
    P = st.geo.npix
    gpu_ctx.bind_grid(h, w, sf, sc.mask)
    x = np.random.default_rng(5).normal(size=P).astype(f32)

    def run(generic):
        if generic:      # a rough, generic tensor (all six entries of M well away from zero): operator check
            st.s[:, :, :3] = np.random.default_rng(3).normal(size=(2, 3, 3)).astype(f32) * 0.5
            st.s[:, :, 3] = 0.2
            st.rho[:] = np.random.default_rng(4).uniform(0.3, 0.9, size=st.rho.shape).astype(f32)
        else:            # the physical state after one lighting + albedo phase: CG check
            oracle.lighting_estimation(st.s, st.rho, st.N, st.I)
            oracle.albedo_estimation(st.s, st.rho, st.N, st.I)
        res = {}
        try:
            for mode in (1, 2):
                gpu_ctx.set_option("apply_mode", mode)
                z_dev = _t(st.z)
                e = gpu_ctx.depth_estimation(_t(st.s), _t(st.rho), _t(st.N), _t(st.I), _t(st.xx), _t(st.yy), _t(st.dz),
                                             _t(st.z0s), z_dev, st.fx, st.fy, P, sc.n_img, sc.n_ch)
                y = torch.empty(P, device="cuda")
                gpu_ctx.depth_operator_apply(_t(x), P, y)
                gpu_ctx.synchronize()
                res[mode] = (e, z_dev.cpu().numpy(), y.cpu().numpy())
        finally:
            gpu_ctx.set_option("apply_mode", 0)
        return res

    res = run(generic=False)
    assert rel(res[2][2], res[1][2]) < 1e-6                         # operator
    assert rmse(res[2][1], res[1][1]) < 1e-4                        # 101 CG steps
    assert abs(res[2][0] - res[1][0]) / abs(res[1][0]) < 1e-3       # energy
    res = run(generic=True)
    assert rel(res[2][2], res[1][2]) < 1e-6
    assert abs(res[2][0] - res[1][0]) / abs(res[1][0]) < 1e-2       # rough system: fp32 CG paths drift apart
    # and against the assembled matrix of the oracle
    A, A_, B = oracle.assemble_depth_system(st.geo, st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    assert rel(res[2][2], A_.astype(np.float64) @ x.astype(np.float64)) < 2e-5


@pytest.mark.parametrize("kind,sf,h,w,n", [("ragged", 2, 300, 72, 4), ("full", 4, 128, 96, 5), ("ellipse", 1, 64, 80, 3)])
def test_tensor_recompute_equals_stored_tensor(pkg, oracle, kind, sf, h, w, n):
    """operator kernel that rebuilds M from (rho_c/dz)^2 and 6 lighting constants per channel
    (29 B/unknown) == the one that streams the stored 6-plane tensor (41 B/unknown) == the oracle"""
    import torch
    sc = pkg.synth.make_scene(h, w, sf, n, seed=51, mask_kind=kind)
    ctx = pkg.Context(device_id=0)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    ctx.lighting(); ctx.albedo()
    P = ctx.dims()["npix"]
    x = np.random.default_rng(6).normal(size=P).astype(f32)
    out = {}
    for rec in (0, 1):
        ctx.set_option("tensor_recompute", rec)
        ctx.depth_partial()
        y = torch.empty(P, device="cuda")
        ctx.depth_operator_apply(_t(x), P, y)
        ctx.synchronize()
        out[rec] = y.cpu().numpy()
    assert rel(out[1], out[0]) < 3e-6
    st, _ = _state(oracle, sc)
    st.s[:] = ctx.get("s").reshape(st.s.shape); st.rho[:] = ctx.get("rho").reshape(st.rho.shape)
    A, A_, B = oracle.assemble_depth_system(st.geo, st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    ref = A_.astype(np.float64) @ x.astype(np.float64)
    assert rel(out[1], ref) < 2e-5 and rel(out[0], ref) < 2e-5
    # whole depth step in both forms
    z = {}
    for rec in (0, 1):
        ctx.set_option("tensor_recompute", rec)
        ctx.set("z", st.z)
        e = ctx.depth()
        z[rec] = (ctx.get("z"), e)
    assert rmse(z[1][0], z[0][0]) < 1e-4 and abs(z[1][1] - z[0][1]) / z[0][1] < 1e-3
    ctx.close()


@pytest.mark.parametrize("kind,sf,h,w,n,n_ch", [("ragged", 2, 96, 72, 5, 3), ("full", 4, 64, 48, 23, 3), ("ellipse", 1, 50, 46, 2, 1)])
def test_fused_energy_and_lighting_sweep_equals_separate_passes(pkg, kind, sf, h, w, n, n_ch):
    """the sweep over I that evaluates the energy of pass k also leaves the lighting sums of pass k+1
    (normals recomputed in registers): same lighting, albedo, depth and energies as the two separate
    passes up to the summation order; stale sums are never used after the state was written"""
    sc = pkg.synth.make_scene(h, w, sf, n, seed=77, n_ch=n_ch, mask_kind=kind)
    dh = pkg.DataHandler.from_scene(sc)
    res = {}
    for fuse in (0, 1):
        ctx = pkg.Context(device_id=0)
        ctx.set_option("fuse_energy_lighting", fuse)
        ctx.setup(dh)
        en = pkg.alternating_loop(ctx, None, max_outer=3)
        res[fuse] = (en, ctx.get("s"), ctx.get("rho"), ctx.get("z"), ctx.get("N"))
        if fuse:
            # overwrite rho after an energy pass: the cached sums must be dropped, not reused
            ctx.energy_partial(); ctx.energy_finish()
            ctx.normals()
            ctx.set("rho", (res[1][2] * 0.5).astype(f32))
            ctx.lighting()
            s_after = ctx.get("s")
            ctx.set_option("fuse_energy_lighting", 0)
            ctx.lighting()                                  # warm start = the converged s: must not move much
            np.testing.assert_allclose(ctx.get("s"), s_after, rtol=1e-2, atol=1e-3)
            assert not np.allclose(s_after, res[1][1], rtol=1e-2)          # the lighting did react to the new rho
        ctx.close()
    # the fused sweep adds in a different order (2 pixels per thread instead of 4): equal up to rounding
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-3)
    for k in (1, 2, 3, 4):
        np.testing.assert_allclose(res[1][k], res[0][k], rtol=2e-3, atol=1e-3)
    assert rmse(res[1][3], res[0][3]) < 2e-5


@pytest.mark.parametrize("kind,sf,h,w,n,n_ch", [("ragged", 2, 96, 72, 5, 3), ("full", 4, 64, 48, 23, 3), ("ellipse", 1, 50, 46, 2, 1), ("ragged", 1, 57, 33, 4, 2)])
def test_depth_assembly_from_the_sums_of_the_albedo_sweep(pkg, kind, sf, h, w, n, n_ch):
    """the albedo sweep over I also leaves SA = sum_i fx s_i0 I_i, SA' = sum_i fy s_i1 I_i, SB = sum_i s_i2 I_i per channel;
    the depth assembly then forms q = sum_c g [(SA - xx SB, SA' - yy SB, -SB) - rho (CA - xx CB, CA' - yy CB, -CB)] without a second
    pass over I: same right-hand side planes as the assembly that streams I (rounding), same solve"""
    sc = pkg.synth.make_scene(h, w, sf, n, seed=31, n_ch=n_ch, mask_kind=kind)
    dh = pkg.DataHandler.from_scene(sc)
    out = {}
    for sums in (0, 1):
        ctx = pkg.Context(device_id=0)
        ctx.set_option("assemble_from_sums", sums)
        ctx.setup(dh)
        ctx.lighting(); ctx.albedo()
        ctx.depth_partial()
        q = ctx.exchange("depth").cpu().numpy().copy()
        ctx.depth_solve(); ctx.energy_partial(); e = ctx.energy_finish()
        out[sums] = (q, ctx.get("z"), e)
        # a second assembly after the lighting was overwritten must not use stale sums
        if sums:
            ctx.set("s", (ctx.get("s") * 1.5).astype(f32))
            ctx.depth_partial()
            q2 = ctx.exchange("depth").cpu().numpy()
            assert rel(q2, q) > 1e-2
        ctx.close()
    assert rel(out[1][0], out[0][0]) < 1e-5
    assert rmse(out[1][1], out[0][1]) < 1e-5 and abs(out[1][2] - out[0][2]) <= 1e-4 * abs(out[0][2])


def test_operator_level_depth_with_principal_point_hint(gpu_ctx, oracle, pkg):
    """srps_depth_estimation (the reference's signature: xx, yy as arrays) with and without the optional
    srps_set_principal_point hint that switches it to the tensor-recompute operator"""
    sc = pkg.synth.make_scene(64, 80, 2, 4, seed=61, mask_kind="ragged")
    st, _ = _state(oracle, sc)
    oracle.lighting_estimation(st.s, st.rho, st.N, st.I); oracle.albedo_estimation(st.s, st.rho, st.N, st.I)
    P = st.geo.npix
    res = []
    for hint in (False, True):
        gpu_ctx.bind_grid(sc.h, sc.w, sc.sf, sc.mask)              # also clears a previous hint
        if hint:
            gpu_ctx.set_principal_point(sc.K[6], sc.K[7])
        z_dev = _t(st.z)
        e = gpu_ctx.depth_estimation(_t(st.s), _t(st.rho), _t(st.N), _t(st.I), _t(st.xx), _t(st.yy), _t(st.dz), _t(st.z0s),
                                     z_dev, st.fx, st.fy, P, sc.n_img, sc.n_ch)
        res.append((e, z_dev.cpu().numpy()))
    z_ref = st.z.copy()
    e_ref = oracle.depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z_ref, st.fx, st.fy)
    for e, z in res:
        assert rmse(z, z_ref) < 1e-4 and abs(e - e_ref) / e_ref < 1e-3
    assert rmse(res[0][1], res[1][1]) < 1e-4


def test_image_sharding_equals_single_context(pkg, oracle):
    """SURVEY 8e on one GPU: two contexts hold disjoint image shards; the host sums their exchange
    buffers (what the RCCL all-reduce does across GPUs) -> same result as one context with all images"""
    import torch
    n = 5
    sc_all = pkg.synth.make_scene(40, 48, 2, n, seed=31, mask_kind="ragged")
    ref_ctx = pkg.Context(device_id=0)
    ref = pkg.SRPS(pkg.DataHandler.from_scene(sc_all), ctx=ref_ctx)
    e_ref = ref.execute(max_outer=3)
    ctxs = []
    for rank in range(2):
        lo, hi = pkg.shard_range(n, 2, rank)
        sc = pkg.synth.make_scene(40, 48, 2, n, seed=31, mask_kind="ragged", img_begin=lo, img_end=hi)
        c = pkg.Context(device_id=0)
        c.setup(pkg.DataHandler.from_scene(sc))
        ctxs.append(c)

    class Pair:
        """drives both shards in lockstep; 'all-reduce' = sum of the two exchange buffers"""
        def __getattr__(self, name):
            def call(*a):
                out = [getattr(c, name)(*a) for c in ctxs]
                return out[0]
            return call

        def exchange(self, which):
            return [c.exchange(which) for c in ctxs]

    def all_reduce(bufs):
        for c in ctxs: c.synchronize()
        tot = bufs[0] + bufs[1]
        bufs[0].copy_(tot); bufs[1].copy_(tot)
        torch.cuda.synchronize()

    e_sh = pkg.alternating_loop(Pair(), all_reduce, max_outer=3)
    np.testing.assert_allclose(e_sh, e_ref, rtol=2e-4)
    for c in ctxs:
        assert rmse(c.get("z"), ref.z()) < 2e-5
        assert np.abs(c.get("rho") - ref.ctx.get("rho")).max() < 2e-4
    np.testing.assert_array_equal(ctxs[0].get("z"), ctxs[1].get("z"))       # replicas stay bit-identical
    np.testing.assert_array_equal(ctxs[0].get("s"), ctxs[1].get("s"))
    for c in ctxs + [ref_ctx]: c.close()


def test_csr_operators(gpu_ctx, oracle, pkg):
    import torch
    import scipy.sparse as sp
    rng = np.random.default_rng(4)
    n_r, n_c = 300, 257
    A = sp.random(n_r, n_c, density=0.03, random_state=5, dtype=np.float32).tocoo()
    perm = rng.permutation(A.nnz)
    row, col, val = A.row[perm], A.col[perm], A.data[perm]
    rp = torch.empty(n_r + 1, dtype=torch.int32, device="cuda"); ci = torch.empty(A.nnz, dtype=torch.int32, device="cuda")
    vv = torch.empty(A.nnz, device="cuda")
    gpu_ctx.host_COO_to_device_CSR(row, col, val, n_r, n_c, rp, ci, vv)
    csr = sp.csr_matrix((val, (row, col)), shape=(n_r, n_c))
    np.testing.assert_array_equal(rp.cpu().numpy(), csr.indptr)
    x = rng.normal(size=n_c).astype(f32); xt = rng.normal(size=n_r).astype(f32)
    y = torch.empty(n_r, device="cuda"); yt = torch.empty(n_c, device="cuda")
    gpu_ctx.sparsemat_densevec_mul(rp, ci, vv, n_r, n_c, A.nnz, _t(x), y)
    gpu_ctx.sparsemat_densevec_mul(rp, ci, vv, n_r, n_c, A.nnz, _t(xt), yt, transpose=True)
    gpu_ctx.synchronize()
    np.testing.assert_allclose(y.cpu().numpy(), csr @ x, atol=1e-5)
    np.testing.assert_allclose(yt.cpu().numpy(), csr.T @ xt, atol=1e-5)
    # CG on an SPD CSR matrix against the oracle's restatement of dc.cu:229-279
    n = 200
    Bm = sp.random(n, n, density=0.05, random_state=6, dtype=np.float64)
    S = (Bm.T @ Bm + sp.identity(n) * 0.5).tocoo()
    rp2 = torch.empty(n + 1, dtype=torch.int32, device="cuda"); ci2 = torch.empty(S.nnz, dtype=torch.int32, device="cuda")
    v2 = torch.empty(S.nnz, device="cuda")
    gpu_ctx.host_COO_to_device_CSR(S.row, S.col, S.data.astype(f32), n, n, rp2, ci2, v2)
    b = rng.normal(size=n).astype(f32); x0 = np.zeros(n, f32)
    xd = _t(x0); bd = _t(b)
    it = gpu_ctx.conjugate_gradient(rp2, ci2, v2, n, S.nnz, xd, bd)
    S32 = S.tocsr().astype(f32)
    xr = x0.copy(); br = b.copy()
    it_ref = oracle.conjugate_gradient(lambda v: (S32 @ v).astype(f32), xr, br)
    assert abs(it - it_ref) <= 2, (it, it_ref)
    np.testing.assert_allclose(xd.cpu().numpy(), xr, atol=2e-5)
    np.testing.assert_allclose(xd.cpu().numpy(), np.linalg.solve(S.toarray(), b.astype(np.float64)), atol=1e-4)


def test_error_behaviour(gpu_ctx, pkg):
    """status codes + message instead of the reference's exit(1)/throw"""
    with pytest.raises(pkg.SRPSError) as ei:
        gpu_ctx.bind_grid(10, 10, 3, np.ones(100, f32))         # 10 not a multiple of 3
    assert ei.value.code == 1
    bad = np.ones(64, f32); bad[3] = 0.5
    with pytest.raises(pkg.SRPSError):
        gpu_ctx.bind_grid(8, 8, 2, bad)                          # mask must be {0,1}
    with pytest.raises(pkg.SRPSError):
        gpu_ctx.bind_grid(8, 8, 2, np.zeros(64, f32))            # empty mask
    c2 = pkg.Context(device_id=0)
    with pytest.raises(pkg.SRPSError) as ei:
        c2.lighting()                                            # no setup yet
    assert ei.value.code == 3
    c2.close()
    with pytest.raises(pkg.SRPSError):
        pkg.Context(device_id=99)


def test_command_line_program_end_to_end(pkg, oracle, tmp_path):
    """`srps --dstype matlab --dsloc scene.mat` (C++ host: MAT5 loader, depth pre-processing,
    SRPS::execute, MAT5 dumps) == the Python host on the same file == the oracle"""
    import subprocess
    import scipy.io
    pkg.host.load()
    sc = pkg.synth.make_scene(40, 48, 2, 4, seed=41, mask_kind="ragged")
    h, w = sc.h, sc.w
    I4 = np.transpose(sc.I.reshape(sc.n_img, sc.n_ch, w, h), (3, 2, 1, 0)).astype(np.float64)      # h x w x c x n
    Kmat = sc.K.reshape(3, 3).T.astype(np.float64)
    z0 = sc.z0.reshape(w // sc.sf, h // sc.sf).T.astype(np.float64)
    z0[3, 4] = 0.0                                                                                   # one invalid sample -> inpainted
    path = str(tmp_path / "scene.mat")
    scipy.io.savemat(path, {"I": I4, "K": Kmat, "mask": sc.mask.reshape(w, h).T.astype(np.uint8), "sf": float(sc.sf), "z0": z0},
                     do_compression=True)
    out = subprocess.run([pkg.host.CLI, "--dstype=matlab", f"--dsloc={path}", "-o", str(tmp_path), "--images"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    from PIL import Image
    for name in ("Normals-Initial.png", "Normals-Current-Iteration.png", "Albedo.png", "Depth.png"):     # the imshow windows of SRPS.cu:321-327
        assert Image.open(str(tmp_path / name)).mode == "RGB"
    assert "Lightning Estimation" in out.stdout and "Iteration 01 summary" in out.stdout and "Done!" in out.stdout
    z_cli = scipy.io.loadmat(str(tmp_path / "z.mat"))["x"][:, 0]
    rho_cli = scipy.io.loadmat(str(tmp_path / "rho.mat"))["x"][:, 0]
    s_cli = scipy.io.loadmat(str(tmp_path / "s.mat"))["x"][:, 0]
    # Python host on the same file, same C++ loader and pre-processing
    dh = pkg.host.load_dataset("matlab", path)
    np.testing.assert_array_equal(scipy.io.loadmat(str(tmp_path / "zs_init.mat"))["x"][:, 0], dh.zs_lr)
    ctx = pkg.Context(device_id=0)
    srps = pkg.SRPS(dh, ctx=ctx)
    en = srps.execute()
    np.testing.assert_array_equal(z_cli, srps.z())                  # same library, same inputs: bit-identical
    np.testing.assert_array_equal(rho_cli, srps.rho().reshape(-1)); np.testing.assert_array_equal(s_cli, srps.s().reshape(-1))
    assert out.stdout.count("Iteration") == len(en)
    ctx.close()
    # oracle on the same pre-processed inputs
    ref = oracle.execute(oracle.Problem(h, w, sc.sf, dh.mask, dh.K, dh.I, dh.zs_lr, dh.z_full), depth="faithful")
    assert ref.iterations == len(en)
    assert rmse(z_cli, ref.z) < 1e-4
