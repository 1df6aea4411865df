"""Known answers that pin the oracle without a reference run (the reference cannot be built here: VERDICT / DESIGN section 5).

1. A depth system small enough to do by hand: 4 x 4 HR grid, full mask, sf 2, two images, one channel, every coefficient a
   small integer, so that A_ = KT'KT + A'A and rhs = KT'z0s + A'b (devicecalls.cu:583-599, 676-691, 734-745) are exact in
   fp32.  The system is built here by a literal, loop-by-loop transcription of those lines -- no code shared with the
   oracle -- one diagonal entry is derived by hand in the test, and the numpy oracle (faithful SpGEMM route), the C oracle
   (row-wise assembly and matrix-free operator) and, on the GPU, the HIP library have to reproduce it.
2. The same system solved densely in fp64: the reference's CG (devicecalls.cu:229-279) on 16 unknowns has converged long
   before its 101st step, so oracle, C oracle and HIP library must all land on the dense solution.
3. The two oracle implementations (numpy, C -- written separately, the C one without scipy's sparse kernels) on the real
   data of the reference's Mitten set (tests/golden/mitten_crop.npz): one depth step from the same state.
"""
import os
import subprocess

import numpy as np
import pytest

f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = W = 4
SF = 2
FX = FY = 2.0


def _tiny_inputs():
    """s, rho, dz chosen so that a1, a2, a3 are small integers: xx = j, yy = i (principal point 0), g = rho / dz = 1"""
    s = np.array([[[1, 0, 1, 0]], [[0, 1, -1, 1]]], f32)                # [image][channel][4]
    P = H * W
    rho = np.ones((1, P), f32); dz = np.ones(P, f32)
    jj, ii = np.divmod(np.arange(P), H)                                   # column-major: p = i + j*H
    xx = jj.astype(f32); yy = ii.astype(f32)
    I = np.stack([((ii + 2 * jj) % 5).astype(f32) / 4, ((3 * ii + jj) % 7).astype(f32) / 8])[:, None, :]      # [2][1][P], multiples of 1/8
    z0s = np.array([1.0, 1.5, 2.0, 1.25], f32)                            # LR depth, column-major 2 x 2
    z = (1 + (ii + jj) / 8).astype(f32)
    return s, rho, dz, xx, yy, I, z0s, z


def _literal_system():
    """devicecalls.cu:583-599 (a1, a2, a3), 550-581 (b), 676-691 (rows of A), 734-745 (A_, rhs), with make_gradient
    (SRPS.cu:29-47) and the KT filter (SRPS.cu:176-190, Utilities.cpp:216) written out as plain loops in fp64"""
    s, rho, dz, xx, yy, I, z0s, _ = _tiny_inputs()
    P = H * W
    Dx = np.zeros((P, P)); Dy = np.zeros((P, P))
    for j in range(W):
        for i in range(H):
            p = i + j * H
            if i + 1 < H: Dy[p, p + 1] = 1; Dy[p, p] = -1                 # forward in y (SRPS.cu:31-34)
            else: Dy[p, p] = 1; Dy[p, p - 1] = -1                         # backward       (SRPS.cu:35-38)
            if j + 1 < W: Dx[p, p + H] = 1; Dx[p, p] = -1                 # forward in x  (SRPS.cu:39-42)
            else: Dx[p, p] = 1; Dx[p, p - H] = -1                         # backward       (SRPS.cu:43-46)
    KT = np.zeros((4, P))
    for bj in range(W // SF):
        for bi in range(H // SF):
            for dj in range(SF):
                for di in range(SF):
                    KT[bi + bj * (H // SF), (bi * SF + di) + (bj * SF + dj) * H] = 1.0 / (SF * SF)      # SRPS.cu:188
    rows, rhs_b = [], []
    for c in range(1):
        for im in range(2):
            for p in range(P):
                g = rho[c, p] / dz[p]
                a1 = g * (FX * s[im, c, 0] - xx[p] * s[im, c, 2])          # dc.cu:588 (launch 616)
                a2 = g * (FY * s[im, c, 1] - yy[p] * s[im, c, 2])          # dc.cu:588 (launch 617)
                a3 = g * s[im, c, 2]                                       # dc.cu:597
                e = np.zeros(P); e[p] = 1
                rows.append(a1 * Dx[p] + a2 * Dy[p] - a3 * e)              # dc.cu:676-691
                rhs_b.append(I[im, c, p] - rho[c, p] * s[im, c, 3])        # dc.cu:554, 573 (N3 == 1)
    A = np.array(rows); b = np.array(rhs_b)
    A_ = KT.T @ KT + 1.0 * (A.T @ A)                                       # dc.cu:734-736, lambda = 1
    rhs = KT.T @ z0s.astype(np.float64) + A.T @ b                          # dc.cu:743-745
    return Dx, Dy, KT, A_, rhs


def test_one_diagonal_entry_by_hand():
    """A_[p, p] for the pixel (i, j) = (1, 1), p = 5.  Rows of A that touch p and their coefficient at p:
         its own rows (forward in x and y: -a1 - a2 - a3): image 0 has a1 = 2 - j = 1, a2 = -i = -1, a3 = 1 -> -1;
                                                           image 1 has a1 = j = 1, a2 = 2 + i = 3, a3 = -1 -> -3;
         the left neighbour (1, 0), forward in x (+a1): image 0: a1 = 2 -> 2; image 1: a1 = 0 -> 0;
         the upper neighbour (0, 1), forward in y (+a2): image 0: a2 = 0 -> 0; image 1: a2 = 2 -> 2.
       A'A[p, p] = 1 + 9 + 4 + 0 + 0 + 4 = 18, KT'KT[p, p] = (1/4)^2 = 1/16: A_[p, p] = 18.0625."""
    _, _, _, A_, _ = _literal_system()
    assert A_[5, 5] == 18.0625
    assert np.array_equal(A_, A_.T) and np.all(np.linalg.eigvalsh(A_) > 0)


@pytest.fixture(scope="module")
def CO():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import c_oracle
    return c_oracle


def test_oracles_reproduce_the_hand_built_system(oracle, CO):
    s, rho, dz, xx, yy, I, z0s, _ = _tiny_inputs()
    Dx, Dy, KT, A_, rhs = _literal_system()
    mask = np.ones(H * W, f32)
    geo = oracle.build_geometry(H, W, SF, mask)
    assert np.array_equal(geo.Dx.toarray(), Dx) and np.array_equal(geo.Dy.toarray(), Dy) and np.array_equal(geo.KT.toarray(), KT)
    # numpy oracle, faithful route (SpGEMM normal equations in fp32): every entry is exactly representable
    A, A_or, B = oracle.assemble_depth_system(geo, s, rho, dz, xx, yy, FX, FY, I)
    assert np.array_equal(A_or.toarray().astype(np.float64), A_)
    rhs_or = (geo.KT.T @ z0s + 1.0 * (A.T @ B)).astype(np.float64)
    assert np.array_equal(rhs_or, rhs)
    # numpy oracle, matrix-free route
    M, q, _ = oracle.mf_tensor(s, rho, dz, xx, yy, FX, FY, I)
    for k in range(16):
        e = np.zeros(16); e[k] = 1
        assert np.array_equal(oracle.mf_apply(geo, M, e), A_[:, k])
    assert np.array_equal(oracle.mf_rhs(geo, q, z0s), rhs)
    # C oracle: row-wise assembly, matrix-free operator, right-hand side
    st = CO.Structure(H, W, SF, mask)
    Mc, qc = CO.tensor(st, s, rho, dz, xx, yy, FX, FY, I)
    rp, ci, v = CO.assemble(st, Mc)
    import scipy.sparse as sp
    assert np.array_equal(sp.csr_matrix((v, ci, rp), shape=(16, 16)).toarray().astype(np.float64), A_)
    for k in range(16):
        e = np.zeros(16, f32); e[k] = 1
        assert np.array_equal(CO.mf_apply(st, Mc, e).astype(np.float64), A_[:, k])
    assert np.array_equal(CO.rhs(st, qc, z0s).astype(np.float64), rhs)


def test_cg_lands_on_the_dense_solution(oracle, CO):
    """dc.cu:758-759: residual, then CG from the warm start; 16 unknowns converge (r.r <= 1e-18) well inside 101 steps"""
    s, rho, dz, xx, yy, I, z0s, z0 = _tiny_inputs()
    _, _, _, A_, rhs = _literal_system()
    x_dense = np.linalg.solve(A_, rhs)
    geo = oracle.build_geometry(H, W, SF, np.ones(H * W, f32))
    z = z0.copy()
    oracle.depth_estimation(geo, s, rho, I, xx, yy, dz, z0s, z, FX, FY)
    assert np.abs(z - x_dense).max() < 2e-5
    z64 = z0.copy()
    oracle.mf_depth_estimation(geo, s, rho, I, xx, yy, dz, z0s, z64, FX, FY)
    assert np.abs(z64 - x_dense).max() < 2e-5
    st = CO.Structure(H, W, SF, np.ones(H * W, f32))
    for assembled in (True, False):
        zc = z0.copy()
        _, it = CO.depth_estimation(st, s, rho, I, xx, yy, dz, z0s, zc, FX, FY, assembled)
        assert it < 101 and np.abs(zc - x_dense).max() < 2e-5


@pytest.mark.gpu
def test_hip_library_on_the_hand_built_system(pkg):
    """the operator-level HIP path (bind_grid + depth_estimation + depth_operator_apply) against the hand-built matrix and
    its dense solution: a pin of the product that does not pass through the oracle at all"""
    import torch
    s, rho, dz, xx, yy, I, z0s, z0 = _tiny_inputs()
    _, _, _, A_, rhs = _literal_system()
    x_dense = np.linalg.solve(A_, rhs)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).pin_memory().cuda().contiguous()
    for hint in (False, True):                                  # stored tensor / tensor rebuilt from (rho/dz)^2 in the kernel
        ctx = pkg.Context(device_id=0)
        ctx.bind_grid(H, W, SF, np.ones(H * W, f32))
        if hint:
            ctx.set_principal_point(0.0, 0.0)
        z = t(z0)
        N = np.zeros((4, 16), f32)
        ctx.depth_estimation(t(s), t(rho), t(N), t(I), t(xx), t(yy), t(dz), t(z0s), z, FX, FY, 16, 2, 1)
        assert np.abs(z.cpu().numpy() - x_dense).max() < 2e-5
        for k in (0, 5, 15):
            e = np.zeros(16, f32); e[k] = 1
            y = torch.empty(16, device="cuda")
            ctx.depth_operator_apply(t(e), 16, y)
            ctx.synchronize()
            np.testing.assert_allclose(y.cpu().numpy(), A_[:, k], rtol=0, atol=1e-5)
        ctx.close()


def test_numpy_and_c_oracle_agree_on_the_mitten_crop(oracle, CO):
    G = np.load(os.path.join(ROOT, "tests", "golden", "mitten_crop.npz"))
    I = G["I_u8"].astype(f32) / f32(255)
    h, w, sf = int(G["h"]), int(G["w"]), int(G["sf"])
    st = oracle.setup(oracle.Problem(h, w, sf, G["mask"].astype(f32), G["K"], I, G["zs_lr"], G["z_full"]))
    oracle.lighting_estimation(st.s, st.rho, st.N, st.I)
    oracle.albedo_estimation(st.s, st.rho, st.N, st.I)
    z_np = st.z.copy()
    e_np = oracle.depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z_np, st.fx, st.fy)
    cs = CO.Structure(h, w, sf, G["mask"].astype(f32))
    assert cs.P == st.geo.npix and np.array_equal(cs.imask, st.geo.imask)
    scale = np.sqrt(np.mean(z_np.astype(np.float64) ** 2))               # depth ~ 700: relative RMSE
    for assembled in (True, False):
        z_c = st.z.copy()
        e_c, it = CO.depth_estimation(cs, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z_c, st.fx, st.fy, assembled)
        assert it == 101
        assert np.sqrt(np.mean((z_c.astype(np.float64) - z_np) ** 2)) / scale < 1e-4
        assert abs(e_c - e_np) <= 1e-3 * abs(e_np)


def test_cpu_budget_and_the_baseline_entry_point():
    """oracle/cpu_budget.py (the CPUs the process may really use: affinity and cgroup quota) and the C oracle's timing entry point behind
    bench.py's cpu_baseline leg (oc_bench_cg_csr: own first-touched copies, a warm-up, `reps` timed solves on `threads` threads)"""
    import os
    import cpu_budget
    import c_oracle as CO
    n = cpu_budget.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert CO._EFF >= 1
    st = CO.Structure(32, 24, 2, np.ones(32 * 24, np.float32))
    rng = np.random.default_rng(1)
    M = np.abs(rng.normal(size=(6, st.P))).astype(np.float32); M[[1, 2, 4]] *= 0.1
    rp, ci, v = CO.assemble(st, M.reshape(-1))
    b = rng.normal(size=st.P).astype(np.float32)
    sec = CO.bench_cg_csr(rp, ci, v, b, iters=5, reps=3, threads=1)
    assert sec.shape == (3,) and (sec > 0).all()
    assert CO.num_threads() <= max(CO._EFF, 1)                      # the team is bounded again after the call
    # the timed recurrence is the oracle's CG: the same five steps through the ordinary entry point give the same iterate as a second call
    x1 = np.zeros(st.P, np.float32); x2 = np.zeros(st.P, np.float32)
    CO.cg_csr(rp, ci, v, x1, b.copy(), fixed_iters=5); CO.cg_csr(rp, ci, v, x2, b.copy(), fixed_iters=5)
    np.testing.assert_array_equal(x1, x2)
