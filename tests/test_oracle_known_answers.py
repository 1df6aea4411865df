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


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: lighting (devicecalls.cu:408-444) and albedo (devicecalls.cu:447-548) by LITERAL transcription -- flat arrays, the
# reference's own index expressions, plain loops, fp64; no code shared with the oracle.  Small integers (and image values that are
# multiples of 1/8), so that every Gram entry, right-hand side and expansion value is exact in fp32 as well; two channels, so that the
# channel strides of `d_s + i*4*nchannels + j*4` (:417), `d_s + c*4` with stride `4*nchannels` (:503, :520) and
# `d_I + c*npix + i*npix*nchannels` (:526) are exercised.
# ---------------------------------------------------------------------------------------------------------------------------------
NPIX, NIMG, NCH = 16, 2, 2


def _tiny_photometric_inputs():
    ii, jj = np.arange(NPIX) % 4, np.arange(NPIX) // 4
    N = np.stack([ii - 1, jj - 2, -1 - (ii + jj) % 2, np.ones(NPIX)]).astype(f32)            # d_N[h*npix + i]: not unit vectors -- the operator does not ask
    rho = np.stack([1 + (ii + 2 * jj) % 2, 1 + (ii * jj) % 3]).astype(f32)                    # d_rho[c*npix + i]
    I = np.stack([[((3 * ii + jj + 2 * c + i) % 8) / 8 for c in range(NCH)] for i in range(NIMG)]).astype(f32)     # d_I[i][c][p]
    s = np.array([[[1, 0, -1, 0], [0, 1, -1, 1]], [[-1, 1, 0, 1], [1, 1, -1, 0]]], f32)       # d_s[i][c][4]
    return N, rho, I, s


def _literal_cg(row_ptr, col, val, n, x, b):
    """cuda_based_conjugate_gradient, devicecalls.cu:229-279, on CSR arrays, statement by statement (fp64)"""
    tol, max_iter = float(f32(1e-9)), 100
    spmv = lambda v: np.array([sum(val[t] * v[col[t]] for t in range(row_ptr[r], row_ptr[r + 1])) for r in range(n)])
    k, r0 = 0, 0.0
    r1 = float(b @ b)                                                                          # :251
    p = np.zeros(n)
    while r1 > tol * tol and k <= max_iter:                                                    # :252
        k += 1
        if k == 1:
            p = b.copy()                                                                       # :258
        else:
            beta = r1 / r0                                                                     # :262
            p = beta * p                                                                       # :263
            p = p + b                                                                          # :264
        omega = spmv(p)                                                                        # :267
        alpha = r1 / float(p @ omega)                                                          # :268-269
        x += alpha * p                                                                         # :270
        b -= alpha * omega                                                                     # :272
        r0 = r1
        r1 = float(b @ b)                                                                      # :274
    return k


def _literal_lighting():
    """devicecalls.cu:376-383 (A_for_lightning_estimation) and :408-444.  Returns the updated s, and per (image, channel) the Gram
    matrix and the residual right-hand side the CG was started with."""
    N, rho, I, s = _tiny_photometric_inputs()
    d_N, d_rho, d_I, d_s = N.reshape(-1).astype(np.float64), rho.reshape(-1).astype(np.float64), I.reshape(-1).astype(np.float64), s.reshape(-1).astype(np.float64)
    npix, nimages, nchannels = NPIX, NIMG, NCH
    d_A = np.zeros(npix * 4 * nchannels)
    for c in range(nchannels):
        for h in range(4):
            for i in range(npix):
                d_A[c * npix * 4 + h * npix + i] = d_rho[c * npix + i] * d_N[h * npix + i]     # :381
    grams, rhss = {}, {}
    for i in range(nimages):                                                                   # :410
        for j in range(nchannels):                                                             # :411
            A_ij = j * npix * 4                                                                # :412  (column-major npix x 4, lda = npix)
            b_ij = i * npix * nchannels + j * npix                                             # :413
            x_ij = i * 4 * nchannels + j * 4                                                   # :414
            ATA = np.zeros(16)                                                                 # column-major 4 x 4, ld 4
            for r in range(4):
                for cc in range(4):                                                            # sgemm(T, N, 4, 4, npix) :422
                    ATA[r + 4 * cc] = sum(d_A[A_ij + r * npix + p] * d_A[A_ij + cc * npix + p] for p in range(npix))
            ATb = np.array([sum(d_A[A_ij + r * npix + p] * d_I[b_ij + p] for p in range(npix)) for r in range(4)])   # sgemv(T) :423
            for r in range(4):                                                                 # sgemv(N, alpha = -1, beta = 1) :424
                ATb[r] = -sum(ATA[r + 4 * cc] * d_s[x_ij + cc] for cc in range(4)) + ATb[r]
            row_idx = [m // 4 for m in range(16)]; col_idx = [m % 4 for m in range(16)]        # :430-431
            val = [ATA[row_idx[m] + 4 * col_idx[m]] for m in range(16)]                        # :432
            row_ptr = [0, 4, 8, 12, 16]                                                        # coo2csr of row_idx :436
            grams[(i, j)] = np.array(val).reshape(4, 4); rhss[(i, j)] = ATb.copy()
            x = d_s[x_ij:x_ij + 4].copy()
            _literal_cg(row_ptr, col_idx, val, 4, x, ATb)                                      # :437
            d_s[x_ij:x_ij + 4] = x
    return d_s.reshape(NIMG, NCH, 4), grams, rhss


def _literal_albedo():
    """devicecalls.cu:447-463 (fill_A_expansion, fill_AT_expansion), :497-511 (A_for_albedo), :395-406 (MA_Mb), :513-548.
    Returns the updated rho and per channel the diagonal of A'A and the residual right-hand side."""
    N, rho, I, s = _tiny_photometric_inputs()
    d_N, d_rho, d_I, d_s = N.reshape(-1).astype(np.float64), rho.reshape(-1).astype(np.float64), I.reshape(-1).astype(np.float64), s.reshape(-1).astype(np.float64)
    npix, nimages, nchannels = NPIX, NIMG, NCH
    diags, rhss = {}, {}
    for c in range(nchannels):                                                                 # :518
        s_base = c * 4                                                                         # d_s + c * 4 :520
        d_s_buff = np.zeros(4 * nimages)
        for i in range(nimages):                                                               # :502-504
            d_s_buff[i * 4:i * 4 + 4] = d_s[s_base + i * 4 * nchannels: s_base + i * 4 * nchannels + 4]
        d_A = np.zeros(npix * nimages)                                                         # sgemm(N, N, npix, nimages, 4): column-major npix x nimages :507
        for i in range(nimages):
            for p in range(npix):
                d_A[p + i * npix] = sum(d_N[p + k * npix] * d_s_buff[k + i * 4] for k in range(4))
        n_e = npix * nimages
        A_row = [t for t in range(n_e)]; A_col = [t % npix for t in range(n_e)]; A_val = [d_A[t] for t in range(n_e)]      # fill_A_expansion :447-454
        AT_col = [t // nimages + (t % nimages) * npix for t in range(n_e)]                     # fill_AT_expansion :459
        AT_row = [t // nimages for t in range(n_e)]                                            # :460
        AT_val = [d_A[AT_col[t]] for t in range(n_e)]                                          # :461
        d_b = np.zeros(n_e)
        for i in range(nimages):                                                               # :525-527
            d_b[npix * i: npix * i + npix] = d_I[c * npix + i * npix * nchannels: c * npix + i * npix * nchannels + npix]
        # cuda_based_MA_Mb :395-406 with M = AT (npix x n_e), A (n_e x npix): MA = M A (csrgemm), Mb = M b - MA x
        MA = np.zeros((npix, npix))
        for t in range(n_e):                                                                   # entry (AT_row, AT_col) of M times row AT_col of A
            k = AT_col[t]
            assert A_row[k] == k
            MA[AT_row[t], A_col[k]] += AT_val[t] * A_val[k]
        Mb = np.zeros(npix)
        for t in range(n_e):
            Mb[AT_row[t]] += AT_val[t] * d_b[AT_col[t]]                                        # csrmv :404
        x = d_rho[npix * c: npix * c + npix].copy()
        Mb = -(MA @ x) + Mb                                                                    # csrmv(alpha = -1, beta = 1) :405
        assert np.count_nonzero(MA - np.diag(np.diag(MA))) == 0                                # the normal equations are diagonal
        diags[c] = np.diag(MA).copy(); rhss[c] = Mb.copy()
        nz = [(r, cc) for r in range(npix) for cc in range(npix) if MA[r, cc] != 0]
        row_ptr = [0] * (npix + 1)
        for r, _ in nz:
            row_ptr[r + 1] += 1
        row_ptr = list(np.cumsum(row_ptr))
        _literal_cg(row_ptr, [cc for _, cc in nz], [MA[r, cc] for r, cc in nz], npix, x, Mb)   # :540
        d_rho[npix * c: npix * c + npix] = x
    return d_rho.reshape(NCH, NPIX), diags, rhss


def test_lighting_by_hand_one_gram_entry_and_one_right_hand_side():
    """image 1, channel 0: A = rho_0 (.) [N0..N3].  Gram entry (3, 3) = sum rho_0^2 = sum over the 16 pixels of (1 + (i + 2j) % 2)^2
    = sum (1 + i % 2)^2 = 8 * 1 + 8 * 4 = 40; entry (0, 3) = sum rho_0^2 (i - 1) = sum_j [ (1)(-1) + (4)(0) + (1)(1) + (4)(2) ] = 4 * 8 = 32."""
    _, grams, _ = _literal_lighting()
    assert grams[(1, 0)][3, 3] == 40.0 and grams[(1, 0)][0, 3] == 32.0 and grams[(0, 0)][3, 3] == 40.0
    for g in grams.values():
        assert np.array_equal(g, g.T) and np.all(np.linalg.eigvalsh(g) > 0)


def test_oracles_reproduce_the_literal_lighting(oracle, CO):
    N, rho, I, s = _tiny_photometric_inputs()
    s_lit, grams, rhss = _literal_lighting()
    # the 4 x 4 systems have converged (r.r <= 1e-18) long before the cap: the literal result IS the least-squares solution
    for (i, j), g in grams.items():
        A = (rho[j] * N).astype(np.float64)
        assert np.array_equal(g, A @ A.T)
        assert np.array_equal(rhss[(i, j)], A @ I[i, j].astype(np.float64) - g @ s[i, j].astype(np.float64))
        np.testing.assert_allclose(s_lit[i, j], np.linalg.solve(g, A @ I[i, j].astype(np.float64)), rtol=0, atol=1e-9)
    s_np = s.copy()
    oracle.lighting_estimation(s_np, rho.copy(), N, I)
    np.testing.assert_allclose(s_np, s_lit, rtol=0, atol=2e-5)
    for prec, tol in (("f64", 1e-9), ("f32", 2e-5)):
        R = CO._Real(prec)
        sv = s.astype(R.dtype).copy()
        rc = R.fn("lighting")(NPIX, NIMG, NCH, R.p(rho.astype(R.dtype).reshape(-1)), R.p(N.astype(R.dtype).reshape(-1)), CO._f(np.ascontiguousarray(I)),
                              R.p(sv.reshape(-1)), None)
        assert rc == 0
        np.testing.assert_allclose(sv, s_lit, rtol=0, atol=tol)


def test_albedo_by_hand_one_diagonal_entry():
    """channel 1, pixel p = 5 = (i, j) = (1, 1): N = (0, -1, -1, 1); s_01 = (0, 1, -1, 1) -> a = -1 + 1 + 1 = 1; s_11 = (1, 1, -1, 0) ->
    a = -1 + 1 = 0: (A'A)[5, 5] = 1; (A'b)[5] = 1 * I[0][1][5] = ((3 + 1 + 2) % 8) / 8 = 0.75; rho_1[5] = 1 + 1 % 3 = 2: residual -1.25."""
    _, diags, rhss = _literal_albedo()
    assert diags[1][5] == 1.0 and rhss[1][5] == -1.25


def test_oracles_reproduce_the_literal_albedo(oracle, CO):
    N, rho, I, s = _tiny_photometric_inputs()
    rho_lit, diags, rhss = _literal_albedo()
    for c in range(NCH):
        sh = s[:, c, :].astype(np.float64) @ N.astype(np.float64)                              # [image][p]
        assert np.array_equal(diags[c], (sh * sh).sum(0))
        assert np.array_equal(rhss[c], (sh * I[:, c, :]).sum(0) - diags[c] * rho[c])
        ok = diags[c] > 0
        np.testing.assert_allclose(rho_lit[c][ok], ((sh * I[:, c, :]).sum(0) / np.where(ok, diags[c], 1))[ok], rtol=0, atol=1e-9)
        assert np.array_equal(rho_lit[c][~ok], rho[c][~ok].astype(np.float64))                 # a zero row: the CG never moves that pixel
    assert any((diags[c] == 0).any() for c in range(NCH)), "the case is meant to contain a pixel no image lights"
    rho_np = rho.copy()
    oracle.albedo_estimation(s, rho_np, N, I)
    np.testing.assert_allclose(rho_np, rho_lit, rtol=0, atol=2e-5)
    num, den = oracle.albedo_numden(s, N, I)
    rho_nd = rho.copy()
    oracle.albedo_solve_numden(rho_nd, num, den)
    np.testing.assert_allclose(rho_nd, rho_lit, rtol=0, atol=2e-5)
    for prec, tol in (("f64", 1e-9), ("f32", 2e-5)):
        R = CO._Real(prec)
        rv = rho.astype(R.dtype).copy()
        rc = R.fn("albedo")(NPIX, NIMG, NCH, R.p(s.astype(R.dtype).reshape(-1)), R.p(N.astype(R.dtype).reshape(-1)), CO._f(np.ascontiguousarray(I)),
                            R.p(rv.reshape(-1)), None)
        assert rc == 0
        np.testing.assert_allclose(rv, rho_lit, rtol=0, atol=tol)


@pytest.mark.gpu
def test_hip_library_on_the_literal_lighting_and_albedo(pkg):
    """the operator-level HIP entries (srps_lightning_estimation, srps_albedo_estimation: the counterparts of devicecalls.cuh:34-35) on
    the same case against the literal transcriptions -- pins of the product that do not pass through the oracle; both albedo modes"""
    import torch
    N, rho, I, s = _tiny_photometric_inputs()
    s_lit, _, _ = _literal_lighting()
    rho_lit, diags, _ = _literal_albedo()
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).pin_memory().cuda().contiguous()
    ctx = pkg.Context(device_id=0)
    d_s = t(s)
    ctx.lightning_estimation(d_s, t(rho), t(N), t(I), NPIX, NIMG, NCH)
    ctx.synchronize()
    np.testing.assert_allclose(d_s.cpu().numpy(), s_lit, rtol=0, atol=2e-5)
    for mode in (0, 3):
        ctx.set_option("albedo_mode", mode)
        d_rho = t(rho)
        ctx.albedo_estimation(t(s), d_rho, t(N), t(I), NPIX, NIMG, NCH)
        ctx.synchronize()
        np.testing.assert_allclose(d_rho.cpu().numpy(), rho_lit, rtol=0, atol=2e-5)
    ctx.close()
