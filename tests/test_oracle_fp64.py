"""The whole-loop restatement in one arithmetic type (oracle/srps_solve_mf.c: `oc64_*` fp64, `oc32_*` fp32) -- pinned on the CPU.

Why it exists: the GPU whole-solve tests compare the HIP library with the fp32 assembled-CSR oracle, and the two drift apart pass by
pass (energies 5e-5 -> 7.8e-4 over seven passes at 2048 x 2048: VERDICT round 5, weak #2).  Both are fp32 roundings of the same
truncated recurrences; the fp64 run of the same loop is what tells them apart (tests/test_gpu_drift_calibration.py).  Before it may
judge anything it is pinned here:

  1. against an INDEPENDENT numpy fp64 statement of the same reference lines, written in this file with dense per-pixel loops / numpy
     reductions (no code shared with srps_solve_mf.c or with srps_oracle.py's phases): lighting dc.cu:408-444, albedo dc.cu:513-548,
     depth dc.cu:636-786 as a DENSE matrix (rows of A by dc.cu:676-691, A_ = KT'KT + A'A), CG dc.cu:229-279, normals dc.cu:171-223,
     stop rule SRPS.cu:297-302 -- on a ragged mask (backward / empty gradient rows, partial LR blocks);
  2. its fp32 build against the numpy oracle's faithful fp32 loop (assembled matrices) at the tolerance the two fp32 orderings have;
  3. the calibration itself at a CPU size: both fp32 restatements measured from fp64.
"""
import os
import subprocess

import numpy as np
import pytest

f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def CO():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import c_oracle
    return c_oracle


def _start(sc, CO):
    st = CO.Structure(sc.h, sc.w, sc.sf, sc.mask)
    I = np.ascontiguousarray(sc.I[:, :, st.imask])
    K = np.asarray(sc.K, f32)
    xx = ((st.imask // sc.h).astype(f32) - K[6]).astype(f32)
    yy = ((st.imask % sc.h).astype(f32) - K[7]).astype(f32)
    first = st.imask[st.blk_pix.reshape(st.Ps, -1)[:, 0]]
    z0s = np.asarray(sc.zs_lr, f32)[((first // sc.h) // sc.sf) * (sc.h // sc.sf) + (first % sc.h) // sc.sf]
    return st, I, z0s, np.ascontiguousarray(sc.z_init[st.imask]), xx, yy, float(K[0]), float(K[4])


# ---- the independent fp64 statement ------------------------------------------------------------------------------------------
CG_CAP = 100                                                              # dc.cu:231; lowered by the pin test below


def _cg64(A, x, b):
    """dc.cu:229-279 on a dense matrix, fp64"""
    tol = float(f32(1e-9))
    r1 = float(b @ b); r0 = 0.0; k = 0
    p = np.zeros_like(b)
    while r1 > tol * tol and k <= CG_CAP:
        k += 1
        if k == 1:
            p = b.copy()
        else:
            p = (r1 / r0) * p + b
        om = A @ p
        alpha = r1 / float(p @ om)
        x += alpha * p
        b -= alpha * om
        r0 = r1
        r1 = float(b @ b)
    return k


def _dense_structure(sc):
    """Dx, Dy (SRPS.cu:29-47) and KT (SRPS.cu:170-193, Utilities.cpp:216) as dense fp64 matrices, by plain loops over the image"""
    h, w, sf = sc.h, sc.w, sc.sf
    m = sc.mask.reshape(w, h).T != 0
    idx = -np.ones((h, w), int)
    p = 0
    for j in range(w):
        for i in range(h):
            if m[i, j]:
                idx[i, j] = p; p += 1
    P = p
    Dx = np.zeros((P, P)); Dy = np.zeros((P, P))
    for j in range(w):
        for i in range(h):
            if not m[i, j]:
                continue
            c = idx[i, j]
            if i + 1 < h and m[i + 1, j]: Dy[c, idx[i + 1, j]] = 1; Dy[c, c] = -1
            elif i - 1 >= 0 and m[i - 1, j]: Dy[c, c] = 1; Dy[c, idx[i - 1, j]] = -1
            if j + 1 < w and m[i, j + 1]: Dx[c, idx[i, j + 1]] = 1; Dx[c, c] = -1
            elif j - 1 >= 0 and m[i, j - 1]: Dx[c, c] = 1; Dx[c, idx[i, j - 1]] = -1
    rows = []
    for bj in range(w // sf):
        for bi in range(h // sf):
            blk = idx[bi * sf:(bi + 1) * sf, bj * sf:(bj + 1) * sf]
            if (blk >= 0).all():
                r = np.zeros(P); r[blk.reshape(-1)] = 1.0 / (sf * sf)
                rows.append(r)
    return Dx, Dy, np.array(rows).reshape(len(rows), P)


def _numpy_fp64_solve(sc, st_inputs, max_outer=None):
    _, I32, z0s, z_init, xx, yy, fx, fy = st_inputs
    I = I32.astype(np.float64)
    n_img, n_ch, P = I.shape
    Dx, Dy, KT = _dense_structure(sc)
    assert Dx.shape[0] == P
    z0s = z0s.astype(np.float64); z = z_init.astype(np.float64); xx = xx.astype(np.float64); yy = yy.astype(np.float64)

    def normals(z):
        zx, zy = Dx @ z, Dy @ z
        n = np.stack([fx * zx, fy * zy, -z - xx * zx - yy * zy, np.ones(P)])
        dz = np.maximum(float(f32(1e-10)), np.sqrt(n[0] ** 2 + n[1] ** 2 + n[2] ** 2))
        n[:3] /= dz
        return n, dz

    s = np.zeros((n_img, n_ch, 4)); s[:, :, 2] = -1
    rho = np.full((n_ch, P), 0.5)
    N, dz = normals(z)
    energies, last, iteration = [], float("nan"), 1
    while True:
        for i in range(n_img):                                            # dc.cu:410-411: images outside, channels inside
            for j in range(n_ch):
                A = rho[j] * N                                            # [4][P]  dc.cu:381
                ATA = A @ A.T
                b = A @ I[i, j] - ATA @ s[i, j]
                x = s[i, j].copy()
                _cg64(ATA, x, b)
                s[i, j] = x
        for c in range(n_ch):
            A = s[:, c, :] @ N                                            # [n_img][P]  dc.cu:507 (transposed storage)
            den = (A * A).sum(0); num = (A * I[:, c, :]).sum(0)
            b = num - den * rho[c]
            x = rho[c].copy()
            _cg64(np.diag(den), x, b)
            rho[c] = x
        rows, rhs_b = [], []
        for c in range(n_ch):
            for i in range(n_img):
                g = rho[c] / dz
                a1 = g * (fx * s[i, c, 0] - xx * s[i, c, 2]); a2 = g * (fy * s[i, c, 1] - yy * s[i, c, 2]); a3 = g * s[i, c, 2]
                rows.append(a1[:, None] * Dx + a2[:, None] * Dy - np.diag(a3))     # dc.cu:676-691
                rhs_b.append(I[i, c] - rho[c] * s[i, c, 3])                         # dc.cu:554
        A = np.concatenate(rows); B = np.concatenate(rhs_b)
        A_ = KT.T @ KT + A.T @ A
        rhs = KT.T @ z0s + A.T @ B
        b = rhs - A_ @ z
        _cg64(A_, z, b)
        e = float(((KT @ z - z0s) ** 2).sum() + ((A @ z - B) ** 2).sum())
        energies.append(e)
        with np.errstate(invalid="ignore"):
            rel = abs(last - e) / abs(e)
        stop = (e > last) or (rel < 5e-3) or (iteration > 10)
        last = e; iteration += 1
        if stop or (max_outer is not None and len(energies) >= max_outer):
            break
        N, dz = normals(z)
    return dict(energies=energies, z=z, rho=rho, s=s)


def test_fp64_solve_matches_an_independent_numpy_fp64_statement(pkg, CO):
    """Statement against statement.  A TRUNCATED CG is a chaotic map of its rounding errors once Ritz values have converged: on this
    460-unknown system two fp64 runs that differ only in the order of their dot products are 1e-16 apart after 20 steps, 8e-6 after
    40 and 1e-4 after 101 (fp64 against 80-bit: the same) -- so the two statements are compared with the CG capped at 16 steps, where
    they must agree to rounding, and over the whole loop at the reference's cap only as closely as fp64 pins such a small system.
    (At the sizes the GPU tests use, 1e6 unknowns and more, 101 steps are far from any converged Ritz value and the fp64 run moves
    by 1e-8 ... 4e-7 in depth with the summation order: tests/test_gpu_drift_calibration.py measures that on every run.)"""
    global CG_CAP
    sc = pkg.synth.make_scene(24, 32, 2, 3, seed=11, mask_kind="ragged")
    inp = _start(sc, CO)
    try:
        CG_CAP = 15
        CO._L.oc64_set_max_iter(15)
        got = CO.solve_mf(*inp, precision="f64", max_outer=3)
        ref = _numpy_fp64_solve(sc, inp, max_outer=3)
    finally:
        CG_CAP = 100
        CO._L.oc64_set_max_iter(100)
    assert len(got["energies"]) == len(ref["energies"]) == 3
    np.testing.assert_allclose(got["energies"], ref["energies"], rtol=1e-10)
    assert np.abs(got["z"] - ref["z"]).max() < 1e-11
    assert np.abs(got["rho"] - ref["rho"]).max() < 1e-10
    assert np.abs(got["s"] - ref["s"]).max() < 1e-9
    assert all(p["depth"] == 16 for p in got["steps"])
    got = CO.solve_mf(*inp, precision="f64")
    ref = _numpy_fp64_solve(sc, inp)
    assert len(got["energies"]) == len(ref["energies"]) >= 2
    np.testing.assert_allclose(got["energies"], ref["energies"], rtol=5e-4)
    assert np.abs(got["z"] - ref["z"]).max() < 1e-3
    assert all(p["depth"] == 101 for p in got["steps"])


def test_the_fp64_run_pins_itself_at_a_larger_size(pkg, CO):
    """the reference run's own uncertainty: the same fp64 solve with another summation order of its dot products (blocks of 1000
    instead of 256) -- at 256 x 256 the depth moves by < 1e-6 per pass, two orders below what the fp32 runs differ by"""
    sc = pkg.synth.make_scene(256, 256, 4, 4, seed=3, mask_kind="full")
    inp = _start(sc, CO)
    a = CO.solve_mf(*inp, precision="f64", keep_z=True, max_outer=4)
    try:
        CO._L.oc64_set_dot_block(1000)
        b = CO.solve_mf(*inp, precision="f64", keep_z=True, max_outer=4)
    finally:
        CO._L.oc64_set_dot_block(256)
    d = [float(np.sqrt(np.mean((x - y) ** 2))) for x, y in zip(a["z_pass"], b["z_pass"])]
    print("fp64 against itself (dot blocks 256 / 1000), depth RMSE per pass:", ["%.1e" % v for v in d])
    assert max(d) < 2e-6


def test_fp32_build_against_the_faithful_numpy_oracle(pkg, oracle, CO):
    """the same C statements in fp32 against srps_oracle.py's faithful loop (assembled matrices, fp32): two fp32 orderings of one algorithm"""
    sc = pkg.synth.make_scene(48, 64, 2, 4, seed=5, mask_kind="ragged")
    inp = _start(sc, CO)
    got = CO.solve_mf(*inp, precision="f32")
    ref = oracle.execute(oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init), depth="faithful")
    n = min(len(got["energies"]), len(ref.energies))
    assert abs(len(got["energies"]) - len(ref.energies)) <= 1
    rel = [abs(a - b) / abs(b) for a, b in zip(got["energies"][:n], ref.energies[:n])]
    assert max(rel) < 1e-3, rel
    if len(got["energies"]) == len(ref.energies):
        assert np.sqrt(np.mean((got["z"].astype(np.float64) - ref.z) ** 2)) < 1e-4


def test_calibration_at_a_cpu_size(pkg, oracle, CO):
    """what tests/test_gpu_drift_calibration.py does at the bench's sizes, here at 96 x 128: per pass, the deviation of each fp32
    restatement FROM fp64"""
    sc = pkg.synth.make_scene(96, 128, 4, 6, seed=21, mask_kind="ellipse")
    inp = _start(sc, CO)
    r64 = CO.solve_mf(*inp, precision="f64", keep_z=True)
    n = len(r64["energies"])
    r32 = CO.solve_mf(*inp, precision="f32", keep_z=True, max_outer=n)
    st = oracle.setup(oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init))
    e_as, z_as = [], []
    for _ in range(n):
        e_as.append(oracle.outer_iteration(st, depth="faithful")); z_as.append(st.z.copy())
    dev = lambda es: [abs(a - b) / abs(b) for a, b in zip(es, r64["energies"])]
    rm = lambda zs: [float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2))) for a, b in zip(zs, r64["z_pass"])]
    d_mf, d_as = dev(r32["energies"][:n]), dev(e_as)
    print("passes", n, "\n energy dev from fp64: matrix-free fp32", ["%.1e" % v for v in d_mf], "\n                      assembled  fp32", ["%.1e" % v for v in d_as],
          "\n depth RMSE from fp64: matrix-free fp32", ["%.1e" % v for v in rm(r32["z_pass"])], "\n                      assembled  fp32", ["%.1e" % v for v in rm(z_as)])
    z_mf, z_asd = rm(r32["z_pass"]), rm(z_as)
    assert max(d_mf) < 1e-3 and max(d_as) < 1e-3
    # what the calibration finds at every size tried (96 x 128 here; 256^2, 512^2 on the CPU; 1024^2, 2048^2 in the GPU test): the
    # matrix-free fp32 solve stays ~1e-5 from fp64 in depth, pass after pass; the ASSEMBLED fp32 solve (the reference's formulation:
    # A_ = KT'KT + A'A rounded entry by entry, dc.cu:734-736) moves away from fp64 monotonically with the passes -- past north_star's
    # 1e-4 here.  The drift between the HIP library and the assembled oracle is the assembled side's.
    assert max(z_mf) < 1e-4
    assert z_mf[-1] < z_asd[-1] and z_asd[-1] > z_asd[0]
    assert max(z_asd) < 5e-4
