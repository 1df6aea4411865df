"""Operator-level known answers for the oracle (CPU).  The reference has no tests or golden vectors
(SURVEY 4), so these pins are derived by hand from the reference's source lines, plus the
agreement of the oracle's independent restatements (faithful assembled / matrix-free / C)."""
import numpy as np
import pytest
import scipy.sparse as sp

f32 = np.float32


def cm(a2d):
    return np.ascontiguousarray(np.asarray(a2d).T).reshape(-1)


def test_downsampling_matrix_column_formula(oracle):
    """Utilities.cpp:209-216 on a 4 x 6 grid, sf = 2, by hand: LR pixel i = ii + jj*2 averages the HR
    block rows 2ii..2ii+1, columns 2jj..2jj+1; HR linear index = row + col*4."""
    row, col, val, n_row, n_col = oracle.downsampling_coo(4, 6, 2)
    assert (n_row, n_col) == (6, 24)
    expect = {0: [0, 1, 4, 5], 1: [2, 3, 6, 7], 2: [8, 9, 12, 13], 3: [10, 11, 14, 15], 4: [16, 17, 20, 21], 5: [18, 19, 22, 23]}
    for i, cols in expect.items():
        assert sorted(col[row == i].tolist()) == cols
    assert np.all(val == f32(0.25))


def test_gradient_rows_by_hand(oracle):
    """make_gradient SRPS.cu:29-47 on a mask with an isolated pixel, a 1-wide strip and an edge.
        j: 0 1 2 3
    i=0    1 1 0 1      (0,3) has only a lower neighbour
    i=1    0 1 0 1
    i=2    0 0 0 0
    i=3    1 0 0 0      (3,0) is isolated
    """
    m = np.array([[1, 1, 0, 1], [0, 1, 0, 1], [0, 0, 0, 0], [1, 0, 0, 0]], dtype=f32)
    h, w = m.shape
    geo = oracle.build_geometry(h, w, 1, cm(m))
    # compact order = ascending column-major index: (0,0)=0, (3,0)=1, (0,1)=2, (1,1)=3, (0,3)=4, (1,3)=5
    assert geo.imask.tolist() == [0, 3, 4, 5, 12, 13]
    Dx = geo.Dx.toarray(); Dy = geo.Dy.toarray()
    z = np.arange(1, 7, dtype=f32) ** 2          # 1,4,9,16,25,36
    # Dx (along j): (0,0): right neighbour (0,1) masked -> forward z[2]-z[0]
    #               (3,0): none -> 0 ; (0,1): right (0,2) unmasked, left (0,0) masked -> backward z[2]-z[0]
    #               (1,1): right unmasked, left (1,0) unmasked -> 0 ; (0,3),(1,3): right out of range, left unmasked -> 0
    np.testing.assert_array_equal(Dx @ z, [9 - 1, 0, 9 - 1, 0, 0, 0])
    # Dy (along i): (0,0): (1,0) unmasked, no upper -> 0 ; (3,0): lower out of range, upper (2,0) unmasked -> 0
    #               (0,1): lower (1,1) masked -> forward z[3]-z[2] ; (1,1): lower (2,1) unmasked, upper (0,1) -> backward z[3]-z[2]
    #               (0,3): forward z[5]-z[4] ; (1,3): backward z[5]-z[4]
    np.testing.assert_array_equal(Dy @ z, [0, 0, 16 - 9, 16 - 9, 36 - 25, 36 - 25])
    assert np.count_nonzero(Dx) == 4 and np.count_nonzero(Dy) == 8


def test_gradient_vectorised_equals_literal_double_loop(oracle, pkg):
    """the vectorised make_gradient against a literal transcription of the loop at SRPS.cu:29-47"""
    m2 = pkg.synth.make_mask(20, 14, 1, "ragged")
    h, w = m2.shape
    mask = cm(m2)
    geo = oracle.build_geometry(h, w, 1, mask)
    idx = geo.index_in_masked
    rows_x, cols_x, vals_x, rows_y, cols_y, vals_y = [], [], [], [], [], []
    for j in range(w):
        for i in range(h):
            c = i + j * h
            if mask[c] == 0:
                continue
            if i + 1 < h and mask[c + 1] != 0:
                rows_y += [idx[c], idx[c]]; cols_y += [idx[c + 1], idx[c]]; vals_y += [1, -1]
            elif i - 1 >= 0 and mask[c - 1] != 0:
                rows_y += [idx[c], idx[c]]; cols_y += [idx[c - 1], idx[c]]; vals_y += [-1, 1]
            if j + 1 < w and mask[c + h] != 0:
                rows_x += [idx[c], idx[c]]; cols_x += [idx[c + h], idx[c]]; vals_x += [1, -1]
            elif j - 1 >= 0 and mask[c - h] != 0:
                rows_x += [idx[c], idx[c]]; cols_x += [idx[c - h], idx[c]]; vals_x += [-1, 1]
    P = geo.npix
    Dx = sp.csr_matrix((vals_x, (rows_x, cols_x)), shape=(P, P)); Dy = sp.csr_matrix((vals_y, (rows_y, cols_y)), shape=(P, P))
    assert abs(Dx - geo.Dx).max() == 0 and abs(Dy - geo.Dy).max() == 0


def test_KT_rows_are_fully_masked_blocks(oracle, pkg):
    m2 = pkg.synth.make_mask(24, 16, 4, "ragged")
    geo = oracle.build_geometry(24, 16, 4, cm(m2))
    KT = geo.KT.toarray()
    assert np.all((KT != 0).sum(1) == 16) and np.allclose(KT[KT != 0], 1 / 16)       # SRPS.cu:188
    blocks = m2.reshape(6, 4, 4, 4).transpose(0, 2, 1, 3).reshape(6, 4, 16).all(-1)
    assert geo.npixs == int(blocks.sum())


def test_mean_across_channels_divides_by_channel_count(oracle):
    z0 = np.array([[2.0, 0.0, 3.0], [4.0, 6.0, 0.0]], dtype=f32)       # [nc=2][3 pixels]
    mean, flag = oracle.mean_across_channels(z0, 3, 1, 2)
    np.testing.assert_array_equal(mean, [3.0, 3.0, 1.5])                # dc.cu:108: sum of non-zero / nc
    np.testing.assert_array_equal(flag, [0, 1, 1])


def test_cg_small_spd_system(oracle):
    rng = np.random.default_rng(0)
    B = rng.normal(size=(4, 4)); A = (B @ B.T + 0.5 * np.eye(4)).astype(f32)
    b = rng.normal(size=4).astype(f32)
    x = np.zeros(4, f32); r = b.copy()
    it = oracle.conjugate_gradient(lambda v: (A @ v).astype(f32), x, r)
    np.testing.assert_allclose(x, np.linalg.solve(A.astype(np.float64), b), rtol=2e-4, atol=2e-5)
    assert 4 <= it <= 101


def test_cg_runs_at_most_101_steps(oracle):
    """k <= max_iter with max_iter = 100 (dc.cu:231, 252) => 101 steps on a system it cannot finish"""
    n = 400
    A = sp.diags(np.linspace(1e-4, 1e4, n)).astype(f32).tocsr()
    b = np.ones(n, f32); x = np.zeros(n, f32)
    assert oracle.conjugate_gradient(lambda v: (A @ v).astype(f32), x, b) == 101


@pytest.fixture(scope="module")
def small_state(oracle, pkg):
    sc = pkg.synth.make_scene(24, 20, 2, 4, seed=13, mask_kind="ragged")
    st = oracle.setup(oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init))
    oracle.lighting_estimation(st.s, st.rho, st.N, st.I)
    oracle.albedo_estimation(st.s, st.rho, st.N, st.I)
    return sc, st


def test_assembled_equals_matrix_free(oracle, small_state):
    """(KT'KT + A'A) x with the assembled A of dc.cu:668-736  ==  the matrix-free form"""
    sc, st = small_state
    A, A_, B = oracle.assemble_depth_system(st.geo, st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    M, q, bb = oracle.mf_tensor(st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    rng = np.random.default_rng(1)
    for _ in range(3):
        x = rng.normal(size=st.geo.npix)
        ref = (st.geo.KT.T @ (st.geo.KT @ x)).astype(np.float64) + A.astype(np.float64).T @ (A.astype(np.float64) @ x)
        got = oracle.mf_apply(st.geo, M, x)
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-6
        assert np.linalg.norm(A_.astype(np.float64) @ x - ref) / np.linalg.norm(ref) < 1e-5     # fp32 SpGEMM
    rhs_ref = st.geo.KT.T.astype(np.float64) @ st.z0s + A.astype(np.float64).T @ B.astype(np.float64)
    np.testing.assert_allclose(oracle.mf_rhs(st.geo, q, st.z0s), rhs_ref, rtol=1e-5, atol=1e-5)
    # energy: direct evaluation == expansion z'A'Az - 2 z'A'b + b'b
    z = st.z.astype(np.float64)
    e_direct = oracle.energy(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z, st.fx, st.fy)
    Az = A.astype(np.float64) @ z
    e_asm = np.sum((st.geo.KT @ z - st.z0s) ** 2) + np.sum((Az - B) ** 2)
    assert abs(e_direct - e_asm) / e_asm < 1e-5


def test_albedo_closed_form_is_the_cg_fixed_point(oracle, small_state):
    sc, st = small_state
    rho_cg = oracle.albedo_estimation(st.s, st.rho.copy(), st.N, st.I)
    rho_cf = oracle.albedo_closed_form(st.s, st.rho, st.N, st.I)
    assert np.abs(rho_cg - rho_cf).max() < 5e-6
    num, den = oracle.albedo_numden(st.s, st.N, st.I)
    rho_nd = oracle.albedo_solve_numden(st.rho.copy(), num, den)
    assert np.abs(rho_nd - rho_cg).max() < 5e-6


def test_faithful_and_matrix_free_depth_steps_agree(oracle, small_state):
    sc, st = small_state
    z1 = st.z.copy(); z2 = st.z.copy()
    tr = []
    e1 = oracle.depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z1, st.fx, st.fy, cg_trace=tr)
    e2 = oracle.mf_depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z2, st.fx, st.fy)
    assert len(tr) == 101                                   # depth CG never converges to 1e-9: SURVEY 7.4-1
    assert np.sqrt(np.mean((z1 - z2) ** 2)) < 1e-4
    assert abs(e1 - e2) / e2 < 2e-2


def test_stop_rule(oracle, pkg):
    """SRPS.cu:297-302: NaN on the first pass continues; at most 11 passes; stops on rel < 5e-3"""
    sc = pkg.synth.make_scene(16, 16, 2, 3, seed=2, mask_kind="full")
    prob = oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init)
    st = oracle.execute(prob, depth="faithful")
    assert 2 <= st.iterations <= 11
    e = st.energies
    if st.iterations < 11:
        rel = abs(e[-2] - e[-1]) / abs(e[-1])
        assert e[-1] > e[-2] or rel < 5e-3
    for a, b in zip(e[:-2], e[1:-1]):                        # every earlier pass decreased the energy by >= 0.5 %
        assert b <= a and abs(a - b) / abs(b) >= 5e-3


def test_sharded_partial_sums_compose(oracle, small_state):
    """partial sums over disjoint image shards add up to the unsharded quantities"""
    sc, st = small_state
    n = sc.n_img
    num, den = oracle.albedo_numden(st.s, st.N, st.I)
    parts = [oracle.albedo_numden(st.s[a:b], st.N, st.I[a:b]) for a, b in ((0, 1), (1, n))]
    np.testing.assert_allclose(parts[0][0] + parts[1][0], num, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(parts[0][1] + parts[1][1], den, rtol=1e-5, atol=1e-6)
    M, q, _ = oracle.mf_tensor(st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    Ms, qs = zip(*[oracle.mf_tensor_split(st.s, st.s[a:b], st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I[a:b], dtype=np.float64)
                   for a, b in ((0, 1), (1, n))])
    np.testing.assert_allclose(Ms[0], M, rtol=1e-10); np.testing.assert_allclose(Ms[1], M, rtol=1e-10)
    np.testing.assert_allclose(qs[0] + qs[1], q, rtol=1e-9, atol=1e-12)
