"""The C restatement of the oracle (oracle/srps_oracle.c, the CPU baseline of bench.py) against the
numpy restatement: structure, tensor, assembled CSR pattern and values, operator, CG, energy."""
import os
import subprocess
import numpy as np
import pytest
import scipy.sparse as sp

f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def CO():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import c_oracle
    return c_oracle


@pytest.mark.parametrize("kind,sf,h,w", [("ragged", 2, 48, 40), ("full", 4, 32, 48), ("ragged", 1, 24, 20), ("ragged", 3, 36, 30), ("ellipse", 4, 64, 48)])
def test_c_oracle_matches_numpy_oracle(CO, oracle, pkg, kind, sf, h, w):
    sc = pkg.synth.make_scene(h, w, sf, 4, seed=5, mask_kind=kind)
    st = oracle.setup(oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init))
    oracle.lighting_estimation(st.s, st.rho, st.N, st.I); oracle.albedo_estimation(st.s, st.rho, st.N, st.I)
    cs = CO.Structure(h, w, sf, sc.mask)
    assert (cs.P, cs.Ps) == (st.geo.npix, st.geo.npixs)
    np.testing.assert_array_equal(cs.imask, st.geo.imask)
    x = np.random.default_rng(0).normal(size=cs.P).astype(f32)
    gx = np.empty(cs.P, f32); gy = np.empty(cs.P, f32)
    CO._L.oc_gradient(cs.P, CO._i(cs.nb), CO._f(x), CO._f(gx), CO._f(gy))
    np.testing.assert_allclose(gx, st.geo.Dx @ x, atol=1e-6); np.testing.assert_allclose(gy, st.geo.Dy @ x, atol=1e-6)
    M, q = CO.tensor(cs, st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    M64, q64, _ = oracle.mf_tensor(st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    assert np.abs(M.reshape(6, -1) - M64).max() / np.abs(M64).max() < 1e-6
    assert np.abs(q.reshape(3, -1) - q64).max() / np.abs(q64).max() < 1e-6
    A, A_, B = oracle.assemble_depth_system(st.geo, st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    rp, ci, v = CO.assemble(cs, M)
    Ac = sp.csr_matrix((v, ci, rp), shape=(cs.P, cs.P))
    assert Ac.nnz == A_.nnz and np.array_equal(Ac.indptr, A_.indptr)          # same sparsity as the SpGEMM result
    assert abs(Ac - A_).max() < 2e-6 * abs(A_).max()
    yr = A_.astype(np.float64) @ x
    assert np.linalg.norm(CO.mf_apply(cs, M, x) - yr) / np.linalg.norm(yr) < 1e-6
    assert np.linalg.norm(CO.csr_spmv(rp, ci, v, x) - yr) / np.linalg.norm(yr) < 1e-6
    z_np = st.z.copy(); e_np = oracle.depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z_np, st.fx, st.fy)
    for assembled in (True, False):
        z = st.z.copy()
        e, it = CO.depth_estimation(cs, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, z, st.fx, st.fy, assembled)
        assert it == 101
        assert np.sqrt(np.mean((z - z_np) ** 2)) < 1e-4
        assert abs(e - e_np) / e_np < 1e-3
