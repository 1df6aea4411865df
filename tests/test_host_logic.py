"""Host-side logic that needs no GPU: image sharding, the stop rule of the alternating loop
(SRPS.cu:297-302) and the synthetic-scene generator."""
import numpy as np
import pytest


def test_shard_range_partitions_contiguously(pkg):
    for n in (1, 5, 20, 40, 64):
        for world in (1, 2, 3, 4, 8):
            pieces = [pkg.shard_range(n, world, r) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
            sizes = [b - a for a, b in pieces]
            assert max(sizes) - min(sizes) <= 1


class FakeEngine:
    """returns a scripted energy sequence; counts phase calls"""
    def __init__(self, energies):
        self.e = list(energies); self.k = 0; self.calls = []
    def __getattr__(self, name):
        def f(*a):
            self.calls.append(name)
        return f
    def energy_finish(self):
        v = self.e[self.k]; self.k += 1
        return v


def test_stop_rule_of_the_alternating_loop(pkg):
    # NaN on the first pass does not stop (SRPS.cu:273, 298-299); stops when the relative change < 5e-3
    en = pkg.alternating_loop(FakeEngine([10.0, 8.0, 7.99, 1.0]))
    assert en == pytest.approx([10.0, 8.0, 7.99])
    # an increase of the energy stops the loop and keeps that pass (SRPS.cu:299)
    assert pkg.alternating_loop(FakeEngine([10.0, 8.0, 9.0, 1.0])) == [10.0, 8.0, 9.0]
    # 'iteration > MAX_ITERATIONS' is tested before the increment => 11 passes at most
    seq = [100.0 * 0.9 ** k for k in range(30)]
    assert len(pkg.alternating_loop(FakeEngine(seq))) == 11
    assert len(pkg.alternating_loop(FakeEngine(seq), max_outer=3)) == 3
    # phase order of one pass (SRPS.cu:281-315)
    fe = FakeEngine([1.0])
    pkg.alternating_loop(fe, max_outer=1)
    assert fe.calls == ["lighting_local", "albedo_partial", "albedo_finish", "depth_partial", "depth_solve", "energy_partial", "normals"]
    # with an all-reduce the four exchange buffers are summed, in this order
    seen = []
    class E2(FakeEngine):
        def exchange(self, which):
            return which
    pkg.alternating_loop(E2([1.0]), all_reduce=seen.append, max_outer=1)
    assert seen == ["s", "albedo", "depth", "energy"]


def test_synthetic_scene_is_consistent_with_the_oracle_operators(pkg, oracle):
    sc = pkg.synth.make_scene(24, 28, 2, 3, seed=4, mask_kind="ragged", noise_I=0.0)
    geo = oracle.build_geometry(sc.h, sc.w, sc.sf, sc.mask)
    sel = sc.mask == 1
    # the generator's masked gradients are the oracle's Dx, Dy
    zt = sc.z_true[sel].astype(np.float64)
    m2 = pkg.synth.from_cm(sc.mask, sc.h, sc.w)
    zx, zy = pkg.synth.masked_gradients(pkg.synth.from_cm(sc.z_true, sc.h, sc.w).astype(np.float64), m2)
    np.testing.assert_allclose(pkg.synth.to_cm(zx)[sel], geo.Dx @ zt, atol=1e-6)
    np.testing.assert_allclose(pkg.synth.to_cm(zy)[sel], geo.Dy @ zt, atol=1e-6)
    # noise-free images are exactly rho * (s . [n; 1]) with the oracle's normals of the true depth
    xx, yy = oracle.meshgrid_masked(geo, sc.K)
    N, dz = oracle.normal_init(zt, geo.Dx @ zt, geo.Dy @ zt, xx, yy, sc.K[0], sc.K[4])
    for i in range(3):
        pred = sc.rho_true[:, sel] * (sc.s_true[i] @ N)
        np.testing.assert_allclose(sc.I[i][:, sel], np.clip(pred, 0, 1), atol=2e-5)
    # shards are slices of the same scene
    part = pkg.synth.make_scene(24, 28, 2, 3, seed=4, mask_kind="ragged", noise_I=0.0, img_begin=1, img_end=3)
    np.testing.assert_array_equal(part.I, sc.I[1:3]); assert part.img_offset == 1 and part.n_img_total == 3
