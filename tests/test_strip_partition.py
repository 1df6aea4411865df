"""The strip-partitioned depth CG (tests/_strip_protocol.py) under gloo, world sizes 2 and 3, on CPU: the protocol --
column strips cut at multiples of sf, one-column halo of p per step, the step's dot products as all-reduces -- driven with an
engine built from the oracle's ASSEMBLED system (A_ = KT'KT + A'A and the right-hand side of devicecalls.cu:734-745), against
the oracle's serial CG.  Also the structural claim the design rests on: a row of A_ at an owned pixel references no column
further away than one."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32
H, W, SF, N_IMG = 24, 36, 2, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _system(O, synth, kind):
    """the depth system of the first outer pass (after lighting + albedo), assembled by the oracle"""
    sc = synth.make_scene(H, W, SF, N_IMG, seed=23, mask_kind=kind)
    st = O.setup(O.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init))
    O.lighting_estimation(st.s, st.rho, st.N, st.I)
    O.albedo_estimation(st.s, st.rho, st.N, st.I)
    A, A_, B = O.assemble_depth_system(st.geo, st.s, st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
    rhs = (st.geo.KT.T @ st.z0s + O.LAMBDA * (A.T @ B)).astype(f32)
    return st, A_.tocsr().astype(f32), rhs


class StripEngine:
    """engine of strips.strip_cg for one rank: rows of the assembled A_ at its owned pixels; vectors are kept global-sized (the
    halo entries live at their global positions), which keeps this stand-in short -- a GPU engine holds strip + halo only"""

    def __init__(self, st, A_, rhs, col_begin, col_end):
        import torch
        self.torch = torch
        h = st.geo.h
        jj = st.geo.imask // h                                    # image column of every masked pixel
        j0 = int(jj.min() // SF * SF)                             # bounding box, aligned to sf (as srps_bind_grid does)
        self.h = h
        self.rows_of = lambda j: np.flatnonzero(jj == j0 + j)     # compact indices of bounding-box column j
        self.own = np.flatnonzero((jj >= j0 + col_begin) & (jj < j0 + col_end))
        self.first_col, self.last_col = col_begin, col_end - 1
        self.A_own = A_[self.own]
        self.b = rhs
        P = st.geo.npix
        self.x = np.zeros(P, f32); self.x[self.own] = st.z[self.own]
        self.p = np.zeros(P, f32); self.r = np.zeros(P, f32); self.w = np.zeros(P, f32)
        self.imask_row = st.geo.imask % h
        # structural claim: the owned rows reference nothing beyond one column outside the strip
        cols = np.unique(self.A_own.indices)
        assert jj[cols].min() >= j0 + col_begin - 1 and jj[cols].max() <= j0 + col_end

    def _column(self, vec, j):
        out = np.zeros(self.h, f32)
        idx = self.rows_of(j)
        out[self.imask_row[idx]] = vec[idx]
        return self.torch.from_numpy(out)

    def edge(self, name):
        v = getattr(self, name)
        return self._column(v, self.first_col), self._column(v, self.last_col)

    def set_halo(self, name, left, right):
        v = getattr(self, name)
        for t, j in ((left, self.first_col - 1), (right, self.last_col + 1)):
            if t is not None:
                idx = self.rows_of(j)
                v[idx] = t.numpy()[self.imask_row[idx]]

    def residual_init(self):
        self.r[self.own] = (self.b[self.own] - self.A_own @ self.x).astype(f32)

    def apply(self):
        self.w[self.own] = (self.A_own @ self.p).astype(f32)
        return float(np.dot(self.p[self.own].astype(np.float64), self.w[self.own].astype(np.float64)))

    def dot_rr(self):
        return float(np.dot(self.r[self.own].astype(np.float64), self.r[self.own].astype(np.float64)))

    def update_x_r(self, alpha):
        self.x[self.own] += alpha * self.p[self.own]
        self.r[self.own] -= alpha * self.w[self.own]

    def update_p(self, beta, first):
        self.p[self.own] = self.r[self.own] if first else (beta * self.p[self.own] + self.r[self.own]).astype(f32)


def _worker(rank, world, port, kind, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _strip_protocol as strips
    synth = importlib.import_module("srmeetsps-cuda_amd.synth")
    import srps_oracle as O
    from test_strip_partition import StripEngine, _system
    st, A_, rhs = _system(O, synth, kind)
    jj = st.geo.imask // st.geo.h
    j0 = int(jj.min() // SF * SF); j1 = int(-(-(jj.max() + 1) // SF) * SF)
    lo, hi = strips.strip_ranges(j1 - j0, SF, world)[rank]
    eng = StripEngine(st, A_, rhs, lo, hi)
    steps = strips.strip_cg(eng, strips.TorchDistComm(dist))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), own=eng.own, x=eng.x[eng.own], steps=steps)
    dist.barrier()
    dist.destroy_process_group()


def test_strip_ranges(pkg):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _strip_protocol as strips
    for n, sf, world in ((4096, 2, 8), (36, 2, 3), (2048, 4, 5), (8, 4, 2)):
        r = strips.strip_ranges(n, sf, world)
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all(b % sf == 0 and e % sf == 0 for b, e in r)
        sizes = [e - b for b, e in r]
        assert max(sizes) - min(sizes) <= sf
    assert strips.strip_ranges(4096, 2, 8)[0] == (0, 512)          # 4096 x 512 strips: 128 tiles of 256 x 64, the resident kernel's size


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,kind", [(2, "ragged"), (3, "full")])
def test_strip_partitioned_cg_equals_serial_cg(tmp_path, pkg, oracle, world, kind):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, kind, str(tmp_path)), nprocs=world, join=True)
    st, A_, rhs = _system(oracle, pkg.synth, kind)
    # serial: the oracle's CG on the same assembled matrix (dc.cu:758-759)
    x = st.z.copy()
    b = (rhs - A_ @ x).astype(f32)
    it = oracle.conjugate_gradient(lambda v: (A_ @ v).astype(f32), x, b)
    got = np.zeros_like(x); seen = np.zeros(x.size, bool); steps = set()
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        got[d["own"]] = d["x"]; seen[d["own"]] = True; steps.add(int(d["steps"]))
    assert seen.all() and steps == {it} and it == 101               # every pixel owned once; the same truncated 101 steps
    assert np.sqrt(np.mean((got - x) ** 2)) < 2e-5                  # dot products are summed in another order, nothing else differs


def test_partitions_of_the_library_equal_the_python_ones():
    """srps_strip_range / srps_shard_range (the C side: srps_depth_solve with cg_partition, the C++ host's --gpus) against
    strips.strip_ranges / api.shard_range (the Python side) -- pure functions, no device -- and their invariants: the ranks'
    pieces tile the whole range in order, strips are cut at multiples of sf, sizes differ by at most one block / one image"""
    import ctypes as C
    pkg = importlib.import_module("srmeetsps-cuda_amd")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _strip_protocol as strips
    lib = pkg.load()
    a, b = C.c_int(0), C.c_int(0)
    for sf in (1, 2, 3, 4):
        for blocks in (1, 2, 7, 8, 64, 1023, 2048):
            cols = blocks * sf
            for world in (1, 2, 3, 4, 8, 16):
                if world > blocks:
                    continue
                ref = strips.strip_ranges(cols, sf, world)
                got = []
                for r in range(world):
                    assert lib.srps_strip_range(cols, sf, world, r, C.byref(a), C.byref(b)) == 0
                    got.append((a.value, a.value + b.value))
                assert got == ref
                assert got[0][0] == 0 and got[-1][1] == cols and all(x[1] == y[0] for x, y in zip(got, got[1:]))
                assert all((e - s) % sf == 0 and e > s for s, e in got)
                assert max(e - s for s, e in got) - min(e - s for s, e in got) <= sf
    for n in (0, 1, 5, 8, 20, 40, 64, 65):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                assert lib.srps_shard_range(n, world, r, C.byref(a), C.byref(b)) == 0
                got.append((a.value, a.value + b.value))
                assert got[-1] == pkg.shard_range(n, world, r)
            assert got[0][0] == 0 and got[-1][1] == n and all(x[1] == y[0] for x, y in zip(got, got[1:]))
            assert max(e - s for s, e in got) - min(e - s for s, e in got) <= 1
    assert lib.srps_strip_range(10, 4, 2, 0, C.byref(a), C.byref(b)) == 1          # columns not a multiple of sf
    assert lib.srps_shard_range(5, 2, 2, C.byref(a), C.byref(b)) == 1               # rank out of range
