"""CPU stand-in for the library context, built ONLY from oracle functions, with the phase-split
interface `alternating_loop` drives (lighting_local / *_partial / *_finish / exchange).  It exists so
that the host-side sharding logic (image partition, which buffers are summed, when) can be run under
gloo without a GPU.  Test infrastructure: never imported by the product."""
import numpy as np
import torch

f32 = np.float32


class OracleEngine:
    def __init__(self, O, scene):
        """scene: a synth.Scene holding this rank's image shard"""
        self.O = O
        prob = O.Problem(scene.h, scene.w, scene.sf, scene.mask, scene.K, scene.I, scene.zs_lr, scene.z_init)
        st = O.setup(prob)
        self.st = st
        self.lo = scene.img_offset
        self.hi = scene.img_offset + scene.n_img
        self.n_total = scene.n_img_total
        self.sharded = scene.n_img != scene.n_img_total
        s = np.zeros((self.n_total, scene.n_ch, 4), dtype=f32); s[:, :, 2] = -1
        self.s = s
        self.ex = {}

    # ---- phases --------------------------------------------------------------------------
    def lighting_local(self):
        st = self.st
        loc = self.s[self.lo:self.hi].copy()
        self.O.lighting_estimation(loc, st.rho, st.N, st.I)
        if self.sharded:
            self.s[:] = 0
        self.s[self.lo:self.hi] = loc
        self.ex["s"] = torch.from_numpy(self.s.reshape(-1))

    def albedo_partial(self):
        st = self.st
        num, den = self.O.albedo_numden(self.s[self.lo:self.hi], st.N, st.I)
        if self.sharded:
            # den = sum_i (N . s_ic)^2 does not involve the images: every rank forms it over ALL images from the all-reduced s
            # (srps_albedo_partial does the same); only num is exchanged
            sh = np.einsum("ick,kp->icp", self.s.astype(f32), st.N.astype(f32)).astype(f32)
            den = (sh * sh).sum(axis=0, dtype=f32)
        self.numden = np.ascontiguousarray(np.stack([num, den]), dtype=f32)
        self.ex["albedo"] = torch.from_numpy(self.numden[0].reshape(-1))      # shares memory with numden[0]

    def albedo_finish(self):
        self.O.albedo_solve_numden(self.st.rho, self.numden[0], self.numden[1])

    def depth_partial(self):
        st = self.st
        self.M, self.q = self.O.mf_tensor_split(self.s, self.s[self.lo:self.hi], st.rho, st.dz, st.xx, st.yy, st.fx, st.fy, st.I)
        self.q = np.ascontiguousarray(self.q, dtype=f32)
        self.ex["depth"] = torch.from_numpy(self.q.reshape(-1))

    def depth_solve(self):
        st = self.st; O = self.O
        M = self.M.astype(np.float64)
        z = st.z.astype(np.float64)
        rhs = O.mf_rhs(st.geo, self.q.astype(np.float64), st.z0s) - O.mf_apply(st.geo, M, z)
        O.conjugate_gradient(lambda v: O.mf_apply(st.geo, M, v), z, rhs, dtype=np.float64)
        st.z[:] = z.astype(f32)

    def energy_partial(self):
        st = self.st
        t1, t2 = self.O.energy_split(st.geo, self.s[self.lo:self.hi], st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, st.z, st.fx, st.fy)
        self.t1 = t1
        self.t2 = np.array([t2], dtype=np.float64)
        self.ex["energy"] = torch.from_numpy(self.t2)

    def energy_finish(self):
        return float(self.t1 + float(self.O.LAMBDA) * self.t2[0])

    def normals(self):
        st = self.st
        zx = (st.geo.Dx @ st.z).astype(f32); zy = (st.geo.Dy @ st.z).astype(f32)
        st.N, st.dz = self.O.normal_init(st.z, zx, zy, st.xx, st.yy, st.fx, st.fy)

    def exchange(self, which):
        return self.ex[which]
