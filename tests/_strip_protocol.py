"""TEST INFRASTRUCTURE (moved out of the package in round 4: nothing in the product imports it -- the product's strip CG is
csrc/srps_strips.hip).  Strip-partitioned depth CG over several GPUs: host-side protocol (design for grids from 4096 x 4096 up, where the
replicated CG of the image-sharded mode is what bounds the pass).

The reference's CG (devicecalls.cu:229-279) on A_ = KT'KT + lambda A'A is kept step for step; what is partitioned is the
grid: rank r owns the columns [j_r, j_{r+1}) of the bounding box, cut at multiples of sf so that no sf x sf block of KT
(SRPS.cu:176-190) straddles two ranks.  A row of A_ at an owned pixel references, besides its own strip, only the ONE column
to the left and the ONE to the right of it (Dx, Dx' are forward / backward differences between horizontal neighbours;
Dy, Dy' and KT'KT stay inside a column resp. a block) -- tests/test_strip_partition.py asserts this on the assembled
matrix.  Per CG step a rank therefore needs

  * the edge columns of p from its two neighbours (2 x Hg floats each way: point-to-point, xGMI peer copies), and
  * the two dot products of the step as all-reduces of ONE float each (p.A p, then r.r) -- or, in the one-wait form of
    kernels_resident.hip, one all-reduce of three floats.

At 4096 x 4096 on 8 ranks a strip is 4096 x 512 = 128 tiles of 256 x 64: it fits the resident kernel (one tile per CU), so
the per-step cost becomes that kernel's ~10 us plus one small all-reduce instead of 170 us of streaming.

`strip_cg` below is that protocol written against an abstract engine (the part that will drive the HIP kernels); the test
suite runs it under gloo with an engine made from the oracle's assembled matrix and compares with the serial CG.

The GPU engine exists since round 3 (csrc/srps_strips.hip, option "cg_partition"): RCCL between the launches on several GPUs,
device copies between several contexts of one process (srps_strip_group_solve), or -- `HostedTransport` below -- any transport
the caller brings (srps_set_strip_transport): here torch.distributed, which lets two PROCESSES on one GPU run the strips over gloo.
"""
from __future__ import annotations

import importlib

import numpy as np

CG_TOL = np.float32(1e-9)       # devicecalls.cu:230
CG_MAX_ITER = 100               # devicecalls.cu:231 ("k <= max_iter": up to 101 steps)


def strip_ranges(n_cols: int, sf: int, world: int) -> list[tuple[int, int]]:
    """column ranges [begin, end) of the ranks: multiples of sf, sizes differ by at most one block column"""
    assert n_cols % sf == 0 and world >= 1
    blocks = n_cols // sf
    base, rem = divmod(blocks, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < rem else 0)
        out.append((b * sf, e * sf))
        b = e
    return out


def strip_cg(engine, comm, max_iter: int = CG_MAX_ITER, tol: float = CG_TOL) -> int:
    """The reference's recurrence (devicecalls.cu:252-275) on a strip.

    engine (one per rank; all vectors live on its owned pixels):
        residual_init()          r = b - A_ x  (x halo already exchanged by the caller of the first exchange_halo("x"))
        edge(name) -> (left, right)   the first / last owned column of vector `name` ("x" or "p") as 1-D float32 tensors
        set_halo(name, left, right)   the neighbours' edge columns (None at the outer strips)
        apply() -> float         omega = A_ p on the owned pixels (uses the halo of p); returns the local p.omega
        dot_rr() -> float        local r.r
        update_x_r(alpha)        x += alpha p ; r -= alpha omega          (dc.cu:270-272)
        update_p(beta, first)    p = r (first) or p = beta p + r          (dc.cu:256-264)
    comm: exchange(left, right) -> (from_left, from_right) point-to-point with the neighbour ranks; allreduce(float) -> float
    Returns the number of steps executed."""
    def exchange_halo(name):
        left, right = engine.edge(name)
        engine.set_halo(name, *comm.exchange(left, right))

    exchange_halo("x")
    engine.residual_init()                                     # dc.cu:758
    r1 = np.float32(comm.allreduce(engine.dot_rr()))           # dc.cu:249
    r0 = np.float32(0)
    k = 0                                                      # dc.cu:242
    while r1 > np.float32(tol) * np.float32(tol) and k <= max_iter:        # dc.cu:252: tested before the increment => up to 101 steps
        k += 1
        if k > 1:
            engine.update_p(np.float32(r1 / r0), False)        # dc.cu:262-264
        else:
            engine.update_p(np.float32(0), True)               # dc.cu:258
        exchange_halo("p")
        dot = np.float32(comm.allreduce(engine.apply()))       # dc.cu:267-268
        alpha = np.float32(r1 / dot)                           # dc.cu:269
        engine.update_x_r(alpha)
        r0 = r1
        r1 = np.float32(comm.allreduce(engine.dot_rr()))       # dc.cu:274
    return k


class TorchDistComm:
    """comm for `strip_cg` on torch.distributed (gloo on CPU in the tests, nccl = RCCL over xGMI on GPUs)"""

    def __init__(self, dist, device=None):
        import torch
        self.dist, self.torch, self.device = dist, torch, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def exchange(self, left, right):
        torch, dist = self.torch, self.dist
        from_left = torch.empty_like(left) if self.rank > 0 else None
        from_right = torch.empty_like(right) if self.rank < self.world - 1 else None
        ops = []
        if self.rank > 0:
            ops += [dist.P2POp(dist.isend, left, self.rank - 1), dist.P2POp(dist.irecv, from_left, self.rank - 1)]
        if self.rank < self.world - 1:
            ops += [dist.P2POp(dist.isend, right, self.rank + 1), dist.P2POp(dist.irecv, from_right, self.rank + 1)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return from_left, from_right

    def allreduce(self, value: float) -> float:
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())



class HostedTransport:
    """The three host functions of srps_set_strip_transport on torch.distributed (gloo in the tests: two processes that share one
    GPU cannot form an RCCL communicator).  Device memory is reached through zero-copy torch views of the pointers the library
    hands over; the collectives run on CPU copies."""

    def __init__(self, ctx, dist, device: str = "cuda:0"):
        import ctypes as C
        import torch
        _lib = importlib.import_module("srmeetsps-cuda_amd._lib")
        self.ctx, self.dist, self.torch, self.device = ctx, dist, torch, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.errors = []

        def view(ptr, n, typestr="<f4"):
            class V:
                pass
            v = V()
            v.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(v, device=device)

        def guarded(fn):
            def wrapper(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as exc:          # an exception must not cross the C boundary
                    self.errors.append(repr(exc))
                    return 1
            return wrapper

        def to_host(g):                       # pinned host memory: the device never touches pageable pages (csrc/srps_xfer.hip)
            t = torch.empty(g.shape, dtype=g.dtype, pin_memory=True)
            t.copy_(g)
            return t

        def host_buf(n):
            return torch.empty(n, dtype=torch.float32, pin_memory=True)

        @guarded
        def allreduce(user, d_in, d_out):
            t = to_host(view(d_in, 4, "<f8"))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            view(d_out, 4, "<f8").copy_(t)
            torch.cuda.synchronize()

        @guarded
        def exchange(user, nbuf, sl, rl, sr, rr, n):
            ops, landing = [], []
            for b in range(nbuf):
                if sl:
                    out = to_host(view(sl[b], n)); inn = host_buf(n)
                    ops += [dist.P2POp(dist.isend, out, self.rank - 1), dist.P2POp(dist.irecv, inn, self.rank - 1)]
                    landing.append((rl[b], inn))
                if sr:
                    out = to_host(view(sr[b], n)); inn = host_buf(n)
                    ops += [dist.P2POp(dist.isend, out, self.rank + 1), dist.P2POp(dist.irecv, inn, self.rank + 1)]
                    landing.append((rr[b], inn))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for ptr, t in landing:
                view(ptr, n).copy_(t)
            torch.cuda.synchronize()

        @guarded
        def allgather(user, d_x, offset, count):
            for q in range(self.world):
                if count[q] == 0:
                    continue
                piece = view(d_x + 4 * offset[q], count[q])
                t = to_host(piece) if q == self.rank else host_buf(count[q])
                dist.broadcast(t, src=q)
                if q != self.rank:
                    piece.copy_(t)
            torch.cuda.synchronize()

        # the ctypes objects must outlive the registration
        self._fns = (_lib.STRIP_ALLREDUCE_FN(allreduce), _lib.STRIP_EXCHANGE_FN(exchange), _lib.STRIP_ALLGATHER_FN(allgather))
        _lib.check(ctx.lib.srps_set_strip_transport(ctx.h, self.rank, self.world, *self._fns, None))

    def remove(self):
        _lib = importlib.import_module("srmeetsps-cuda_amd._lib")
        _lib.check(self.ctx.lib.srps_set_strip_transport(self.ctx.h, 0, 1, _lib.STRIP_ALLREDUCE_FN(), _lib.STRIP_EXCHANGE_FN(), _lib.STRIP_ALLGATHER_FN(), None))


def HostedCollectives(ctx, dist, device: str = "cuda:0"):
    """moved into the package in round 4 (api.TorchCollectives: bench.py's one-GPU dry run of the sharded loop uses it too)"""
    api = importlib.import_module("srmeetsps-cuda_amd.api")
    return api.TorchCollectives(ctx, dist, device)


class ThreadCollectives:
    """srps_set_host_collectives for ranks that are THREADS of one process (what `srps --gpus N` and srps_comm_init_all are, with RCCL
    in the place of this): every rank's all-reduce / broadcast meets the others' at a threading.Barrier; the sum is formed in rank
    order, so all ranks hold the same bits.  Test infrastructure for one-GPU boxes, where RCCL refuses two ranks on one device."""

    def __init__(self, world: int):
        import threading
        self.world = world
        self.bar = threading.Barrier(world)
        self.slot = [None] * world
        self.errors = []
        self._keep = []

    def bind(self, pkg, ctx, rank: int, device: str = "cuda:0"):
        import torch
        lib = pkg._lib

        def view(ptr, n, typestr):
            class V:
                pass
            v = V()
            v.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(v, device=device)

        # Every rank's copies run on a NON-BLOCKING stream of its own and only that stream is waited for: a device-wide wait (or the
        # default stream, which waits for the others) inside one rank's collective would wait for the OTHER rank's resident kernel --
        # launched the moment that rank left the barrier, and itself waiting for this rank's kernel: neither would ever be launched
        # (seen as a fall-back in one run of three before this was a stream per rank).
        st = torch.cuda.Stream()

        def allreduce(user, d_buf, n, f64):
            try:
                with torch.cuda.stream(st):
                    g = view(d_buf, n, "<f8" if f64 else "<f4")
                    h = torch.empty(g.shape, dtype=g.dtype, pin_memory=True)
                    h.copy_(g, non_blocking=True)
                    st.synchronize()
                    self.slot[rank] = h
                    self.bar.wait(timeout=120)
                    tot = self.slot[0].clone()
                    for q in range(1, self.world):
                        tot += self.slot[q]
                    self.bar.wait(timeout=120)
                    g.copy_(tot.pin_memory(), non_blocking=True)
                    st.synchronize()
                return 0
            except Exception as exc:                    # a broken barrier ends the other ranks' waits too
                self.errors.append(repr(exc)); self.bar.abort()
                return 1

        def broadcast(user, d_buf, n, root):
            try:
                with torch.cuda.stream(st):
                    g = view(d_buf, n, "<f4")
                    if rank == root:
                        h = torch.empty(g.shape, dtype=g.dtype, pin_memory=True)
                        h.copy_(g, non_blocking=True)
                        st.synchronize()
                        self.slot[root] = h
                    self.bar.wait(timeout=120)
                    if rank != root:
                        g.copy_(self.slot[root], non_blocking=True)
                        st.synchronize()
                    self.bar.wait(timeout=120)
                return 0
            except Exception as exc:
                self.errors.append(repr(exc)); self.bar.abort()
                return 1

        fns = (lib.HOST_ALLREDUCE_FN(allreduce), lib.HOST_BROADCAST_FN(broadcast))
        self._keep.append(fns)
        pkg._lib.check(ctx.lib.srps_set_host_collectives(ctx.h, rank, self.world, *fns, None))

    def unbind(self, pkg, ctx):
        pkg._lib.check(ctx.lib.srps_set_host_collectives(ctx.h, 0, 1, pkg._lib.HOST_ALLREDUCE_FN(), pkg._lib.HOST_BROADCAST_FN(), None))
