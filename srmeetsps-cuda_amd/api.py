"""Host-side mirror of the reference's SRPS / DataHandler / Preferences surface on top of the C ABI.

Reference interface mirrored here (paths under /root/reference/SRmeetsPS-GPU/):
  * ``Preferences``   -- Utilities.h:224-230 (blockX, blockY, deviceId; Main.cpp:5-7)
  * ``DataHandler``   -- Utilities.h:166-181 (I, mask, K, sf, z0 + sizes)
  * ``SRPS``          -- SRPS.h:10-18 (``SRPS(dh)``, ``execute()``), loop SRPS.cu:272-335
  * ``Context``       -- one method per C-ABI entry point (= one per cuda_based_* operator,
                         devicecalls.cuh:26-37)
The compute is entirely in libsrps_hip.so; PyTorch only provides device tensors, the current
stream and torch.distributed (RCCL) for the image-sharded mode.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import SRPSError, Problem, check

f32 = np.float32


class Preferences:
    """Utilities.h:224-230 / Main.cpp:5-7."""
    blockX = 256
    blockY = 4
    deviceId = 0


@dataclass
class DataHandler:
    """Utilities.h:166-181. All arrays are float32, flat column-major (index i + j*h).

    ``zs_lr`` / ``z_full`` are the outputs of the reference's CPU pre-processing
    (SRPS.cu:117-149: channel mean, inpainting, bilateral filter, cubic resize), which sits
    outside the hot path; ``preprocess()`` fills them with this package's stand-in."""
    I: np.ndarray            # [I_n][I_c][I_h*I_w]
    mask: np.ndarray         # [I_h*I_w] in {0,1}
    K: np.ndarray            # [9] column-major 3x3
    sf: int
    z0: np.ndarray           # [z0_n][z0_h*z0_w]
    I_h: int
    I_w: int
    I_c: int = 3
    I_n: int = 0             # images held locally
    I_n_total: int = 0       # images of the whole job
    image_offset: int = 0
    z0_n: int = 1
    zs_lr: np.ndarray | None = None
    z_full: np.ndarray | None = None
    I_u8: np.ndarray | None = None   # [I_n][I_c][I_h*I_w] uint8: the images as the bytes the image-folder loader read (I = byte / 255.f,
                                     # Utilities.cpp:343); when set, the bytes are what crosses PCIe and I may be None

    @property
    def z0_h(self):
        return self.I_h // self.sf

    @property
    def z0_w(self):
        return self.I_w // self.sf

    @classmethod
    def from_scene(cls, sc) -> "DataHandler":
        return cls(I=sc.I, mask=sc.mask, K=sc.K, sf=sc.sf, z0=sc.z0.reshape(1, -1), I_h=sc.h, I_w=sc.w,
                   I_c=sc.n_ch, I_n=sc.n_img, I_n_total=sc.n_img_total, image_offset=sc.img_offset,
                   z0_n=1, zs_lr=sc.zs_lr, z_full=sc.z_init)


def _ptr(x):
    """device pointer of a torch tensor (float32/int32, contiguous) or a raw integer address"""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous(), "device arrays must be contiguous"
        return C.c_void_p(x.data_ptr())
    raise TypeError(f"cannot take a device pointer of {type(x)}")


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _host(a, dtype=f32):
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


class _DevView:
    """zero-copy view of library-owned device memory for torch.as_tensor"""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class Context:
    """Owns one ``srps_ctx``.  Replaces cudaSetDevice + the cuBLAS/cuSPARSE handles of SRPS.cu:88-98."""

    def __init__(self, device_id: int | None = None, block_x: int | None = None, block_y: int | None = None,
                 stream: int | None = None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.srps_create(Preferences.deviceId if device_id is None else device_id,
                                   Preferences.blockX if block_x is None else block_x,
                                   Preferences.blockY if block_y is None else block_y, C.byref(h)))
        self.h = h
        self._keep = []
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.srps_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing --------------------------------------------------------------------------
    def set_stream(self, stream: int | None):
        check(self.lib.srps_set_stream(self.h, C.c_void_p(stream) if stream else None))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream)

    def synchronize(self):
        check(self.lib.srps_synchronize(self.h))

    def set_option(self, name: str, value: int):
        check(self.lib.srps_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        check(self.lib.srps_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    # -- operator level (devicecalls.cuh) ------------------------------------------------------
    def host_COO_to_device_CSR(self, row, col, val, n_row, n_col, d_row_ptr, d_col_ind, d_val):
        row = _host(row, np.int32); col = _host(col, np.int32); val = _host(val)
        check(self.lib.srps_host_COO_to_device_CSR(self.h, row.ctypes.data_as(C.POINTER(C.c_int)),
                                                   col.ctypes.data_as(C.POINTER(C.c_int)), _fptr(val), n_row, n_col,
                                                   int(row.size), _ptr(d_row_ptr), _ptr(d_col_ind), _ptr(d_val)))

    def sparsemat_densevec_mul(self, d_row_ptr, d_col_ind, d_val, n_rows, n_cols, nnz, d_x, d_y, transpose=False):
        check(self.lib.srps_sparsemat_densevec_mul(self.h, _ptr(d_row_ptr), _ptr(d_col_ind), _ptr(d_val), n_rows, n_cols,
                                                   nnz, _ptr(d_x), 1 if transpose else 0, _ptr(d_y)))

    def conjugate_gradient(self, d_row_ptr, d_col_ind, d_val, n, nnz, d_x, d_b) -> int:
        it = C.c_int(0)
        check(self.lib.srps_conjugate_gradient(self.h, _ptr(d_row_ptr), _ptr(d_col_ind), _ptr(d_val), n, nnz, _ptr(d_x),
                                               _ptr(d_b), C.byref(it)))
        return it.value

    def mean_across_channels(self, data_host, h, w, nc, d_mean, d_inpaint):
        a = _host(data_host)
        check(self.lib.srps_mean_across_channels(self.h, _fptr(a), h, w, nc, _ptr(d_mean), _ptr(d_inpaint)))

    def rho_init(self, d_rho, npix, nc):
        check(self.lib.srps_rho_init(self.h, _ptr(d_rho), npix, nc))

    def meshgrid_create(self, w, h, K02, K12, d_xx, d_yy):
        check(self.lib.srps_meshgrid_create(self.h, w, h, float(K02), float(K12), _ptr(d_xx), _ptr(d_yy)))

    def normal_init(self, d_z, d_zx, d_zy, d_xx, d_yy, npix, K00, K11, d_N, d_dz):
        check(self.lib.srps_normal_init(self.h, _ptr(d_z), _ptr(d_zx), _ptr(d_zy), _ptr(d_xx), _ptr(d_yy), npix,
                                        float(K00), float(K11), _ptr(d_N), _ptr(d_dz)))

    def lightning_estimation(self, d_s, d_rho, d_N, d_I, npix, nimages, nchannels):
        check(self.lib.srps_lightning_estimation(self.h, _ptr(d_s), _ptr(d_rho), _ptr(d_N), _ptr(d_I), npix, nimages, nchannels))

    def albedo_estimation(self, d_s, d_rho, d_N, d_I, npix, nimages, nchannels):
        check(self.lib.srps_albedo_estimation(self.h, _ptr(d_s), _ptr(d_rho), _ptr(d_N), _ptr(d_I), npix, nimages, nchannels))

    def bind_grid(self, h, w, sf, mask_host):
        m = _host(mask_host)
        assert m.size == h * w
        check(self.lib.srps_bind_grid(self.h, h, w, int(sf), _fptr(m)))

    def depth_estimation(self, d_s, d_rho, d_N, d_I, d_xx, d_yy, d_dz, d_z0s, d_z, K00, K11, npix, nimages, nchannels) -> float:
        e = C.c_float(0)
        check(self.lib.srps_depth_estimation(self.h, _ptr(d_s), _ptr(d_rho), _ptr(d_N), _ptr(d_I), _ptr(d_xx), _ptr(d_yy),
                                             _ptr(d_dz), _ptr(d_z0s), _ptr(d_z), float(K00), float(K11), npix, nimages,
                                             nchannels, C.byref(e)))
        return e.value

    def depth_estimation_csr(self, d_s, d_rho, d_N, d_I, d_xx, d_yy, d_dz, Dx, Dy, KT, d_z0s, d_z, K00, K11, npix, nimages, nchannels) -> float:
        """the reference's argument list (devicecalls.cuh:36): Dx, Dy, KT = (d_row_ptr, d_col_ind, d_val, n_rows, n_cols, nnz)"""
        e = C.c_float(0)
        mats = []
        for M in (Dx, Dy, KT):
            mats += [_ptr(M[0]), _ptr(M[1]), _ptr(M[2]), int(M[3]), int(M[4]), int(M[5])]
        check(self.lib.srps_depth_estimation_csr(self.h, _ptr(d_s), _ptr(d_rho), _ptr(d_N), _ptr(d_I), _ptr(d_xx), _ptr(d_yy), _ptr(d_dz),
                                                 *mats, _ptr(d_z0s), _ptr(d_z), float(K00), float(K11), npix, nimages, nchannels, C.byref(e)))
        return e.value

    def set_principal_point(self, K02, K12):
        check(self.lib.srps_set_principal_point(self.h, float(K02), float(K12)))

    def gradient(self, d_z, npix, d_zx, d_zy):
        check(self.lib.srps_gradient(self.h, _ptr(d_z), npix, _ptr(d_zx), _ptr(d_zy)))

    def depth_operator_apply(self, d_x, npix, d_y):
        check(self.lib.srps_depth_operator_apply(self.h, _ptr(d_x), npix, _ptr(d_y)))

    # -- pipeline level (SRPS::execute) --------------------------------------------------------
    def setup(self, dh: DataHandler):
        assert dh.zs_lr is not None and dh.z_full is not None, "run the depth pre-processing first"
        mask = _host(dh.mask); K = _host(dh.K); zs = _host(dh.zs_lr); zf = _host(dh.z_full)
        I8 = np.ascontiguousarray(dh.I_u8, dtype=np.uint8) if getattr(dh, "I_u8", None) is not None else None
        I = _host(dh.I) if (dh.I is not None and I8 is None) else None
        n_loc = dh.I_n if dh.I_n else (I.shape[0] if I is not None else (I8.shape[0] if I8 is not None else 0))
        n_tot = dh.I_n_total if dh.I_n_total else n_loc
        pr = Problem(dh.I_h, dh.I_w, dh.I_c, n_loc, n_tot, dh.image_offset, int(dh.sf),
                     _fptr(mask), _fptr(K), _fptr(I) if I is not None else None, _fptr(zs), _fptr(zf),
                     I8.ctypes.data_as(C.POINTER(C.c_ubyte)) if I8 is not None else None)
        self._keep = [mask, K, zs, zf, I, I8]
        check(self.lib.srps_setup(self.h, C.byref(pr)))
        self._keep = []

    def upload_image(self, local_index: int, image_host):
        a = np.asarray(image_host)
        if a.dtype == np.uint8:                      # the bytes of the image loader: I = byte / 255.f is formed on the device
            a = np.ascontiguousarray(a)
            check(self.lib.srps_upload_image_u8(self.h, local_index, a.ctypes.data_as(C.POINTER(C.c_ubyte))))
            return
        a = _host(image_host)
        check(self.lib.srps_upload_image(self.h, local_index, _fptr(a)))

    def dims(self) -> dict:
        v = [C.c_int(0) for _ in range(6)]
        check(self.lib.srps_dims(self.h, *[C.byref(x) for x in v]))
        return dict(zip(["npix", "npixs", "grid_h", "grid_w", "n_images", "n_channels"], [x.value for x in v]))

    def lighting(self): check(self.lib.srps_lighting(self.h))
    def albedo(self): check(self.lib.srps_albedo(self.h))
    def normals(self): check(self.lib.srps_normals(self.h))
    def lighting_local(self): check(self.lib.srps_lighting_local(self.h))
    def albedo_partial(self): check(self.lib.srps_albedo_partial(self.h))
    def albedo_finish(self): check(self.lib.srps_albedo_finish(self.h))
    def depth_partial(self): check(self.lib.srps_depth_partial(self.h))
    def depth_solve(self): check(self.lib.srps_depth_solve(self.h))
    def energy_partial(self): check(self.lib.srps_energy_partial(self.h))

    def depth(self) -> float:
        e = C.c_float(0)
        check(self.lib.srps_depth(self.h, C.byref(e)))
        return e.value

    def energy_finish(self) -> float:
        e = C.c_float(0)
        check(self.lib.srps_energy_finish(self.h, C.byref(e)))
        return e.value

    def exchange_ptr(self, which: str):
        p = C.c_void_p(); n = C.c_size_t(0)
        check(self.lib.srps_exchange(self.h, which.encode(), C.byref(p), C.byref(n)))
        return p.value, n.value

    def exchange(self, which: str):
        """torch view (no copy) of the exchange buffer the last *_partial call filled"""
        import torch
        ptr, n = self.exchange_ptr(which)
        return torch.as_tensor(_DevView(ptr, n), device=f"cuda:{torch.cuda.current_device()}")

    def execute(self, max_outer: int = 0):
        buf = (C.c_float * 64)()
        n = C.c_int(0)
        check(self.lib.srps_execute(self.h, max_outer, buf, C.byref(n)))
        return [buf[i] for i in range(min(n.value, 64))]

    # -- multi-GPU through the boundary: RCCL inside the library (srps.h "multi-GPU through the boundary") ----------
    COMM_ID_BYTES = 128

    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId through the library: rank 0 makes it, the launcher hands its bytes to every rank"""
        buf = C.create_string_buffer(Context.COMM_ID_BYTES)
        check(_lib.load().srps_comm_unique_id(buf))
        return buf.raw

    def comm_init_rank(self, uid: bytes, rank: int, world: int):
        assert len(uid) == self.COMM_ID_BYTES
        check(self.lib.srps_comm_init_rank(self.h, C.c_char_p(uid), int(rank), int(world)))

    @staticmethod
    def comm_init_all(contexts):
        """one process, one context per device: ncclCommInitAll; rank = position in `contexts`"""
        arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
        check(_lib.load().srps_comm_init_all(arr, len(contexts)))

    @staticmethod
    def strip_group_solve(contexts):
        """the depth solve of `contexts` (one process, one device, one stream) as the ranks of the strip-partitioned CG"""
        arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
        check(_lib.load().srps_strip_group_solve(arr, len(contexts)))

    @staticmethod
    def strip_group_solve_resident(contexts):
        """the same with the resident CG kernel on every strip: one persistent launch per context, each on its own stream, sums and
        border edges exchanged through each other's memory while the kernels run"""
        arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
        check(_lib.load().srps_strip_group_solve_resident(arr, len(contexts)))

    def comm_release(self):
        check(self.lib.srps_comm_release(self.h))

    def comm_info(self) -> tuple[int, int]:
        r = C.c_int(0); w = C.c_int(0)
        check(self.lib.srps_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def all_reduce(self, which: str):
        check(self.lib.srps_all_reduce(self.h, which.encode()))

    def execute_sharded(self, max_outer: int = 0):
        """the alternating loop on a shard with the all-reduces inside the library (ncclAllReduce on the context's stream)"""
        buf = (C.c_float * 64)()
        n = C.c_int(0)
        check(self.lib.srps_execute_sharded(self.h, max_outer, buf, C.byref(n)))
        return [buf[i] for i in range(min(n.value, 64))]

    def get(self, name: str) -> np.ndarray:
        n = C.c_size_t(0)
        check(self.lib.srps_array_size(self.h, name.encode(), C.byref(n)))
        out = np.empty(n.value, dtype=f32)
        check(self.lib.srps_get(self.h, name.encode(), _fptr(out), n.value))
        return out

    def device_ptr(self, name: str) -> tuple[int, int]:
        """(device address, length in floats) of a state array; the caller may write through it (srps.h: srps_get_device_ptr)"""
        p = C.c_void_p(); n = C.c_size_t(0)
        check(self.lib.srps_get_device_ptr(self.h, name.encode(), C.byref(p), C.byref(n)))
        return int(p.value or 0), int(n.value)

    def set(self, name: str, value):
        a = _host(value).reshape(-1)
        check(self.lib.srps_set(self.h, name.encode(), _fptr(a), a.size))

    def last_cg_iterations(self) -> dict:
        d = C.c_int(0); l = C.c_int(0); a = (C.c_int * 8)()
        check(self.lib.srps_last_cg_iterations(self.h, C.byref(d), a, C.byref(l)))
        return {"depth": d.value, "albedo": list(a), "lighting_max": l.value}

    def timings(self) -> dict:
        """milliseconds per pipeline phase since the last call (option "phase_timing" must be on); replaces the host Timer of
        SRPS.cu:277-295 -- HIP events on the stream, no synchronisation per phase"""
        n = 7
        ms = (C.c_float * n)()
        check(self.lib.srps_get_timings(self.h, ms))
        return {self.lib.srps_phase_name(k).decode(): ms[k] for k in range(n) if ms[k] >= 0}

    def bench_cg(self, solves: int, iters: int = 101) -> dict:
        s = C.c_double(0); a = C.c_double(0); u = C.c_double(0)
        check(self.lib.srps_bench_cg(self.h, solves, iters, C.byref(s), C.byref(a), C.byref(u)))
        ab = C.c_double(0); ub = C.c_double(0)
        check(self.lib.srps_cg_bytes(self.h, C.byref(ab), C.byref(ub)))
        return {"seconds": s.value, "apply_us": a.value, "update_us": u.value, "apply_bytes": ab.value,
                "update_bytes": ub.value, "iterations": solves * iters}


# --------------------------------------------------------------------------------------------
# the alternating loop, host side (SRPS.cu:272-335).  `engine` is a Context; the test-suite drives
# the same function with a CPU stand-in engine to exercise the sharding logic under gloo.
# --------------------------------------------------------------------------------------------
TOLERANCE = 5e-3          # SRPS.cu:85
MAX_ITERATIONS = 10       # SRPS.cu:86


def shard_range(n_images: int, world: int, rank: int) -> tuple[int, int]:
    """contiguous image shard [begin, end) of rank `rank`; sizes differ by at most 1"""
    base, rem = divmod(n_images, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def alternating_loop(engine, all_reduce=None, max_outer: int | None = None, on_iteration=None) -> list[float]:
    """lighting -> albedo -> depth -> stop test -> normals.  With `all_reduce` (a callable that sums a
    tensor in place over the ranks) the per-image phases are sharded: every rank calls the
    *_partial phase on its images, the partial sums are all-reduced, and every rank finishes the
    phase on identical data (replicated CG)."""
    ar = all_reduce
    last_error = float("nan")             # SRPS.cu:273
    iteration = 1
    energies: list[float] = []
    while True:
        engine.lighting_local()                                   # SRPS.cu:281
        if ar: ar(engine.exchange("s"))
        engine.albedo_partial()                                   # SRPS.cu:287
        if ar: ar(engine.exchange("albedo"))
        engine.albedo_finish()
        engine.depth_partial()                                    # SRPS.cu:293
        if ar: ar(engine.exchange("depth"))
        engine.depth_solve()
        engine.energy_partial()
        if ar: ar(engine.exchange("energy"))
        # The normals of the new depth (SRPS.cu:310-315) do not depend on the stop test: they are enqueued before the host
        # waits for the energy, so the GPU is not idle during the read-back.
        engine.normals()
        error = float(f32(engine.energy_finish()))
        with np.errstate(invalid="ignore", divide="ignore"):
            rel_err = float(abs(f32(last_error) - f32(error)) / abs(f32(error)))       # SRPS.cu:298
        stop = (error > last_error) or (rel_err < TOLERANCE) or (iteration > MAX_ITERATIONS)   # SRPS.cu:299
        last_error = error
        energies.append(error)
        if on_iteration:
            on_iteration(iteration, error, rel_err)
        iteration += 1
        if stop or (max_outer is not None and len(energies) >= max_outer):
            break
    return energies


class SRPS:
    """SRPS.h:10-18.  ``SRPS(dh).execute()`` runs the whole alternating optimisation on the GPU.

    With ``torch.distributed`` initialised (one process per GPU, backend "nccl" = RCCL) and
    ``distributed=True`` the images of ``dh`` are taken to be this rank's shard."""

    def __init__(self, dh: DataHandler, distributed: bool = False, ctx: Context | None = None):
        self.dh = dh
        self.distributed = distributed
        self.ctx = ctx or Context()
        self.energies: list[float] = []

    def execute(self, max_outer: int | None = None, verbose: bool = False):
        cb = None
        if verbose:
            def cb(it, err, rel):
                print(f"\nIteration {it:02d} summary\n{'Error':<25}: {err:<6.3f}\n{'Relative Error':<25}: {rel:<6.3f}")
        if self.distributed:
            import torch
            import torch.distributed as dist
            # One explicit (non-default) torch stream carries BOTH the library's kernels and the
            # collectives' stream dependencies: torch.distributed orders a collective after the work
            # already enqueued on the current stream and makes the current stream wait for its result.
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                self.ctx.set_stream(stream.cuda_stream)
                self.ctx.setup(self.dh)

                def ar(t):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                self.energies = alternating_loop(self.ctx, ar, max_outer, cb)
                self.ctx.synchronize()
            self.ctx.set_stream(None)
            return self.energies
        self.ctx.setup(self.dh)
        if verbose or max_outer is not None:
            self.energies = alternating_loop(self.ctx, None, max_outer, cb)
        else:
            self.energies = self.ctx.execute(0)       # the loop inside the library (C++)
        self.ctx.synchronize()
        return self.energies

    # results in the reference's layouts
    def z(self): return self.ctx.get("z")
    def rho(self): return self.ctx.get("rho").reshape(self.dh.I_c, -1)
    def s(self): return self.ctx.get("s").reshape(-1, self.dh.I_c, 4)
    def N(self): return self.ctx.get("N").reshape(4, -1)


class TorchCollectives:
    """The two host functions of srps_set_host_collectives on torch.distributed (any backend, e.g. gloo): the library's own sharded loop
    (srps_execute_sharded), the hipIpc handshake and the barrier of the resident strips (cg_partition = 2) then run over the caller's
    process group -- between processes that share one GPU (where RCCL refuses a second rank) as between GPUs.  Host functions on device
    pointers: every call drains the stream; a dry-run / portability path, not the fast one (that is RCCL inside the library)."""

    def __init__(self, ctx, dist, device: str = "cuda:0"):
        import torch
        self.ctx, self.errors = ctx, []

        def view(ptr, n, typestr):
            class V:
                pass
            v = V()
            v.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(v, device=device)

        def to_host(g):
            # into PINNED host memory (torch's caching host allocator): the device does not touch pageable pages (csrc/srps_xfer.hip says why)
            t = torch.empty(g.shape, dtype=g.dtype, pin_memory=True)
            t.copy_(g)
            return t

        def allreduce(user, d_buf, n, f64):
            try:
                g = view(d_buf, n, "<f8" if f64 else "<f4")
                t = to_host(g)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                g.copy_(t)
                torch.cuda.synchronize()
                return 0
            except Exception as exc:
                self.errors.append(repr(exc))
                return 1

        def broadcast(user, d_buf, n, root):
            try:
                g = view(d_buf, n, "<f4")
                t = to_host(g)
                dist.broadcast(t, src=root)
                g.copy_(t)
                torch.cuda.synchronize()
                return 0
            except Exception as exc:
                self.errors.append(repr(exc))
                return 1

        self._fns = (_lib.HOST_ALLREDUCE_FN(allreduce), _lib.HOST_BROADCAST_FN(broadcast))
        check(ctx.lib.srps_set_host_collectives(ctx.h, dist.get_rank(), dist.get_world_size(), *self._fns, None))

    def remove(self):
        check(self.ctx.lib.srps_set_host_collectives(self.ctx.h, 0, 1, _lib.HOST_ALLREDUCE_FN(), _lib.HOST_BROADCAST_FN(), None))
