"""ctypes binding of libsrps_hip.so (the C ABI declared in include/srps.h).

The HIP library is the product: if it is missing or cannot be loaded this module raises --
there is no CPU or PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
# SRPS_LIB_PATH: another build of the same library (development aid: same-box A/B of variants under srmeetsps-cuda_amd/variants/)
LIB_PATH = os.environ.get("SRPS_LIB_PATH") or os.path.join(_HERE, "libsrps_hip.so")
HEADER = os.path.join(ROOT, "include", "srps.h")

SRPS_OK = 0
ALBEDO_CG, ALBEDO_CLOSED_FORM, ALBEDO_FUSED, ALBEDO_AUTO = 0, 1, 2, 3
APPLY_AUTO, APPLY_SIMPLE, APPLY_MARCH = 0, 1, 2


class SRPSError(RuntimeError):
    """Raised for every non-zero status of the C ABI (the reference throws std::runtime_error on
    library failures, Utilities.cpp:21-31, and exits on CUDA errors, Utilities.cpp:8-19)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"srps error {code}: {msg}")
        self.code = code


class Problem(C.Structure):
    """struct srps_problem of include/srps.h (mirrors DataHandler, Utilities.h:166-181)."""
    _fields_ = [
        ("h", C.c_int), ("w", C.c_int), ("n_channels", C.c_int), ("n_images", C.c_int),
        ("n_images_total", C.c_int), ("image_offset", C.c_int), ("sf", C.c_int),
        ("mask", C.POINTER(C.c_float)), ("K", C.POINTER(C.c_float)), ("I", C.POINTER(C.c_float)),
        ("zs_lr", C.POINTER(C.c_float)), ("z_full", C.POINTER(C.c_float)), ("I_u8", C.POINTER(C.c_ubyte)),
    ]


# callbacks of srps_set_strip_transport (include/srps.h): host functions on device pointers
STRIP_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
STRIP_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t)
STRIP_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))
# callbacks of srps_set_host_collectives
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
HOST_BROADCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)


def build(force: bool = False) -> str:
    """Compile libsrps_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH) or _stale():
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"])
    return LIB_PATH


def _stale() -> bool:
    src_dir = os.path.join(_HERE, "csrc")
    t_lib = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h"))] + [HEADER]
    return any(os.path.getmtime(s) > t_lib for s in srcs)


def declared_symbols() -> list[str]:
    """Every function include/srps.h declares."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(srps_[a-zA-Z0-9_]+)\s*\(", txt)))


_lib = None


def load():
    """Load the shared library; raises if it is absent (build it with build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                          f"(run __graft_entry__.build() or make -C srmeetsps-cuda_amd/csrc); "
                          f"there is no fallback path")
    # torch first: it bundles its own libamdhip64; loading it before ours makes both share ONE HIP
    # runtime (device pointers and streams are exchanged between them)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    p, i, f, vp = C.c_void_p, C.c_int, C.c_float, C.c_void_p
    ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
    sig = {
        "srps_last_error": (C.c_char_p, []),
        "srps_version": (C.c_char_p, []),
        "srps_create": (i, [i, i, i, C.POINTER(vp)]),
        "srps_destroy": (i, [vp]),
        "srps_set_stream": (i, [vp, vp]),
        "srps_synchronize": (i, [vp]),
        "srps_set_option": (i, [vp, C.c_char_p, i]),
        "srps_get_option": (i, [vp, C.c_char_p, C.POINTER(i)]),
        "srps_host_COO_to_device_CSR": (i, [vp, ip, ip, fp, i, i, i, p, p, p]),
        "srps_sparsemat_densevec_mul": (i, [vp, p, p, p, i, i, i, p, i, p]),
        "srps_conjugate_gradient": (i, [vp, p, p, p, i, i, p, p, ip]),
        "srps_mean_across_channels": (i, [vp, fp, i, i, i, p, p]),
        "srps_rho_init": (i, [vp, p, i, i]),
        "srps_meshgrid_create": (i, [vp, i, i, f, f, p, p]),
        "srps_normal_init": (i, [vp, p, p, p, p, p, i, f, f, p, p]),
        "srps_lightning_estimation": (i, [vp, p, p, p, p, i, i, i]),
        "srps_albedo_estimation": (i, [vp, p, p, p, p, i, i, i]),
        "srps_bind_grid": (i, [vp, i, i, i, fp]),
        "srps_depth_estimation": (i, [vp, p, p, p, p, p, p, p, p, p, f, f, i, i, i, fp]),
        "srps_depth_estimation_csr": (i, [vp, p, p, p, p, p, p, p, p, p, p, i, i, i, p, p, p, i, i, i, p, p, p, i, i, i, p, p, f, f, i, i, i, fp]),
        "srps_gradient": (i, [vp, p, i, p, p]),
        "srps_set_principal_point": (i, [vp, f, f]),
        "srps_depth_operator_apply": (i, [vp, p, i, p]),
        "srps_setup": (i, [vp, C.POINTER(Problem)]),
        "srps_upload_image": (i, [vp, i, fp]),
        "srps_upload_image_u8": (i, [vp, i, C.POINTER(C.c_ubyte)]),
        "srps_dims": (i, [vp, ip, ip, ip, ip, ip, ip]),
        "srps_lighting": (i, [vp]),
        "srps_albedo": (i, [vp]),
        "srps_depth": (i, [vp, fp]),
        "srps_normals": (i, [vp]),
        "srps_lighting_local": (i, [vp]),
        "srps_albedo_partial": (i, [vp]),
        "srps_albedo_finish": (i, [vp]),
        "srps_depth_partial": (i, [vp]),
        "srps_depth_solve": (i, [vp]),
        "srps_energy_partial": (i, [vp]),
        "srps_energy_finish": (i, [vp, fp]),
        "srps_exchange": (i, [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "srps_execute": (i, [vp, i, fp, ip]),
        "srps_execute_sharded": (i, [vp, i, fp, ip]),
        "srps_comm_unique_id": (i, [p]),
        "srps_comm_init_rank": (i, [vp, p, i, i]),
        "srps_comm_init_all": (i, [C.POINTER(vp), i]),
        "srps_set_comm": (i, [vp, vp, i, i]),
        "srps_comm_release": (i, [vp]),
        "srps_comm_info": (i, [vp, ip, ip]),
        "srps_all_reduce": (i, [vp, C.c_char_p]),
        "srps_set_host_collectives": (i, [vp, i, i, HOST_ALLREDUCE_FN, HOST_BROADCAST_FN, vp]),
        "srps_strip_group_solve": (i, [C.POINTER(vp), i]),
        "srps_strip_group_solve_resident": (i, [C.POINTER(vp), i]),
        "srps_strip_range": (i, [i, i, i, i, ip, ip]),
        "srps_shard_range": (i, [i, i, i, ip, ip]),
        "srps_device_count": (i, [ip]),
        "srps_transfer_buffers": (i, [ip]),
        "srps_set_strip_transport": (i, [vp, i, i, STRIP_ALLREDUCE_FN, STRIP_EXCHANGE_FN, STRIP_ALLGATHER_FN, vp]),
        "srps_get": (i, [vp, C.c_char_p, fp, C.c_size_t]),
        "srps_set": (i, [vp, C.c_char_p, fp, C.c_size_t]),
        "srps_get_device_ptr": (i, [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "srps_array_size": (i, [vp, C.c_char_p, C.POINTER(C.c_size_t)]),
        "srps_last_cg_iterations": (i, [vp, ip, ip, ip]),
        "srps_bench_cg": (i, [vp, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "srps_cg_bytes": (i, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "srps_get_timings": (i, [vp, fp]),
        "srps_phase_name": (C.c_char_p, [i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    lib._signatures = sig
    _lib = lib
    return lib


def check(rc: int):
    if rc != SRPS_OK:
        raise SRPSError(rc, load().srps_last_error().decode(errors="replace"))
