"""ctypes binding of libsrps_host.so: the CPU side of the C++ host (srmeetsps-cuda_amd/host/) --
the MAT5 / PNG loaders behind DataHandler (reference Utilities.cpp:159-199, 349-395) and the depth
pre-processing of SRPS.cu:117-149 -- so that Python drives exactly the code the `srps` program runs."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .api import DataHandler

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(_HERE, "libsrps_host.so")
CLI = os.path.join(_HERE, "srps")
f32 = np.float32
_fp = C.POINTER(C.c_float)
_lib = None


def build():
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "host"), "-j4"])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB):
            raise ImportError(f"{HOST_LIB} is missing (make -C srmeetsps-cuda_amd/host)")
        _lib = C.CDLL(HOST_LIB)
        _lib.srps_host_last_error.restype = C.c_char_p
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(load().srps_host_last_error().decode(errors="replace"))


def _f(a):
    return a.ctypes.data_as(_fp)


def preprocess_depth(z0, z0_h, z0_w, z0_n, I_h, I_w):
    """(zs, z_full): smoothed LR depth and up-sampled HR depth, flat column-major (SRPS.cu:117-149)"""
    z0 = np.ascontiguousarray(z0, dtype=f32).reshape(-1)
    zs = np.empty(z0_h * z0_w, f32); zf = np.empty(I_h * I_w, f32)
    _check(load().srps_host_preprocess_depth(_f(z0), z0_h, z0_w, z0_n, I_h, I_w, _f(zs), _f(zf)))
    return zs, zf


def inpaint(img2d, flag2d, radius=16):
    a = np.ascontiguousarray(img2d, dtype=f32).copy(); fl = np.ascontiguousarray(flag2d, dtype=np.uint8)
    _check(load().srps_host_inpaint(_f(a), fl.ctypes.data_as(C.POINTER(C.c_ubyte)), a.shape[0], a.shape[1], radius))
    return a


def bilateral(img2d, sigma_color=2.0, sigma_space=2.0):
    a = np.ascontiguousarray(img2d, dtype=f32); out = np.empty_like(a)
    _check(load().srps_host_bilateral(_f(a), _f(out), a.shape[0], a.shape[1], C.c_float(sigma_color), C.c_float(sigma_space)))
    return out


def resize_cubic(img2d, out_rows, out_cols):
    a = np.ascontiguousarray(img2d, dtype=f32); out = np.empty((out_rows, out_cols), f32)
    _check(load().srps_host_resize_cubic(_f(a), a.shape[0], a.shape[1], _f(out), out_rows, out_cols))
    return out


def load_dataset(dstype: str, dsloc: str, preprocess: bool = True) -> DataHandler:
    """MatFileDataHandler::loadDataFromMatFiles / ImageDataHandler::loadDataFromImages (C++), as a
    Python DataHandler; with preprocess=True also runs the depth pre-processing."""
    lib = load()
    h = C.c_void_p()
    _check(lib.srps_host_load(dstype.encode(), dsloc.encode(), C.byref(h)))
    try:
        v = [C.c_int(0) for _ in range(5)]; sf = C.c_float(0)
        lib.srps_host_data_dims(h, *[C.byref(x) for x in v], C.byref(sf))
        I_h, I_w, I_c, I_n, z0_n = [x.value for x in v]
        isf = int(sf.value)
        I = np.empty((I_n, I_c, I_h * I_w), f32); mask = np.empty(I_h * I_w, f32); K = np.empty(9, f32)
        z0 = np.empty((z0_n, (I_h // isf) * (I_w // isf)), f32)
        lib.srps_host_data_copy(h, _f(I), _f(mask), _f(K), _f(z0))
    finally:
        lib.srps_host_data_free(h)
    dh = DataHandler(I=I, mask=mask, K=K, sf=isf, z0=z0, I_h=I_h, I_w=I_w, I_c=I_c, I_n=I_n, I_n_total=I_n, z0_n=z0_n)
    if preprocess:
        dh.zs_lr, dh.z_full = preprocess_depth(z0, dh.z0_h, dh.z0_w, z0_n, I_h, I_w)
    return dh


def write_view(kind: str, data, imask, rows: int, cols: int, path: str, nchannels: int = 3, scale: float = 1.0):
    """PNG of one of the reference's views (Utilities.cpp:242-320): kind in {"normals", "albedo", "depth"}"""
    k = {"normals": 0, "albedo": 1, "depth": 2}[kind]
    d = np.ascontiguousarray(data, dtype=f32).reshape(-1); im = np.ascontiguousarray(imask, dtype=np.int32)
    _check(load().srps_host_write_view(k, _f(d), im.ctypes.data_as(C.POINTER(C.c_int)), int(im.size), rows, cols, nchannels,
                                       C.c_float(scale), path.encode()))
