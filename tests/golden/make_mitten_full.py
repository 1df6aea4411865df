#!/usr/bin/env python3
"""Generates tests/golden/mitten_full.npz from the reference's bundled image dataset
(/root/reference/dataset/Images/Mitten: 20 RGB 1280x960 PNG, 20 16-bit 640x480 depth PNG, mask, K.txt;
BASELINE.json config 2 "mitten_sf2 (sf=2, ~8 images) ... full alternating solve to convergence"): the WHOLE frame, the
first 8 images in cv::glob order, all depth frames.  Loaded with this repo's own C++ loaders (ImageDataHandler) and
pre-processed with its C++ depth pre-processing; only the masked samples are stored (bytes for the images, as the
loader produced them from the PNGs: value = byte / 255.f), plus the oracle's (faithful restatement) outputs.
The fixture is data only.  Re-run:  python tests/golden/make_mitten_full.py   (needs /root/reference)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import srps_oracle as O  # noqa: E402

pkg = importlib.import_module("srmeetsps-cuda_amd")
SRC = "/root/reference/dataset/Images/Mitten"
N = 8


def main():
    pkg.host.build()
    dh = pkg.host.load_dataset("images", SRC, preprocess=True)
    h, w, sf = dh.I_h, dh.I_w, dh.sf
    mask = dh.mask.astype(np.float32)                                     # column-major h*w
    I = dh.I.reshape(dh.I_n, 3, h * w)[:N]
    idx = np.flatnonzero(mask == 1)
    I_u8 = np.rint(I[:, :, idx] * 255).astype(np.uint8)
    assert np.array_equal(I_u8.astype(np.float32) / np.float32(255), I[:, :, idx])      # the loader's floats are bytes / 255.f
    I_f = np.zeros((N, 3, h * w), np.float32); I_f[:, :, idx] = I_u8.astype(np.float32) / np.float32(255)
    prob = O.Problem(h, w, sf, mask, dh.K.copy(), I_f, dh.zs_lr.copy(), dh.z_full.copy())
    geo = O.build_geometry(h, w, sf, mask)
    ref = O.execute(prob, depth="faithful")
    print("P", ref.geo.npix, "Ps", ref.geo.npixs, "outer", ref.iterations, "energies", ref.energies)
    out = dict(h=h, w=w, sf=sf, mask_bits=np.packbits(mask.astype(np.uint8)), K=dh.K.copy(), I_u8=I_u8,
               imasks=geo.imasks.astype(np.int32),            # the low-resolution pixels whose sf x sf block lies in the mask (Utilities.cpp:201-220)
               zs_lr_masked=dh.zs_lr[geo.imasks].astype(np.float32), z_full_masked=dh.z_full[geo.imask].astype(np.float32),
               final_z=ref.z, final_rho=ref.rho, final_s=ref.s, energies=np.array(ref.energies), n_outer=ref.iterations,
               npix=ref.geo.npix, npixs=ref.geo.npixs)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mitten_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
