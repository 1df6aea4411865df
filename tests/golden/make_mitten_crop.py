#!/usr/bin/env python3
"""Generates tests/golden/mitten_crop.npz from the reference's bundled image dataset
(/root/reference/dataset/Images/Mitten: 20 RGB 1280x960 PNG, 20 16-bit 640x480 depth PNG, mask, K.txt;
BASELINE.json config 2 "mitten_sf2"): a 256 x 192 HR window that straddles the mitten's edge, the first
8 images in cv::glob order, all depth frames.  Loaded with this repo's own C++ loaders
(ImageDataHandler) and pre-processed with its C++ depth pre-processing; stored as uint8 / uint16 data
plus the oracle's (faithful restatement) outputs.  The fixture is data only.
Re-run:  python tests/golden/make_mitten_crop.py   (needs /root/reference)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import srps_oracle as O  # noqa: E402

pkg = importlib.import_module("srmeetsps-cuda_amd")
SRC = "/root/reference/dataset/Images/Mitten"


def main():
    pkg.host.build()
    dh = pkg.host.load_dataset("images", SRC, preprocess=False)
    h, w, sf = dh.I_h, dh.I_w, dh.sf
    m2 = dh.mask.reshape(w, h).T
    rows = np.where(m2.any(1))[0]; cols = np.where(m2.any(0))[0]
    print("mask bbox rows", rows[0], rows[-1], "cols", cols[0], cols[-1], "P", int(m2.sum()))
    # window: 256 rows x 192 columns, top-left aligned to sf, placed over the lower-left edge of the mitten
    i0 = int((rows[0] + (rows[-1] - rows[0]) * 0.55) // sf * sf); j0 = int((cols[0] - 40) // sf * sf)
    H, W, N = 256, 192, 8
    sl = (slice(i0, i0 + H), slice(j0, j0 + W))
    mask = m2[sl]
    I = dh.I.reshape(dh.I_n, 3, w, h).transpose(0, 1, 3, 2)[:N, :, sl[0], sl[1]]                 # [n][c][H][W]
    z0 = dh.z0.reshape(dh.z0_n, w // sf, h // sf).transpose(0, 2, 1)[:, i0 // sf:(i0 + H) // sf, j0 // sf:(j0 + W) // sf]
    K = dh.K.copy(); K[6] -= j0; K[7] -= i0                                                    # principal point in window coordinates
    cm = lambda a: np.ascontiguousarray(np.swapaxes(a, -1, -2)).reshape(a.shape[:-2] + (-1,))
    I_u8 = np.rint(cm(I) * 255).astype(np.uint8)
    z0_u16 = np.rint(cm(z0) / 9870.0 * 65535).astype(np.uint16)
    # what the loaders produce from those integers
    I_f = I_u8.astype(np.float32) / np.float32(255)
    z0_f = (z0_u16.astype(np.float32) / np.float32(65535)) * np.float32(9870)
    mask_f = cm(mask).astype(np.float32)
    zs, zf = pkg.host.preprocess_depth(z0_f, H // sf, W // sf, z0_f.shape[0], H, W)
    prob = O.Problem(H, W, sf, mask_f, K, I_f, zs, zf)
    ref = O.execute(prob, depth="faithful")
    print("P", ref.geo.npix, "Ps", ref.geo.npixs, "outer", ref.iterations, "energies", ref.energies)
    out = dict(h=H, w=W, sf=sf, window=np.array([i0, j0]), mask=mask_f.astype(np.uint8), K=K, I_u8=I_u8, z0_u16=z0_u16, zs_lr=zs, z_full=zf,
               final_z=ref.z, final_rho=ref.rho, final_s=ref.s, energies=np.array(ref.energies), n_outer=ref.iterations,
               npix=ref.geo.npix, npixs=ref.geo.npixs)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mitten_crop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
