#!/usr/bin/env python3
"""Generates tests/golden/mitten_full.npz from the reference's bundled image dataset
(/root/reference/dataset/Images/Mitten: 20 RGB 1280x960 PNG, 20 16-bit 640x480 depth PNG, mask, K.txt;
BASELINE.json config 2 "mitten_sf2 (sf=2, ~8 images) ... full alternating solve to convergence"): the WHOLE frame, the
first 8 images in cv::glob order, all depth frames.  Loaded with this repo's own C++ loaders (ImageDataHandler) and
pre-processed with its C++ depth pre-processing; only the masked samples are stored (bytes for the images, as the
loader produced them from the PNGs: value = byte / 255.f), plus the oracle's (faithful restatement) outputs.
Round 6: the reference's own CLI run on this folder solves ALL 20 images (`cv::glob` of RGB/, Utilities.cpp:349-352) --
tests/golden/mitten_full_20.npz holds the masked bytes of images 9 - 20 (glob order) and the oracle's outputs of the 20-image solve;
mask, K and depth maps are those of mitten_full.npz.
The fixtures are data only.  Re-run:  python tests/golden/make_mitten_full.py   (needs /root/reference)"""
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
    idx = np.flatnonzero(mask == 1)
    geo = O.build_geometry(h, w, sf, mask)
    if "--all" in sys.argv:                                               # the 20-image fixture only (mitten_full.npz stays as committed)
        n_all = dh.I_n
        assert n_all == 20
        I = dh.I.reshape(n_all, 3, h * w)
        I_u8 = np.rint(I[:, :, idx] * 255).astype(np.uint8)
        assert np.array_equal(I_u8.astype(np.float32) / np.float32(255), I[:, :, idx])
        old = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mitten_full.npz"))
        assert np.array_equal(old["I_u8"], I_u8[:N]), "mitten_full.npz holds the first 8 images of the same glob order"
        I_f = np.zeros((n_all, 3, h * w), np.float32); I_f[:, :, idx] = I_u8.astype(np.float32) / np.float32(255)
        ref = O.execute(O.Problem(h, w, sf, mask, dh.K.copy(), I_f, dh.zs_lr.copy(), dh.z_full.copy()), depth="faithful")
        print("20 images: P", ref.geo.npix, "outer", ref.iterations, "energies", ref.energies)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mitten_full_20.npz")
        np.savez_compressed(path, I_u8_9_to_20=I_u8[N:], final_z=ref.z, final_rho=ref.rho, final_s=ref.s, energies=np.array(ref.energies), n_outer=ref.iterations)
        print("wrote", path, os.path.getsize(path), "bytes")
        return
    I = dh.I.reshape(dh.I_n, 3, h * w)[:N]
    I_u8 = np.rint(I[:, :, idx] * 255).astype(np.uint8)
    assert np.array_equal(I_u8.astype(np.float32) / np.float32(255), I[:, :, idx])      # the loader's floats are bytes / 255.f
    I_f = np.zeros((N, 3, h * w), np.float32); I_f[:, :, idx] = I_u8.astype(np.float32) / np.float32(255)
    prob = O.Problem(h, w, sf, mask, dh.K.copy(), I_f, dh.zs_lr.copy(), dh.z_full.copy())
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
