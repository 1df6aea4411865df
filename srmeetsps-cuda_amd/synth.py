"""Synthetic SRPS scenes -- inputs for the parity tests and for bench.py.

This is input generation, not part of the solver: it renders ``I = rho*(s.[n(z*);1]) + noise``
from a smooth ground-truth depth with the same masked forward/backward differences the solver
uses (reference: make_gradient, SRPS.cu:23-71; normals dc.cu:171-192), so the ground truth is
an exact fixed point of the discretised model up to the noise.

All image-like arrays are flat column-major vectors (linear index i + j*h), float32, exactly
as DataHandler holds them (Util.h:166-181): I[n][c][h*w], mask[h*w] in {0,1}, K 3x3
column-major (K[0]=fx, K[4]=fy, K[6]=cx, K[7]=cy), z0 on the (h/sf)x(w/sf) grid.
"""
from __future__ import annotations

from dataclasses import dataclass
import numpy as np

f32 = np.float32


@dataclass
class Scene:
    h: int
    w: int
    sf: int
    n_img: int               # images held by THIS shard
    n_img_total: int
    img_offset: int          # index of the shard's first image in the whole set
    n_ch: int
    mask: np.ndarray         # [h*w] float32 {0,1}
    K: np.ndarray            # [9] float32 column-major
    I: np.ndarray            # [n_img][n_ch][h*w] float32
    z0: np.ndarray           # [(h/sf)*(w/sf)] float32 low-resolution depth (z0_n = 1)
    zs_lr: np.ndarray        # smoothed LR depth (stand-in for SRPS.cu:129-141)
    z_init: np.ndarray       # [h*w] up-sampled initial HR depth (stand-in for SRPS.cu:147-149)
    z_true: np.ndarray       # [h*w]
    rho_true: np.ndarray     # [n_ch][h*w]
    s_true: np.ndarray       # [n_img_total][n_ch][4]


def to_cm(a2d):
    return np.ascontiguousarray(np.asarray(a2d).T).reshape(-1)


def from_cm(v, h, w):
    return np.asarray(v).reshape(w, h).T


def masked_gradients(z2d: np.ndarray, m2d: np.ndarray):
    """Forward difference with backward fallback on a mask; zero where isolated.
    Returns (zx, zy): zx differences along columns j, zy along rows i."""
    m = m2d.astype(bool)
    z = z2d
    zx = np.zeros_like(z); zy = np.zeros_like(z)
    right = np.zeros_like(m); right[:, :-1] = m[:, :-1] & m[:, 1:]
    left = np.zeros_like(m); left[:, 1:] = m[:, 1:] & m[:, :-1]; left &= ~right
    bottom = np.zeros_like(m); bottom[:-1, :] = m[:-1, :] & m[1:, :]
    top = np.zeros_like(m); top[1:, :] = m[1:, :] & m[:-1, :]; top &= ~bottom
    d = z[:, 1:] - z[:, :-1]
    zx[:, :-1][right[:, :-1]] = d[right[:, :-1]]
    zx[:, 1:][left[:, 1:]] = d[left[:, 1:]]
    d = z[1:, :] - z[:-1, :]
    zy[:-1, :][bottom[:-1, :]] = d[bottom[:-1, :]]
    zy[1:, :][top[1:, :]] = d[top[1:, :]]
    return zx, zy


def make_mask(h: int, w: int, sf: int, kind: str) -> np.ndarray:
    if kind == "full":
        return np.ones((h, w), dtype=f32)
    if kind == "ellipse":
        # centred ellipse, semi-axes 0.45h x 0.45w, snapped to whole sf x sf blocks
        hs, ws = h // sf, w // sf
        ii, jj = np.meshgrid(np.arange(hs) + 0.5, np.arange(ws) + 0.5, indexing="ij")
        inside = ((ii - hs / 2) / (0.45 * hs)) ** 2 + ((jj - ws / 2) / (0.45 * ws)) ** 2 <= 1.0
        return np.kron(inside, np.ones((sf, sf))).astype(f32)
    if kind == "ragged":
        # ellipse with un-snapped edge plus a few holes, an isolated pixel and a 1-wide strip:
        # exercises backward / empty gradient rows and partially masked LR blocks
        ii, jj = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
        m = ((ii - h / 2) / (0.43 * h)) ** 2 + ((jj - w / 2) / (0.46 * w)) ** 2 <= 1.0
        m[h // 3: h // 3 + 3, w // 3: w // 3 + 5] = False
        m[h // 2, :] |= True                       # 1-pixel-wide horizontal strip to the borders
        m[1, 1] = True; m[0, 1] = False; m[2, 1] = False; m[1, 0] = False; m[1, 2] = False  # isolated
        return m.astype(f32)
    raise ValueError(kind)


def make_scene(h: int, w: int, sf: int, n_img: int, seed: int = 1234, n_ch: int = 3,
               mask_kind: str = "full", img_begin: int = 0, img_end: int | None = None,
               noise_I: float = 0.01, noise_z: float = 0.002) -> Scene:
    """Render a scene. ``img_begin:img_end`` selects the shard of images to render (everything
    else -- depth, albedo, all lighting vectors -- is identical on every shard)."""
    assert h % sf == 0 and w % sf == 0
    if img_end is None:
        img_end = n_img
    rng = np.random.default_rng([seed, 0])
    m2d = make_mask(h, w, sf, mask_kind)
    K = np.zeros(9, dtype=f32)
    # focal length 1.2 x the longer side (field of view ~45 deg): a narrow, tall grid must not get a
    # focal length far below its height (that makes the photometric system degenerate)
    K[0] = 1.2 * max(h, w); K[4] = 1.2 * max(h, w); K[6] = (w - 1) / 2.0; K[7] = (h - 1) / 2.0; K[8] = 1
    iv = np.arange(h, dtype=np.float64)[:, None]          # rows i
    jv = np.arange(w, dtype=np.float64)[None, :]          # columns j
    # ground-truth depth: 1 + 0.1 * sum of 8 Gaussian bumps (separable => outer products)
    z = np.ones((h, w))
    for _ in range(8):
        A = rng.uniform(-1, 1); mu_i = rng.uniform(0.15, 0.85) * h; mu_j = rng.uniform(0.15, 0.85) * w
        sg = rng.uniform(0.08, 0.25) * min(h, w)
        z += (0.1 * A) * (np.exp(-((iv - mu_i) ** 2) / (2 * sg * sg)) * np.exp(-((jv - mu_j) ** 2) / (2 * sg * sg)))
    # albedo: smooth checker in [0.2, 0.9], phase-shifted per channel
    rho_f = np.empty((n_ch, h * w), dtype=f32)
    for c in range(n_ch):
        ph = rng.uniform(0, 2 * np.pi, size=2); per = rng.uniform(0.15, 0.4, size=2)
        r2 = 0.55 + 0.35 * (np.sin(2 * np.pi * iv / (per[0] * h) + ph[0]) * np.sin(2 * np.pi * jv / (per[1] * w) + ph[1]))
        rho_f[c] = to_cm(r2.astype(f32))
    # lighting: direction uniform on the cap l_z < -0.5, ambient 0.2 (same for all channels)
    s_true = np.zeros((n_img, n_ch, 4), dtype=f32)
    for i in range(n_img):
        r_i = np.random.default_rng([seed, 1000 + i])
        lz = -r_i.uniform(0.5, 1.0); phi = r_i.uniform(0, 2 * np.pi); rr = np.sqrt(1 - lz * lz)
        s_true[i, :, 0] = rr * np.cos(phi); s_true[i, :, 1] = rr * np.sin(phi); s_true[i, :, 2] = lz
        s_true[i, :, 3] = 0.2
    # normals of the ground truth with the solver's discretisation
    zx, zy = masked_gradients(z, m2d)
    n0 = K[0] * zx; n1 = K[4] * zy
    n2 = -z - (jv - K[6]) * zx - (iv - K[7]) * zy
    dz = np.maximum(1e-10, np.sqrt(n0 * n0 + n1 * n1 + n2 * n2))
    n0f = to_cm((n0 / dz).astype(f32)); n1f = to_cm((n1 / dz).astype(f32)); n2f = to_cm((n2 / dz).astype(f32))
    del zx, zy, n0, n1, n2, dz
    n_loc = img_end - img_begin
    I = np.empty((n_loc, n_ch, h * w), dtype=f32)
    sh = np.empty(h * w, dtype=f32); tmp = np.empty(h * w, dtype=f32)
    for li, i in enumerate(range(img_begin, img_end)):
        r_i = np.random.default_rng([seed, 2000 + i])
        # shading (identical for all channels: s_true does not depend on c)
        np.multiply(n0f, s_true[i, 0, 0], out=sh)
        np.multiply(n1f, s_true[i, 0, 1], out=tmp); sh += tmp
        np.multiply(n2f, s_true[i, 0, 2], out=tmp); sh += tmp
        sh += s_true[i, 0, 3]
        for c in range(n_ch):
            img = I[li, c]
            np.multiply(rho_f[c], sh, out=img)
            if noise_I > 0:
                r_i.standard_normal(h * w, dtype=f32, out=tmp)
                tmp *= f32(noise_I)
                img += tmp
            np.clip(img, 0, 1, out=img)
    # low-resolution depth: block mean + noise (never zero -> nothing to inpaint)
    hs, ws = h // sf, w // sf
    z0_2d = z.reshape(hs, sf, ws, sf).mean(axis=(1, 3))
    z0_2d = z0_2d + noise_z * np.random.default_rng([seed, 1]).standard_normal((hs, ws))
    zs_2d, zi_2d = preprocess_depth(z0_2d, sf)
    return Scene(h, w, sf, n_loc, n_img, img_begin, n_ch, to_cm(m2d).astype(f32), K, I,
                 to_cm(z0_2d).astype(f32), to_cm(zs_2d).astype(f32), to_cm(zi_2d).astype(f32),
                 to_cm(z.astype(f32)), rho_f, s_true)


def preprocess_depth(z0_2d: np.ndarray, sf: int):
    """Stand-in for the reference's CPU OpenCV pre-processing (SRPS.cu:129-149: bilateral
    smoothing of the LR depth, then INTER_CUBIC resize): Gaussian sigma=1 + cubic-spline zoom.
    Returns (smoothed LR depth, up-sampled HR depth)."""
    from scipy import ndimage
    zs = ndimage.gaussian_filter(z0_2d, 1.0, mode="nearest")
    zi = ndimage.zoom(zs, sf, order=3, mode="nearest", grid_mode=True)
    return zs, zi
