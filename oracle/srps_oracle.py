"""
srps_oracle.py -- CPU restatement of the SRPS alternating-optimisation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
``srmeetsps-cuda_amd``, the C-ABI library, ``bench.py``'s timed region) may import,
call or link this file.  Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker.

PARITY UNPINNED.  The reference (nihalsid/SRmeetsPS-CUDA) ships no tests, no golden
vectors and no recorded outputs; it cannot be compiled here (nvcc / cuBLAS / legacy
cuSPARSE / Thrust absent, OpenCV shared object and the .mat inputs are missing blobs)
and it is not Python, so it cannot be imported either.  This file therefore restates
the reference's algorithm line by line from its sources and is pinned only by
(i) hand-derived operator-level known answers (tests/test_oracle_operators.py) and
(ii) the agreement of two independent restatements in this file -- the *faithful*
one (assembled sparse matrices + the reference's generic CG, fp32, follows
devicecalls.cu statement by statement) and the *matrix-free* one (what the HIP kernels
compute).

Citations: ``SRPS.cu``, ``dc.cu`` (= devicecalls.cu), ``dc.cuh``, ``Util.cpp``,
``Util.h`` all live in /root/reference/SRmeetsPS-GPU/.

Conventions (identical to the reference):
  * every image-like array is a flat vector in column-major (MATLAB) order,
    linear HR index = i + j*h  (i = row, j = column);
  * masked ("compact") vectors list the masked pixels in ascending linear index;
  * I[n][c][p], s[n][c][4], rho[c][p], N[k][p] (k = 0..3), all float32.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np
import scipy.sparse as sp

f32 = np.float32

# Hard-coded constants of the reference
CG_TOL = f32(1e-9)          # dc.cu:230
CG_MAX_ITER = 100           # dc.cu:231  (loop runs while k <= max_iter  => up to 101 steps)
OUTER_TOLERANCE = f32(5e-3)  # SRPS.cu:85
OUTER_MAX_ITERATIONS = 10   # SRPS.cu:86
LAMBDA = f32(1.0)           # dc.cu:644


# --------------------------------------------------------------------------------------
# helpers: column-major flat <-> 2-D
# --------------------------------------------------------------------------------------
def to_cm(img2d: np.ndarray) -> np.ndarray:
    """2-D (h, w) array -> flat column-major vector (index i + j*h)."""
    return np.ascontiguousarray(np.asarray(img2d).T).reshape(-1)


def from_cm(vec: np.ndarray, h: int, w: int) -> np.ndarray:
    """flat column-major vector -> 2-D (h, w) array."""
    return np.asarray(vec).reshape(w, h).T


# --------------------------------------------------------------------------------------
# a3: down-sampling matrix D and its masked restriction KT
# --------------------------------------------------------------------------------------
def downsampling_coo(h: int, w: int, sf: int):
    """DataHandler::initializeDownsamplingMatrix, Util.cpp:201-220.

    Returns (row, col, val, n_row, n_col) of the box filter D: every LR pixel i averages
    an sf x sf HR block; LR pixels are numbered column-major on the (h/sf) x (w/sf) grid.
    Column formula is Util.cpp:216 verbatim (in integer arithmetic).
    """
    sf = int(sf)
    n_row = int(h * w / (sf * sf))
    n_col = h * w
    per = sf * sf
    hs = int(h / sf)
    i = np.arange(n_row, dtype=np.int64)
    base = (i // hs) * h * sf + (i % hs) * sf            # Util.cpp:216, first two terms
    j = np.arange(sf, dtype=np.int64)
    k = np.arange(sf, dtype=np.int64)
    col = base[:, None, None] + (j * h)[None, :, None] + k[None, None, :]
    row = np.repeat(i, per)
    val = np.full(n_row * per, f32(1.0) / f32(sf * sf), dtype=f32)   # Util.cpp:212
    return row.astype(np.int32), col.reshape(-1).astype(np.int32), val, n_row, n_col


def lr_mask(h: int, w: int, sf: int, mask: np.ndarray) -> np.ndarray:
    """SRPS.cu:105-111: masks = D*mask, entries < 1 replaced by 0 (fp32 sequential row sums)."""
    row, col, val, n_row, _ = downsampling_coo(h, w, sf)
    per = sf * sf
    prod = (val * mask.astype(f32)[col]).reshape(n_row, per)
    acc = np.zeros(n_row, dtype=f32)
    for t in range(per):                                   # sequential fp32 accumulation
        acc = (acc + prod[:, t]).astype(f32)
    acc[acc < f32(1.0)] = f32(0.0)                         # is_less_than_one, dc.cuh:13-17
    return acc


@dataclass
class Geometry:
    """Index sets and masked operators built by SRPS::execute, SRPS.cu:151-203."""
    h: int
    w: int
    sf: int
    imask: np.ndarray                 # masked HR linear indices, ascending (SRPS.cu:157-162)
    imasks: np.ndarray                # masked LR linear indices (SRPS.cu:163-166)
    index_in_masked: np.ndarray       # h*w -> compact index (0 where unmasked!)  SRPS.cu:154-161
    KT: sp.csr_matrix                 # Ps x P, values 1/sf^2            (SRPS.cu:170-193)
    Dx: sp.csr_matrix                 # P x P, difference along columns j (SRPS.cu:23-71)
    Dy: sp.csr_matrix                 # P x P, difference along rows i
    masks: np.ndarray                 # LR mask (0/1 floats)

    @property
    def npix(self) -> int:
        return int(self.imask.size)

    @property
    def npixs(self) -> int:
        return int(self.imasks.size)


def make_gradient(mask: np.ndarray, h: int, w: int, index_in_masked: np.ndarray, npix: int):
    """make_gradient + set_sparse_matrix_for_gradient, SRPS.cu:10-71 (same double loop).

    Dx: difference along j (columns): forward (x[i,j+1]-x[i,j]) when the right neighbour is
    masked, else backward (x[i,j]-x[i,j-1]) when the left one is, else an empty row.
    Dy: same along i (rows): "bottom" = i+1 forward, "top" = i-1 backward.
    """
    m = from_cm(mask, h, w) != 0
    idx = from_cm(index_in_masked, h, w)
    # neighbour availability (SRPS.cu:31-46)
    bottom = np.zeros_like(m); bottom[:-1, :] = m[:-1, :] & m[1:, :]
    top = np.zeros_like(m);    top[1:, :] = m[1:, :] & m[:-1, :]
    top &= ~bottom                                       # "else if"
    right = np.zeros_like(m);  right[:, :-1] = m[:, :-1] & m[:, 1:]
    left = np.zeros_like(m);   left[:, 1:] = m[:, 1:] & m[:, :-1]
    left &= ~right

    def pairs(sel, di, dj):
        ii, jj = np.nonzero(sel)
        return idx[ii, jj], idx[ii + di, jj + dj]

    def coo(ic, ir, k1, k2):
        # set_sparse_matrix_for_gradient: rows [ic, ic], cols [ir, ic], vals [k1.., k2..]
        r = np.concatenate([ic, ic]); c = np.concatenate([ir, ic])
        v = np.concatenate([np.full(ic.size, k1, f32), np.full(ic.size, k2, f32)])
        return r, c, v

    ic_r, ir_r = pairs(right, 0, 1)
    ic_l, ir_l = pairs(left, 0, -1)
    ic_b, ir_b = pairs(bottom, 1, 0)
    ic_t, ir_t = pairs(top, -1, 0)
    rx = [coo(ic_r, ir_r, 1, -1), coo(ic_l, ir_l, -1, 1)]     # Dxp, Dxn  SRPS.cu:50-54
    ry = [coo(ic_b, ir_b, 1, -1), coo(ic_t, ir_t, -1, 1)]     # Dyp, Dyn  SRPS.cu:56-60

    def build(parts):
        r = np.concatenate([p[0] for p in parts]); c = np.concatenate([p[1] for p in parts])
        v = np.concatenate([p[2] for p in parts])
        return sp.csr_matrix((v.astype(f32), (r, c)), shape=(npix, npix), dtype=f32)

    return build(rx), build(ry)


def build_geometry(h: int, w: int, sf: int, mask: np.ndarray) -> Geometry:
    """SRPS.cu:100-203 (everything that depends only on h, w, sf, mask)."""
    mask = np.asarray(mask, dtype=f32).reshape(-1)
    assert mask.size == h * w
    masks = lr_mask(h, w, sf, mask)
    imask = np.nonzero(mask != 0)[0].astype(np.int32)             # SRPS.cu:157-162
    index_in_masked = np.zeros(h * w, dtype=np.int32)
    index_in_masked[imask] = np.arange(imask.size, dtype=np.int32)
    imasks = np.nonzero(masks != 0)[0].astype(np.int32)           # SRPS.cu:163-166
    # KT: entries of D whose row is in imasks and column in imask (SRPS.cu:176-189)
    row, col, _, n_row, n_col = downsampling_coo(h, w, sf)
    in_s = np.zeros(n_row, dtype=bool); in_s[imasks] = True
    in_m = mask != 0
    keep = in_s[row] & in_m[col]
    pos_s = np.zeros(n_row, dtype=np.int32); pos_s[imasks] = np.arange(imasks.size, dtype=np.int32)
    kt_r = pos_s[row[keep]]; kt_c = index_in_masked[col[keep]]
    kt_v = np.full(kt_r.size, f32(1.0) / f32(sf * sf), dtype=f32)  # SRPS.cu:188
    KT = sp.csr_matrix((kt_v, (kt_r, kt_c)), shape=(imasks.size, imask.size), dtype=f32)
    Dx, Dy = make_gradient(mask, h, w, index_in_masked, imask.size)
    return Geometry(h, w, int(sf), imask, imasks, index_in_masked, KT, Dx, Dy, masks)


# --------------------------------------------------------------------------------------
# a11: init kernels
# --------------------------------------------------------------------------------------
def mean_across_channels(z0: np.ndarray, h: int, w: int, nc: int):
    """mean_across_channels, dc.cu:95-110. z0 is [nc][h*w] flat col-major.
    Sum of the non-zero samples divided by nc (NOT by the valid count); a zero in any
    channel flags the pixel for inpainting."""
    z0 = np.asarray(z0, dtype=f32).reshape(nc, h * w)
    avg = np.zeros(h * w, dtype=f32)
    flag = np.zeros(h * w, dtype=np.uint8)
    for c in range(nc):
        nz = z0[c] != 0
        avg = np.where(nz, (avg + z0[c]).astype(f32), avg)
        flag[~nz] = 1
    return (avg / f32(nc)).astype(f32), flag


def meshgrid_masked(geo: Geometry, K: np.ndarray):
    """meshgrid_create (dc.cu:151-158) + copy_if (SRPS.cu:253-258): xx = j - K[6], yy = i - K[7].
    (The reference kernel's landscape bug -- SURVEY 2a -- is not reproduced: every masked
    pixel gets its coordinate.)"""
    K = np.asarray(K, dtype=f32).reshape(-1)
    i = (geo.imask % geo.h).astype(f32)
    j = (geo.imask // geo.h).astype(f32)
    return (j - K[6]).astype(f32), (i - K[7]).astype(f32)


def normal_init(z, zx, zy, xx, yy, fx, fy):
    """cuda_based_normal_init + 3 kernels, dc.cu:171-223. Returns N[4][P], dz[P]."""
    z, zx, zy, xx, yy = (np.asarray(a, dtype=f32) for a in (z, zx, zy, xx, yy))
    fx = f32(fx); fy = f32(fy)
    P = z.size
    N = np.zeros((4, P), dtype=f32)
    N[0] = fx * zx                                    # saxpy into zeros, dc.cu:204
    N[1] = fy * zy                                    # dc.cu:211
    N[2] = -z - xx * zx - yy * zy                     # dc.cu:174
    N[3] = 1                                          # dc.cu:175
    dz = np.maximum(f32(1e-10), np.sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2])).astype(f32)  # dc.cu:182
    N[:3] /= dz                                       # dc.cu:190
    return N, dz


# --------------------------------------------------------------------------------------
# a7: the reference's conjugate gradient, dc.cu:229-279
# --------------------------------------------------------------------------------------
def conjugate_gradient(matvec, x: np.ndarray, b: np.ndarray, tol=CG_TOL, max_iter=CG_MAX_ITER,
                       trace: list | None = None, dtype=f32):
    """x: warm start (updated in place), b: residual rhs - A x0 (destroyed), fp32.
    Returns number of iterations executed. Exactly the recurrence of dc.cu:251-275."""
    T = dtype
    r1 = T(np.dot(b, b))
    r0 = T(0)
    k = 0
    p = np.zeros_like(b)
    while r1 > T(tol) * T(tol) and k <= max_iter:
        k += 1
        if k == 1:
            p[:] = b                                   # Scopy
        else:
            beta = T(r1 / r0)
            p *= beta                                  # Sscal
            p += b                                     # Saxpy(1)
        omega = matvec(p).astype(T, copy=False)
        dot = T(np.dot(p, omega))
        alpha = T(r1 / dot)
        x += alpha * p                                 # Saxpy
        b -= alpha * omega                             # Saxpy(-alpha)
        r0 = r1
        r1 = T(np.dot(b, b))
        if trace is not None:
            trace.append((k, float(r1), float(alpha)))
    return k


# --------------------------------------------------------------------------------------
# a8: lighting, dc.cu:376-444
# --------------------------------------------------------------------------------------
def lighting_estimation(s, rho, N, I, cg_iters: list | None = None):
    """In-place update of s[n][c][4]. For every (image i, channel j): 4x4 normal equations of
    A_j = rho_j (.) [N0..N3] (dc.cu:381) against b = I[i][j], warm-started CG (dc.cu:422-437)."""
    n_img, n_ch, P = I.shape
    for j in range(n_ch):
        A = (rho[j][None, :] * N).astype(f32)          # [4][P], dc.cu:381
        ATA = (A @ A.T).astype(f32)                    # sgemm dc.cu:422
        for i in range(n_img):
            ATb = (A @ I[i, j]).astype(f32)            # sgemv dc.cu:423
            ATb = (ATb - ATA @ s[i, j]).astype(f32)    # sgemv dc.cu:424
            x = s[i, j].copy()
            it = conjugate_gradient(lambda v: (ATA @ v).astype(f32), x, ATb)
            s[i, j] = x
            if cg_iters is not None:
                cg_iters.append(it)
    return s


# --------------------------------------------------------------------------------------
# a9: albedo, dc.cu:447-548
# --------------------------------------------------------------------------------------
def albedo_estimation(s, rho, N, I, cg_iters: list | None = None):
    """In-place update of rho[c][P]: per channel a (P*n_img) x P block-diagonal-stacked system
    expanded to sparse (dc.cu:447-495), normal equations by SpGEMM (dc.cu:395-406), global CG."""
    n_img, n_ch, P = I.shape
    for c in range(n_ch):
        s_c = s[:, c, :]                                         # d_s_buff [n_img][4], dc.cu:502-504
        A = (N.T @ s_c.T).astype(f32)                            # P x n_img, sgemm dc.cu:507
        a_flat = A.T.reshape(-1)                                 # column-major storage: index i*P + p
        rows = np.arange(P * n_img, dtype=np.int64)
        cols = rows % P                                          # fill_A_expansion dc.cu:447-454
        Asp = sp.csr_matrix((a_flat, (rows, cols)), shape=(P * n_img, P), dtype=f32)
        b = I[:, c, :].reshape(-1).astype(f32)                   # dc.cu:523-527
        ATA = (Asp.T @ Asp).astype(f32).tocsr()                  # csrgemm dc.cu:398-401
        ATb = (Asp.T @ b).astype(f32)                            # csrmv dc.cu:404
        ATb = (ATb - ATA @ rho[c]).astype(f32)                   # csrmv dc.cu:405
        x = rho[c].copy()
        it = conjugate_gradient(lambda v: (ATA @ v).astype(f32), x, ATb)
        rho[c] = x
        if cg_iters is not None:
            cg_iters.append(it)
    return rho


def albedo_closed_form(s, rho, N, I):
    """The fixed point the reference's diagonal CG converges to: rho = sum_i sh*I / sum_i sh^2,
    sh = N^T s_ic; pixels with zero denominator keep their value."""
    n_img, n_ch, P = I.shape
    out = rho.copy()
    for c in range(n_ch):
        sh = (s[:, c, :].astype(np.float64) @ N.astype(np.float64))      # [n_img][P]
        num = (sh * I[:, c, :]).sum(0); den = (sh * sh).sum(0)
        ok = den > 0
        out[c, ok] = (num[ok] / den[ok]).astype(f32)
    return out


# --------------------------------------------------------------------------------------
# a10: depth, dc.cu:550-786 -- faithful assembled version
# --------------------------------------------------------------------------------------
def depth_coefficients(s, rho, dz, xx, yy, fx, fy, I, dtype=f32):
    """A_ch1, A_ch2, A_ch3 (dc.cu:583-599, launches 616-618) and B (dc.cu:550-581).
    Returned as [c][i][P] arrays."""
    T = dtype
    n_img, n_ch, P = I.shape
    s = s.astype(T); rho = rho.astype(T); dz = dz.astype(T); xx = xx.astype(T); yy = yy.astype(T)
    g = rho / dz[None, :]                                           # [c][P]
    s0 = s[:, :, 0].T; s1 = s[:, :, 1].T; s2 = s[:, :, 2].T; s3 = s[:, :, 3].T   # [c][i]
    a1 = g[:, None, :] * (T(fx) * s0[:, :, None] - xx[None, None, :] * s2[:, :, None])
    a2 = g[:, None, :] * (T(fy) * s1[:, :, None] - yy[None, None, :] * s2[:, :, None])
    a3 = g[:, None, :] * s2[:, :, None]
    B = np.transpose(I, (1, 0, 2)).astype(T) - rho[:, None, :] * s3[:, :, None]  # N3 == 1
    return a1.astype(T), a2.astype(T), a3.astype(T), B.astype(T)


def assemble_depth_system(geo: Geometry, s, rho, dz, xx, yy, fx, fy, I):
    """dc.cu:668-745: A (rows ordered [c][i][p]), A_ = KT'KT + lambda A'A, rhs = KT'z0s + lambda A'B
    is formed by the caller (needs z0s). Returns (A, A_, B_flat)."""
    n_img, n_ch, P = I.shape
    a1, a2, a3, B = depth_coefficients(s, rho, dz, xx, yy, fx, fy, I)
    blocks = []
    eye = sp.identity(P, dtype=f32, format="csr")
    for c in range(n_ch):
        for i in range(n_img):
            blk = (sp.diags(a1[c, i]).astype(f32) @ geo.Dx + sp.diags(a2[c, i]).astype(f32) @ geo.Dy
                   - sp.diags(a3[c, i]).astype(f32) @ eye)          # dc.cu:676-691
            blocks.append(blk.astype(f32))
    A = sp.vstack(blocks, format="csr", dtype=f32)                  # dc.cu:700-723
    KTTKT = (geo.KT.T @ geo.KT).astype(f32)                         # dc.cu:734
    ATA = (A.T @ A).astype(f32)                                     # dc.cu:735
    A_ = (KTTKT + LAMBDA * ATA).astype(f32).tocsr()                 # dc.cu:736
    return A, A_, B.reshape(-1).astype(f32)


def depth_estimation(geo: Geometry, s, rho, I, xx, yy, dz, z0s, z, fx, fy,
                     cg_trace: list | None = None):
    """cuda_based_depth_estimation, dc.cu:636-786. Updates z in place, returns energy."""
    A, A_, B = assemble_depth_system(geo, s, rho, dz, xx, yy, fx, fy, I)
    rhs = (geo.KT.T @ z0s).astype(f32)                              # dc.cu:743
    ATB = (A.T @ B).astype(f32)                                     # dc.cu:744
    rhs = (rhs + LAMBDA * ATB).astype(f32)                          # dc.cu:745
    rhs = (rhs - A_ @ z).astype(f32)                                # dc.cu:758
    conjugate_gradient(lambda v: (A_ @ v).astype(f32), z, rhs, trace=cg_trace)   # dc.cu:759
    t1 = np.sum(((geo.KT @ z).astype(f32) - z0s) ** 2, dtype=f32)   # dc.cu:762-766
    t2 = np.sum(((A @ z).astype(f32) - B) ** 2, dtype=f32)          # dc.cu:763-767
    return float(f32(t1) + LAMBDA * f32(t2))


# --------------------------------------------------------------------------------------
# matrix-free restatement (what the HIP kernels compute)
# --------------------------------------------------------------------------------------
def mf_tensor(s, rho, dz, xx, yy, fx, fy, I, dtype=np.float64):
    """Per-pixel photometric tensor M (6 unique entries of sum v v^T, v=(a1,a2,-a3)),
    q = sum v*b (3 entries) and sum b^2, reduced over channels and images."""
    a1, a2, a3, B = depth_coefficients(s, rho, dz, xx, yy, fx, fy, I, dtype=dtype)
    v0, v1, v2 = a1, a2, -a3
    red = lambda t: t.sum(axis=(0, 1))
    M = np.stack([red(v0 * v0), red(v0 * v1), red(v0 * v2), red(v1 * v1), red(v1 * v2), red(v2 * v2)])
    q = np.stack([red(v0 * B), red(v1 * B), red(v2 * B)])
    bb = red(B * B)
    return M.astype(dtype), q.astype(dtype), bb.astype(dtype)


def mf_apply(geo: Geometry, M, x, dtype=np.float64):
    """A_ x = Dx'u + Dy'v + w + KT'(KT x), (u,v,w) = M (Dx x, Dy x, x)."""
    x = x.astype(dtype)
    Dx = geo.Dx.astype(dtype); Dy = geo.Dy.astype(dtype); KT = geo.KT.astype(dtype)
    gx = Dx @ x; gy = Dy @ x
    u = M[0] * gx + M[1] * gy + M[2] * x
    v = M[1] * gx + M[3] * gy + M[4] * x
    w = M[2] * gx + M[4] * gy + M[5] * x
    return float(LAMBDA) * (Dx.T @ u + Dy.T @ v + w) + KT.T @ (KT @ x)


def mf_rhs(geo: Geometry, q, z0s, dtype=np.float64):
    Dx = geo.Dx.astype(dtype); Dy = geo.Dy.astype(dtype); KT = geo.KT.astype(dtype)
    return float(LAMBDA) * (Dx.T @ q[0] + Dy.T @ q[1] + q[2]) + KT.T @ z0s.astype(dtype)


def mf_depth_estimation(geo: Geometry, s, rho, I, xx, yy, dz, z0s, z, fx, fy, dtype=np.float64,
                        cg_trace: list | None = None):
    """Matrix-free depth step with the same CG recurrence; energy evaluated directly
    (sum over (c,i,p) of (a1 zx + a2 zy - a3 z - b)^2) like dc.cu:762-767."""
    M, q, _ = mf_tensor(s, rho, dz, xx, yy, fx, fy, I, dtype=dtype)
    zz = z.astype(dtype)
    rhs = mf_rhs(geo, q, z0s, dtype) - mf_apply(geo, M, zz, dtype)
    conjugate_gradient(lambda v: mf_apply(geo, M, v, dtype), zz, rhs.astype(dtype), trace=cg_trace, dtype=dtype)
    z[:] = zz.astype(z.dtype)
    e = energy(geo, s, rho, I, xx, yy, dz, z0s, zz, fx, fy, dtype=dtype)
    return e


def energy(geo: Geometry, s, rho, I, xx, yy, dz, z0s, z, fx, fy, dtype=np.float64):
    """t1 + lambda t2 of dc.cu:762-785 evaluated without assembling A."""
    a1, a2, a3, B = depth_coefficients(s, rho, dz, xx, yy, fx, fy, I, dtype=dtype)
    z = z.astype(dtype)
    gx = geo.Dx.astype(dtype) @ z; gy = geo.Dy.astype(dtype) @ z
    res = a1 * gx[None, None, :] + a2 * gy[None, None, :] - a3 * z[None, None, :] - B
    t2 = float(np.sum(res * res, dtype=np.float64))
    t1 = float(np.sum((geo.KT.astype(dtype) @ z - z0s.astype(dtype)) ** 2, dtype=np.float64))
    return t1 + float(LAMBDA) * t2


# --------------------------------------------------------------------------------------
# a1: the alternating loop, SRPS.cu:206-335
# --------------------------------------------------------------------------------------
@dataclass
class Problem:
    """Inputs after the (out-of-scope) CPU pre-processing of SRPS.cu:117-149:
    mask (h*w flat col-major, values {0,1}), K (9 floats, column-major 3x3),
    I_full [n][c][h*w], zs_lr (smoothed LR depth, (h/sf)*(w/sf) flat), z_full (h*w upsampled)."""
    h: int
    w: int
    sf: int
    mask: np.ndarray
    K: np.ndarray
    I_full: np.ndarray
    zs_lr: np.ndarray
    z_full: np.ndarray


@dataclass
class State:
    geo: Geometry
    s: np.ndarray
    rho: np.ndarray
    z: np.ndarray
    N: np.ndarray
    dz: np.ndarray
    I: np.ndarray
    z0s: np.ndarray
    xx: np.ndarray
    yy: np.ndarray
    fx: float
    fy: float
    energies: list = field(default_factory=list)
    iterations: int = 0


def setup(prob: Problem) -> State:
    """SRPS.cu:151-270 on the host: index sets, KT, gradients, compaction, initial values."""
    geo = build_geometry(prob.h, prob.w, prob.sf, prob.mask)
    I_full = np.asarray(prob.I_full, dtype=f32)
    n_img, n_ch, _ = I_full.shape
    sel1 = prob.mask.reshape(-1) == 1                             # is_one, dc.cuh:19-24
    I = np.ascontiguousarray(I_full[:, :, sel1])                  # SRPS.cu:227-232
    s = np.zeros((n_img, n_ch, 4), dtype=f32); s[:, :, 2] = -1    # SRPS.cu:209-217
    rho = np.full((n_ch, geo.npix), 0.5, dtype=f32)               # dc.cu:133-139
    z0s = np.asarray(prob.zs_lr, dtype=f32)[geo.masks == 1].copy()  # SRPS.cu:237-239
    z = np.asarray(prob.z_full, dtype=f32)[sel1].copy()           # SRPS.cu:242-246
    xx, yy = meshgrid_masked(geo, prob.K)
    K = np.asarray(prob.K, dtype=f32).reshape(-1)
    fx, fy = float(K[0]), float(K[4])                             # SRPS.cu:269
    zx = (geo.Dx @ z).astype(f32); zy = (geo.Dy @ z).astype(f32)  # SRPS.cu:264-265
    N, dz = normal_init(z, zx, zy, xx, yy, fx, fy)
    return State(geo, s, rho, z, N, dz, I, z0s, xx, yy, fx, fy)


def outer_iteration(st: State, depth="faithful"):
    """One pass of SRPS.cu:276-315 (lighting -> albedo -> depth -> normals). Returns energy."""
    lighting_estimation(st.s, st.rho, st.N, st.I)
    albedo_estimation(st.s, st.rho, st.N, st.I)
    if depth == "faithful":
        e = depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, st.z, st.fx, st.fy)
    else:
        e = mf_depth_estimation(st.geo, st.s, st.rho, st.I, st.xx, st.yy, st.dz, st.z0s, st.z, st.fx, st.fy,
                                dtype=np.float64 if depth == "mf64" else f32)
    zx = (st.geo.Dx @ st.z).astype(f32); zy = (st.geo.Dy @ st.z).astype(f32)
    st.N, st.dz = normal_init(st.z, zx, zy, st.xx, st.yy, st.fx, st.fy)
    return e


def execute(prob: Problem, depth="faithful", max_outer=None) -> State:
    """SRPS::execute loop incl. the stop rule of SRPS.cu:297-302 (NaN first pass; 'iteration >
    MAX_ITERATIONS' tested before the increment => up to 11 passes)."""
    st = setup(prob)
    last_error = float("nan")
    iteration = 1
    while True:
        error = outer_iteration(st, depth=depth)
        st.energies.append(error)
        with np.errstate(invalid="ignore", divide="ignore"):
            rel_err = abs(f32(last_error) - f32(error)) / abs(f32(error))
        stop = (error > last_error) or (rel_err < OUTER_TOLERANCE) or (iteration > OUTER_MAX_ITERATIONS)
        last_error = error
        iteration += 1
        st.iterations = iteration - 1
        if stop or (max_outer is not None and st.iterations >= max_outer):
            break
    return st


# --------------------------------------------------------------------------------------
# partial sums of the image-sharded formulation -- used by tests/_oracle_engine.py to
# exercise the host-side sharding logic under gloo; algebraically identical to the phases above
# --------------------------------------------------------------------------------------
def albedo_numden(s_loc, N, I_loc):
    """num[c][p] = sum_i sh I, den[c][p] = sum_i sh^2 over the images of one shard (the diagonal
    A'A and A'b of dc.cu:395-406 restricted to those images), fp32 accumulation in image order."""
    n_img, n_ch, P = I_loc.shape
    num = np.zeros((n_ch, P), dtype=f32); den = np.zeros((n_ch, P), dtype=f32)
    for c in range(n_ch):
        for i in range(n_img):
            sh = (s_loc[i, c] @ N).astype(f32)
            num[c] += sh * I_loc[i, c]; den[c] += sh * sh
    return num, den


def albedo_solve_numden(rho, num, den, cg_iters: list | None = None):
    """The reference's CG (dc.cu:540) on the diagonal system diag(den) rho = num, warm start rho."""
    for c in range(rho.shape[0]):
        b = (num[c] - den[c] * rho[c]).astype(f32)                 # dc.cu:404-405
        x = rho[c].copy()
        it = conjugate_gradient(lambda v: (den[c] * v).astype(f32), x, b)
        rho[c] = x
        if cg_iters is not None:
            cg_iters.append(it)
    return rho


def mf_tensor_split(s_all, s_loc, rho, dz, xx, yy, fx, fy, I_loc, dtype=f32):
    """M from the lighting of ALL images (no image data needed), q and sum b^2 from the local images."""
    n_all = s_all.shape[0]
    dummy = np.zeros((n_all, rho.shape[0], rho.shape[1]), dtype=dtype)
    a1, a2, a3, _ = depth_coefficients(s_all, rho, dz, xx, yy, fx, fy, dummy, dtype=dtype)
    red = lambda t: t.sum(axis=(0, 1))
    v0, v1, v2 = a1, a2, -a3
    M = np.stack([red(v0 * v0), red(v0 * v1), red(v0 * v2), red(v1 * v1), red(v1 * v2), red(v2 * v2)]).astype(dtype)
    if I_loc.shape[0] == 0:
        return M, np.zeros((3, rho.shape[1]), dtype=dtype)
    b1, b2, b3, B = depth_coefficients(s_loc, rho, dz, xx, yy, fx, fy, I_loc, dtype=dtype)
    q = np.stack([red(b1 * B), red(b2 * B), red(-b3 * B)]).astype(dtype)
    return M, q


def energy_split(geo: Geometry, s_loc, rho, I_loc, xx, yy, dz, z0s, z, fx, fy, dtype=np.float64):
    """(t1, t2 over the local images) of dc.cu:762-767"""
    z = z.astype(dtype)
    t1 = float(np.sum((geo.KT.astype(dtype) @ z - z0s.astype(dtype)) ** 2, dtype=np.float64))
    if I_loc.shape[0] == 0:
        return t1, 0.0
    a1, a2, a3, B = depth_coefficients(s_loc, rho, dz, xx, yy, fx, fy, I_loc, dtype=dtype)
    gx = geo.Dx.astype(dtype) @ z; gy = geo.Dy.astype(dtype) @ z
    res = a1 * gx[None, None, :] + a2 * gy[None, None, :] - a3 * z[None, None, :] - B
    return t1, float(np.sum(res * res, dtype=np.float64))
