// kernels_pixel.hip -- per-pixel phases on the compact (masked, reference-layout) arrays: init kernels, normals,
// depth-tensor assembly and the photometric energy (lighting: kernels_lighting.hip, albedo: kernels_albedo.hip).
// The sweeps stream I[n][c][p] once, coalesced along p; none has a stencil.
#include "srps_internal.h"
#include "device_utils.h"

namespace srps {

// =============================================================================================
// init kernels  (reference: devicecalls.cu:95-169, SRPS.cu:223-260)
// =============================================================================================
__global__ void k_fill(float* __restrict__ d, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = v;
}
int launch_fill(hipStream_t st, float* d, size_t n, float v) {
    if (n == 0) return SRPS_OK;
    int nb = (int)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_fill, dim3(nb), dim3(256), 0, st, d, n, v);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// masked gather of one image (all channels): out[c][p] = full[c][imask[p]]
// replaces thrust::copy_if with the channel-replicated mask, SRPS.cu:227-232
__global__ void k_gather_image(const float* __restrict__ full, const int* __restrict__ imask, int P, size_t hw,
                               float* __restrict__ out) {
    const int c = blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x)
        out[(size_t)c * P + p] = full[(size_t)c * hw + imask[p]];
}
int launch_gather_image(hipStream_t st, const float* d_full, const int* d_imask, int P, int C, size_t hw, float* d_out) {
    int nb = std::min(cdiv(P, 256), 4096);
    hipLaunchKernelGGL(k_gather_image, dim3(nb, C), dim3(256), 0, st, d_full, d_imask, P, hw, d_out);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// the same for a batch of images (blockIdx.z), four pixels per thread when the rows of the output are 16-byte aligned
template <int V>
__global__ __launch_bounds__(256) void k_gather_images(const float* __restrict__ full, const int* __restrict__ imask, int P, size_t hw, int C, float* __restrict__ out) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float* src = full + ((size_t)n * C + c) * hw;
    float* dst = out + ((size_t)n * C + c) * P;
    if (V == 4) {
        const int P4 = P >> 2;
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < P4; t += gridDim.x * blockDim.x) {
            const int4 ix = reinterpret_cast<const int4*>(imask)[t];
            reinterpret_cast<float4*>(dst)[t] = make_float4(src[ix.x], src[ix.y], src[ix.z], src[ix.w]);
        }
    } else {
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) dst[p] = src[imask[p]];
    }
}
int launch_gather_images(hipStream_t st, const float* d_full, const int* d_imask, int P, int C, size_t hw, int n_img, float* d_out) {
    if (n_img <= 0) return SRPS_OK;
    const bool vec = (P % 4 == 0) && ((uintptr_t)d_out % 16 == 0) && ((uintptr_t)d_imask % 16 == 0);
    const int nb = std::max(1, std::min(cdiv(vec ? P / 4 : P, 256), 2048));
    if (vec) hipLaunchKernelGGL((k_gather_images<4>), dim3(nb, C, n_img), dim3(256), 0, st, d_full, d_imask, P, hw, C, d_out);
    else hipLaunchKernelGGL((k_gather_images<1>), dim3(nb, C, n_img), dim3(256), 0, st, d_full, d_imask, P, hw, C, d_out);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}
// images handed over as the bytes the image-folder loader read: the float is the loader's own expression byte / 255.f
// (Utilities.cpp:343), formed here; the bytes themselves go to the 8-bit image store (d_out8, may be null)
template <int V>
__global__ __launch_bounds__(256) void k_gather_images_u8(const unsigned char* __restrict__ full, const int* __restrict__ imask, int P, size_t hw, int C,
                                                          float* __restrict__ out, unsigned char* __restrict__ out8) {
    const int c = blockIdx.y, n = blockIdx.z;
    const unsigned char* src = full + ((size_t)n * C + c) * hw;
    float* dst = out + ((size_t)n * C + c) * P;
    unsigned char* dst8 = out8 ? out8 + ((size_t)n * C + c) * P : nullptr;
    if (V == 4) {
        const int P4 = P >> 2;
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < P4; t += gridDim.x * blockDim.x) {
            const int4 ix = reinterpret_cast<const int4*>(imask)[t];
            const unsigned b0 = src[ix.x], b1 = src[ix.y], b2 = src[ix.z], b3 = src[ix.w];
            reinterpret_cast<float4*>(dst)[t] = make_float4((float)b0 / 255.f, (float)b1 / 255.f, (float)b2 / 255.f, (float)b3 / 255.f);
            if (dst8) reinterpret_cast<unsigned*>(dst8)[t] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
    } else {
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
            const unsigned b = src[imask[p]];
            dst[p] = (float)b / 255.f;
            if (dst8) dst8[p] = (unsigned char)b;
        }
    }
}
int launch_gather_images_u8(hipStream_t st, const unsigned char* d_full, const int* d_imask, int P, int C, size_t hw, int n_img, float* d_out, unsigned char* d_out8) {
    if (n_img <= 0) return SRPS_OK;
    const bool vec = (P % 4 == 0) && ((uintptr_t)d_out % 16 == 0) && ((uintptr_t)d_imask % 16 == 0) && ((uintptr_t)d_out8 % 4 == 0);
    const int nb = std::max(1, std::min(cdiv(vec ? P / 4 : P, 256), 2048));
    if (vec) hipLaunchKernelGGL((k_gather_images_u8<4>), dim3(nb, C, n_img), dim3(256), 0, st, d_full, d_imask, P, hw, C, d_out, d_out8);
    else hipLaunchKernelGGL((k_gather_images_u8<1>), dim3(nb, C, n_img), dim3(256), 0, st, d_full, d_imask, P, hw, C, d_out, d_out8);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// The 8-bit image store.  d_out == null: only look whether every sample is k / 255.f for a byte k (bit for bit -- a negative
// zero is not); otherwise also write the bytes.  A block that sees the flag raised stops: float-valued images cost next to nothing.
__global__ __launch_bounds__(256) void k_pack_bytes(const float* __restrict__ I, size_t n4, unsigned char* __restrict__ out, int* __restrict__ inexact) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n4; t += (size_t)gridDim.x * blockDim.x) {
        if (__hip_atomic_load(inexact, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        const float4 v = reinterpret_cast<const float4*>(I)[t];
        const float e[4] = {v.x, v.y, v.z, v.w};
        unsigned w = 0;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float k = rintf(e[i] * 255.f);
            const unsigned kb = (k >= 0.f && k <= 255.f) ? (unsigned)k : 0u;
            const bool ok = __float_as_uint((float)kb / 255.f) == __float_as_uint(e[i]);      // the loader's expression, from the byte
            bad |= !ok;
            w |= kb << (8 * i);
        }
        if (bad) { atomicOr(inexact, 1); return; }
        if (out) reinterpret_cast<unsigned*>(out)[t] = w;
    }
}
int launch_pack_bytes(hipStream_t st, const float* d_I, size_t n, unsigned char* d_out, int* d_inexact) {
    const size_t n4 = n / 4;
    const int nb = (int)std::min<size_t>((n4 + 255) / 256, 8192);
    hipLaunchKernelGGL(k_pack_bytes, dim3(std::max(nb, 1)), dim3(256), 0, st, d_I, n4, d_out, d_inexact);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// xx = j - cx, yy = i - cy for the masked pixels (meshgrid_create + copy_if, dc.cu:151-158, SRPS.cu:253-258)
__global__ void k_meshgrid_compact(const int* __restrict__ imask, int P, int h, float cx, float cy,
                                   float* __restrict__ xx, float* __restrict__ yy) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int lin = imask[p];
        const int j = lin / h, i = lin - j * h;
        xx[p] = (float)j - cx;
        yy[p] = (float)i - cy;
    }
}
int launch_meshgrid_compact(hipStream_t st, const int* d_imask, int P, int h, float cx, float cy, float* xx, float* yy) {
    int nb = std::min(cdiv(P, 256), 4096);
    hipLaunchKernelGGL(k_meshgrid_compact, dim3(nb), dim3(256), 0, st, d_imask, P, h, cx, cy, xx, yy);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

__global__ void k_meshgrid_full(int w, int h, float K02, float K12, float* __restrict__ xx, float* __restrict__ yy) {
    const size_t n = (size_t)w * h;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(t / h), i = (int)(t - (size_t)j * h);
        xx[t] = (float)j - K02;
        yy[t] = (float)i - K12;
    }
}
int launch_meshgrid_full(hipStream_t st, int w, int h, float K02, float K12, float* xx, float* yy) {
    int nb = std::min(cdiv((long long)w * h, 256), 4096);
    hipLaunchKernelGGL(k_meshgrid_full, dim3(nb), dim3(256), 0, st, w, h, K02, K12, xx, yy);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// mean over the depth channels; zero samples are skipped but the divisor stays nc (dc.cu:95-110)
__global__ void k_mean_channels(const float* __restrict__ data, size_t hw, int nc, float* __restrict__ mean,
                                uint8_t* __restrict__ flag) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < hw; t += (size_t)gridDim.x * blockDim.x) {
        float avg = 0.f;
        uint8_t f = 0;
        for (int c = 0; c < nc; ++c) {
            const float v = data[(size_t)c * hw + t];
            if (v != 0.f) avg += v; else f = 1;
        }
        mean[t] = avg / (float)nc;
        flag[t] = f;
    }
}
int launch_mean_channels(hipStream_t st, const float* d_data, int h, int w, int nc, float* mean, uint8_t* flag) {
    const size_t hw = (size_t)h * w;
    int nb = std::min(cdiv((long long)hw, 256), 4096);
    hipLaunchKernelGGL(k_mean_channels, dim3(nb), dim3(256), 0, st, d_data, hw, nc, mean, flag);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// =============================================================================================
// normals: one fused kernel for devicecalls.cu:171-223 (2 saxpy + 3 kernels in the reference)
// =============================================================================================
__global__ void k_normals(const float* __restrict__ z, const float* __restrict__ zx, const float* __restrict__ zy,
                          const float* __restrict__ xx, const float* __restrict__ yy, int P, float fx, float fy,
                          float* __restrict__ N, float* __restrict__ dz) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        float n0, n1, n2, nrm;
        perspective_normal(fx, fy, z[p], zx[p], zy[p], xx[p], yy[p], n0, n1, n2, nrm);
        N[p] = n0;
        N[(size_t)P + p] = n1;
        N[2 * (size_t)P + p] = n2;
        N[3 * (size_t)P + p] = 1.f;                                // dc.cu:175
        dz[p] = nrm;
    }
}
int launch_normals(hipStream_t st, const float* z, const float* zx, const float* zy, const float* xx,
                   const float* yy, int P, float fx, float fy, float* N, float* dz) {
    int nb = std::min(cdiv(P, 256), 4096);
    hipLaunchKernelGGL(k_normals, dim3(nb), dim3(256), 0, st, z, zx, zy, xx, yy, P, fx, fy, N, dz);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// single-block finalisation of a partial array: out[0] = sum(part[0..n))
__global__ void k_final_sum(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ double smd[4];
    const double t = sum_partials(part, n, smd);
    if (threadIdx.x == 0) out[0] = (float)t;
}
int launch_final_sum(hipStream_t st, const float* part, int n, float* out) {
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, st, part, n, out);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// =============================================================================================
// depth: per-pixel photometric tensor (matrix-free form of devicecalls.cu:550-745)
//   g = rho_c/dz;  a1 = g (fx s0 - xx s2), a2 = g (fy s1 - yy s2), a3 = g s2   (dc.cu:588, 597)
//   b = I - rho_c s3 (N3 == 1)                                                (dc.cu:554, 573)
//   v = (a1, a2, -a3);  M = sum_{c,i} v v' (6 unique),  q = sum_{c, local i} v b (3)
// M needs only s (all images, replicated on every rank); q needs I (this rank's images).
// Results are scattered to the grid planes the CG works on.
// =============================================================================================
template <int V>
__global__ __launch_bounds__(256) void k_depth_assemble(const float* __restrict__ s, const float* __restrict__ rho,
                                                        const float* __restrict__ I, const float* __restrict__ xx,
                                                        const float* __restrict__ yy, const float* __restrict__ dz,
                                                        float fx, float fy, int P, int n_local, int C, int n_total,
                                                        int img_offset, const int* __restrict__ gofp, size_t plane,
                                                        float* __restrict__ M, float* __restrict__ Q, float* __restrict__ Gp, float* __restrict__ Qc /* compact [3][P] or null */) {
    const int q = (blockIdx.x * 256 + threadIdx.x) * V;
    if (q >= P) return;
    const Vec<V> vdz = ldv<V>(dz + q), vxx = ldv<V>(xx + q), vyy = ldv<V>(yy + q);
    const GridIdx<V> gi = grid_idx<V>(gofp, q);
    float m[6][V], qq[3][V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
#pragma unroll
        for (int t = 0; t < 6; ++t) m[t][e] = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) qq[t][e] = 0.f;
    }
    for (int c = 0; c < C; ++c) {
        const Vec<V> vr = ldv<V>(rho + (size_t)c * P + q);
        float g[V];
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = vr.v[e] / vdz.v[e];
        if (Gp) {                                                            // g_c^2 for the tensor-recompute operator
            float g2[V];
#pragma unroll
            for (int e = 0; e < V; ++e) g2[e] = g[e] * g[e];
            scatter_store<V>(Gp + (size_t)c * plane, gi, g2);
        }
        // M needs the lighting of ALL images (no image data): skipped when the operator rebuilds it
        if (M) {
            for (int i = 0; i < n_total; ++i) {
                const float* sv = s + ((size_t)i * C + c) * 4;
                const float s0 = sv[0], s1 = sv[1], s2 = sv[2];
                const float fs0 = fx * s0, fs1 = fy * s1;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float v0 = g[e] * (fs0 - vxx.v[e] * s2);
                    const float v1 = g[e] * (fs1 - vyy.v[e] * s2);
                    const float v2 = -(g[e] * s2);
                    m[0][e] = fmaf(v0, v0, m[0][e]);
                    m[1][e] = fmaf(v0, v1, m[1][e]);
                    m[2][e] = fmaf(v0, v2, m[2][e]);
                    m[3][e] = fmaf(v1, v1, m[3][e]);
                    m[4][e] = fmaf(v1, v2, m[4][e]);
                    m[5][e] = fmaf(v2, v2, m[5][e]);
                }
            }
        }
        // q needs the images of this rank
#pragma unroll 4
        for (int li = 0; li < n_local; ++li) {
            const float* sv = s + ((size_t)(img_offset + li) * C + c) * 4;
            const float s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
            const Vec<V> iv = ldv_stream<V>(I + ((size_t)li * C + c) * P + q);
            const float fs0 = fx * s0, fs1 = fy * s1;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float v0 = g[e] * (fs0 - vxx.v[e] * s2);
                const float v1 = g[e] * (fs1 - vyy.v[e] * s2);
                const float v2 = -(g[e] * s2);
                const float b = iv.v[e] - vr.v[e] * s3;
                qq[0][e] = fmaf(v0, b, qq[0][e]);
                qq[1][e] = fmaf(v1, b, qq[1][e]);
                qq[2][e] = fmaf(v2, b, qq[2][e]);
            }
        }
    }
    if (M) {
#pragma unroll
        for (int t = 0; t < 6; ++t) scatter_store<V>(M + (size_t)t * plane, gi, m[t]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (Qc) {                                              // sharded: compact [3][P] exchange buffer (scattered after the all-reduce)
            Vec<V> o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.v[e] = qq[t][e];
            stv<V>(Qc + (size_t)t * P + q, o);
        } else scatter_store<V>(Q + (size_t)t * plane, gi, qq[t]);
    }
}

// The same assembly without the pass over I: the image sums SA, SA', SB of every channel were left by the albedo sweep
// (k_albedo_numden<., true>), qc[c] = (CA, CA', CB) are the corresponding sums of s_i3 over this rank's images.
template <int V>
__global__ __launch_bounds__(256) void k_depth_from_sums(const float* __restrict__ s, const float* __restrict__ rho,
                                                         const float* __restrict__ ssum, const float* __restrict__ qc,
                                                         const float* __restrict__ xx, const float* __restrict__ yy,
                                                         const float* __restrict__ dz, float fx, float fy, int P, int C, int n_total,
                                                         const int* __restrict__ gofp, size_t plane,
                                                         float* __restrict__ M, float* __restrict__ Q, float* __restrict__ Gp, float* __restrict__ Qc /* compact [3][P] or null */,
                                                         int blk0) {
    const int q = ((blockIdx.x + blk0) * 256 + threadIdx.x) * V;
    if (q >= P) return;
    const Vec<V> vdz = ldv<V>(dz + q), vxx = ldv<V>(xx + q), vyy = ldv<V>(yy + q);
    const GridIdx<V> gi = grid_idx<V>(gofp, q);
    float m[6][V], qq[3][V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
#pragma unroll
        for (int t = 0; t < 6; ++t) m[t][e] = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) qq[t][e] = 0.f;
    }
    for (int c = 0; c < C; ++c) {
        const Vec<V> vr = ldv<V>(rho + (size_t)c * P + q);
        const Vec<V> sa = ldv<V>(ssum + ((size_t)c * 3 + 0) * P + q), sap = ldv<V>(ssum + ((size_t)c * 3 + 1) * P + q),
                     sb = ldv<V>(ssum + ((size_t)c * 3 + 2) * P + q);
        const float ca = qc[c * 4 + 0], cap = qc[c * 4 + 1], cb = qc[c * 4 + 2];
        float g[V];
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = vr.v[e] / vdz.v[e];
        if (Gp) {                                                            // g_c^2 for the tensor-recompute operator
            float g2[V];
#pragma unroll
            for (int e = 0; e < V; ++e) g2[e] = g[e] * g[e];
            scatter_store<V>(Gp + (size_t)c * plane, gi, g2);
        }
        if (M) {                                                             // stored tensor (no image data), as in k_depth_assemble
            for (int i = 0; i < n_total; ++i) {
                const float* sv = s + ((size_t)i * C + c) * 4;
                const float s0 = sv[0], s1 = sv[1], s2 = sv[2];
                const float fs0 = fx * s0, fs1 = fy * s1;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float v0 = g[e] * (fs0 - vxx.v[e] * s2);
                    const float v1 = g[e] * (fs1 - vyy.v[e] * s2);
                    const float v2 = -(g[e] * s2);
                    m[0][e] = fmaf(v0, v0, m[0][e]);
                    m[1][e] = fmaf(v0, v1, m[1][e]);
                    m[2][e] = fmaf(v0, v2, m[2][e]);
                    m[3][e] = fmaf(v1, v1, m[3][e]);
                    m[4][e] = fmaf(v1, v2, m[4][e]);
                    m[5][e] = fmaf(v2, v2, m[5][e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float t0 = fmaf(-vxx.v[e], sb.v[e], sa.v[e]), t1 = fmaf(-vyy.v[e], sb.v[e], sap.v[e]), t2 = -sb.v[e];
            const float u0 = fmaf(-vxx.v[e], cb, ca), u1 = fmaf(-vyy.v[e], cb, cap), u2 = -cb;
            qq[0][e] = fmaf(g[e], fmaf(-vr.v[e], u0, t0), qq[0][e]);
            qq[1][e] = fmaf(g[e], fmaf(-vr.v[e], u1, t1), qq[1][e]);
            qq[2][e] = fmaf(g[e], fmaf(-vr.v[e], u2, t2), qq[2][e]);
        }
    }
    if (M) {
#pragma unroll
        for (int t = 0; t < 6; ++t) scatter_store<V>(M + (size_t)t * plane, gi, m[t]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (Qc) {                                              // sharded: compact [3][P] exchange buffer (scattered after the all-reduce)
            Vec<V> o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.v[e] = qq[t][e];
            stv<V>(Qc + (size_t)t * P + q, o);
        } else scatter_store<V>(Q + (size_t)t * plane, gi, qq[t]);
    }
}

// Per-channel constants of the tensor-recompute form.  With a_i = fx s_i0, a'_i = fy s_i1, b_i = s_i2
// (image i, channel c) and sums S.. over the images:
//   sum_i t_i t_i' ,  t_i = (a_i - xx b_i, a'_i - yy b_i, -b_i)
//     = [[Sbb dx^2 + R00, Sbb dx dy + R01, Sbb dx], [., Sbb dy^2 + R11, Sbb dy], [., ., Sbb]]
//   dx = xx - Sab/Sbb, dy = yy - Sa'b/Sbb, R00 = Saa - Sab^2/Sbb, R01 = Saa' - Sab Sa'b/Sbb, R11 = Sa'a' - Sa'b^2/Sbb
// (completed squares: every term is non-negative, no cancellation).  Evaluated in double.
__global__ void k_tensor_consts(const float* __restrict__ s, int n_total, int C, float fx, float fy, float* __restrict__ out,
                                int n_local, int img_offset, float* __restrict__ qc /* [C][4], may be null */) {
    const int c = threadIdx.x;
    if (c >= C) return;
    if (qc) {      // (sum_i fx s_i0 s_i3, sum_i fy s_i1 s_i3, sum_i s_i2 s_i3, 0) over this rank's images: right-hand side from sums
        double ca = 0, cap = 0, cb = 0;
        for (int li = 0; li < n_local; ++li) {
            const float* sv = s + ((size_t)(img_offset + li) * C + c) * 4;
            ca += (double)(fx * sv[0]) * sv[3]; cap += (double)(fy * sv[1]) * sv[3]; cb += (double)sv[2] * sv[3];
        }
        qc[c * 4 + 0] = (float)ca; qc[c * 4 + 1] = (float)cap; qc[c * 4 + 2] = (float)cb; qc[c * 4 + 3] = 0.f;
    }
    double Saa = 0, Sab = 0, Sbb = 0, Saap = 0, Sapap = 0, Sapb = 0;
    for (int i = 0; i < n_total; ++i) {
        const float* sv = s + ((size_t)i * C + c) * 4;
        const double a = (double)fx * sv[0], ap = (double)fy * sv[1], b = sv[2];
        Saa += a * a; Sab += a * b; Sbb += b * b; Saap += a * ap; Sapap += ap * ap; Sapb += ap * b;
    }
    double xs = 0, ys = 0, R00 = Saa, R01 = Saap, R11 = Sapap;
    if (Sbb > 0) { xs = Sab / Sbb; ys = Sapb / Sbb; R00 = Saa - Sab * xs; R01 = Saap - Sab * ys; R11 = Sapap - Sapb * ys; }
    float* o = out + c * 8;
    o[0] = (float)Sbb; o[1] = (float)xs; o[2] = (float)ys; o[3] = (float)fmax(R00, 0.0); o[4] = (float)R01; o[5] = (float)fmax(R11, 0.0); o[6] = 0.f; o[7] = 0.f;
}

// q of the sharded depth phase, summed over the ranks in its compact [3][P] exchange buffer, onto the grid planes the solve reads
__global__ void k_scatter3(const float* __restrict__ compact, const int* __restrict__ gofp, int P, size_t plane, float* __restrict__ planes) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int g = gofp[p];
#pragma unroll
        for (int t = 0; t < 3; ++t) planes[(size_t)t * plane + g] = compact[(size_t)t * P + p];
    }
}
int depth_q_scatter(srps_ctx* ctx, const float* d_q_compact) {
    Grid& G = ctx->grid;
    hipLaunchKernelGGL(k_scatter3, dim3(std::min(cdiv(G.P, 256), 4096)), dim3(256), 0, ctx->stream, d_q_compact, G.d_gofp, G.P, G.plane, G.d_q);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// SRPS_ALBEDO_FUSED: what depth_assemble does besides its sweep -- the per-channel constants (tensor recompute form and the
// right-hand side's) and the grid's bookkeeping -- for an albedo sweep that writes g and q itself.  *ok = false: the fused sweep
// does not apply (no tensor-recompute operator: other channel counts, stored tensor wanted), the caller takes the unfused route.
int depth_fused_prepare(srps_ctx* ctx, const float* d_s, float fx, float fy, int C, int n_total, int n_local, int img_offset, float cx, float cy, bool* ok) {
    Grid& G = ctx->grid;
    const bool rec = ctx->tensor_recompute && (C == 1 || C == 3) && cx == cx && cy == cy;
    *ok = rec && march_supported(ctx) && ctx->apply_mode != SRPS_APPLY_SIMPLE && !ctx->keep_stored_tensor && G.G_planes >= (size_t)C;
    if (!*ok) return SRPS_OK;
    hipLaunchKernelGGL(k_tensor_consts, dim3(1), dim3(64), 0, ctx->stream, d_s, n_total, C, fx, fy, G.d_tconsts, n_local, img_offset, G.d_tconsts + 64);
    SRPS_LAUNCH_CHECK();
    G.tensor_channels = C; G.cx = cx; G.cy = cy;
    G.M_valid = false;
    return SRPS_OK;
}

int depth_assemble(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_I, const float* d_xx,
                   const float* d_yy, const float* d_dz, float fx, float fy, int P, int n_local, int C,
                   int n_total, int img_offset, float cx, float cy, const float* d_ssum, float* d_q_compact, int q0, int q1) {
    Grid& G = ctx->grid;
    const bool part = q1 > 0 && d_ssum != nullptr, later = part && q0 > 0;      // a later range of a chunked assembly: the constants exist already
    // tensor-recompute form needs the principal point (xx = j - cx, yy = i - cy are rebuilt in the kernel)
    const bool rec = ctx->tensor_recompute && (C == 1 || C == 3) && cx == cx && cy == cy;
    float* Gp = nullptr;
    if (rec) {
        SRPS_REQUIRE(G.G_planes >= (size_t)C, SRPS_ERR_STATE, "depth assembly: the grid holds %zu g planes, %d channels need %d", G.G_planes, C, C);
        Gp = G.d_G;                                        // [3][plane] of the grid's arena, zero outside the mask since the bind
        if (!later)
            hipLaunchKernelGGL(k_tensor_consts, dim3(1), dim3(64), 0, ctx->stream, d_s, n_total, C, fx, fy, G.d_tconsts, n_local, img_offset,
                               d_ssum ? G.d_tconsts + 64 : (float*)nullptr);
        G.tensor_channels = C; G.cx = cx; G.cy = cy;
    } else {
        G.tensor_channels = 0;
    }
    // the stored tensor is only needed by the simple operator kernel and by the stored-tensor marching form
    const bool want_M = !(rec && use_march(ctx) && !ctx->keep_stored_tensor);
    if (want_M) SRPS_TRY(grid_need_M(ctx));           // allocated (and zeroed) when the first assembly asks for it
    float* Mp = want_M ? G.d_M : nullptr;
    G.M_valid = Mp != nullptr;
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_rho | (uintptr_t)d_I | (uintptr_t)d_xx | (uintptr_t)d_yy | (uintptr_t)d_dz) % 16 == 0);
    if (d_ssum) {
        // the albedo sweep of this pass left the image sums: no second pass over I
        float* qc = G.d_tconsts + 64;                      // [8][4] behind the 8 x 8 tensor constants
        if (!rec && !later) hipLaunchKernelGGL(k_tensor_consts, dim3(1), dim3(64), 0, ctx->stream, d_s, n_total, C, fx, fy, G.d_tconsts, n_local, img_offset, qc);
        const int n_px = part ? std::min(q1, P) - q0 : P;
        if (vec && ((uintptr_t)d_ssum % 16 == 0))
            hipLaunchKernelGGL((k_depth_from_sums<4>), dim3(cdiv(n_px, 1024)), dim3(256), 0, ctx->stream, d_s, d_rho, d_ssum, qc, d_xx, d_yy, d_dz,
                               fx, fy, P, C, n_total, G.d_gofp, G.plane, Mp, G.d_q, Gp, d_q_compact, part ? q0 / 1024 : 0);
        else
            hipLaunchKernelGGL((k_depth_from_sums<1>), dim3(cdiv(n_px, 256)), dim3(256), 0, ctx->stream, d_s, d_rho, d_ssum, qc, d_xx, d_yy, d_dz,
                               fx, fy, P, C, n_total, G.d_gofp, G.plane, Mp, G.d_q, Gp, d_q_compact, part ? q0 / 256 : 0);
        SRPS_LAUNCH_CHECK();
        ctx->tensor_valid = true;
        return SRPS_OK;
    }
    if (vec)
        hipLaunchKernelGGL((k_depth_assemble<4>), dim3(cdiv(P, 1024)), dim3(256), 0, ctx->stream, d_s, d_rho, d_I, d_xx, d_yy, d_dz,
                           fx, fy, P, n_local, C, n_total, img_offset, G.d_gofp, G.plane, Mp, G.d_q, Gp, d_q_compact);
    else
        hipLaunchKernelGGL((k_depth_assemble<1>), dim3(cdiv(P, 256)), dim3(256), 0, ctx->stream, d_s, d_rho, d_I, d_xx, d_yy, d_dz,
                           fx, fy, P, n_local, C, n_total, img_offset, G.d_gofp, G.plane, Mp, G.d_q, Gp, d_q_compact);
    SRPS_LAUNCH_CHECK();
    ctx->tensor_valid = true;
    return SRPS_OK;
}

// photometric energy t2 = sum_{c,i,p} (a1 zx + a2 zy - a3 z - b)^2 over this rank's images
// (= ||A z - B||^2 of devicecalls.cu:763-767 without forming A)
template <int V>
__global__ __launch_bounds__(256) void k_energy_partial(const float* __restrict__ s, const float* __restrict__ rho,
                                                        const float* __restrict__ I, const float* __restrict__ xx,
                                                        const float* __restrict__ yy, const float* __restrict__ dz,
                                                        const float* __restrict__ z, const float* __restrict__ zx,
                                                        const float* __restrict__ zy, float fx, float fy, int P,
                                                        int n_local, int C, int img_offset, float* __restrict__ part) {
    __shared__ float sm[16];
    float acc = 0.f;
    for (int q = (blockIdx.x * 256 + threadIdx.x) * V; q < P; q += gridDim.x * 256 * V) {
        const Vec<V> vdz = ldv<V>(dz + q), vxx = ldv<V>(xx + q), vyy = ldv<V>(yy + q);
        const Vec<V> vz = ldv<V>(z + q), vzx = ldv<V>(zx + q), vzy = ldv<V>(zy + q);
        for (int c = 0; c < C; ++c) {
            const Vec<V> vr = ldv<V>(rho + (size_t)c * P + q);
            float g[V];
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] = vr.v[e] / vdz.v[e];
#pragma unroll 4
            for (int li = 0; li < n_local; ++li) {
                const float* sv = s + ((size_t)(img_offset + li) * C + c) * 4;
                const float s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
                const Vec<V> iv = ldv_stream<V>(I + ((size_t)li * C + c) * P + q);
                const float fs0 = fx * s0, fs1 = fy * s1;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float a1 = g[e] * (fs0 - vxx.v[e] * s2);
                    const float a2 = g[e] * (fs1 - vyy.v[e] * s2);
                    const float a3 = g[e] * s2;
                    const float b = iv.v[e] - vr.v[e] * s3;
                    const float res = a1 * vzx.v[e] + a2 * vzy.v[e] - a3 * vz.v[e] - b;
                    acc = fmaf(res, res, acc);
                }
            }
        }
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

int energy_photometric_partial(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_I,
                               const float* d_xx, const float* d_yy, const float* d_dz, const float* d_z,
                               const float* d_zx, const float* d_zy, float fx, float fy, int P, int n_local,
                               int C, int img_offset, float* d_out) {
    Grid& G = ctx->grid;
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_rho | (uintptr_t)d_I | (uintptr_t)d_xx | (uintptr_t)d_yy |
                                       (uintptr_t)d_dz | (uintptr_t)d_z | (uintptr_t)d_zx | (uintptr_t)d_zy) % 16 == 0);
    const int V = vec ? 4 : 1;
    const int nb = std::max(1, std::min(cdiv(P, 256 * V), 2048));
    float* part = G.d_misc_part;      // 4096 floats
    if (vec)
        hipLaunchKernelGGL((k_energy_partial<4>), dim3(nb), dim3(256), 0, ctx->stream, d_s, d_rho, d_I, d_xx, d_yy, d_dz, d_z, d_zx, d_zy, fx, fy, P, n_local, C, img_offset, part);
    else
        hipLaunchKernelGGL((k_energy_partial<1>), dim3(nb), dim3(256), 0, ctx->stream, d_s, d_rho, d_I, d_xx, d_yy, d_dz, d_z, d_zx, d_zy, fx, fy, P, n_local, C, img_offset, part);
    SRPS_LAUNCH_CHECK();
    return launch_final_sum(ctx->stream, part, nb, d_out);
}

}  // namespace srps
