// kernels_pixel.hip -- per-pixel phases on the compact (masked, reference-layout) arrays:
// init kernels, normals, lighting, albedo, depth-tensor assembly and the photometric energy.
// All of them stream I[n][c][p] once, coalesced along p; none has a stencil.
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
// Shared by k_normals and the fused energy + lighting pass, which must produce the same bits: every
// multiply-add is spelled out so that the compiler's contraction choices cannot differ between the two.
__device__ __forceinline__ void perspective_normal(float fx, float fy, float z, float gx, float gy, float x, float y,
                                                   float& n0, float& n1, float& n2, float& nrm) {
    const float u0 = fx * gx;                                      // dc.cu:204
    const float u1 = fy * gy;                                      // dc.cu:211
    const float u2 = fmaf(-y, gy, fmaf(-x, gx, -z));               // dc.cu:174
    nrm = fmaxf(1e-10f, sqrtf(fmaf(u2, u2, fmaf(u1, u1, u0 * u0))));   // dc.cu:182
    n0 = u0 / nrm;                                                 // dc.cu:190
    n1 = u1 / nrm;
    n2 = u2 / nrm;
}

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

// =============================================================================================
// lighting (reference: devicecalls.cu:376-444)
//   per channel c: A_c[p][k] = rho_c[p] N_k[p];  G_c = A_c' A_c (4x4, image independent);
//   per image i:   ATb_ic = A_c' I_ic;  s_ic <- CG(G_c, warm start s_ic, ATb_ic - G_c s_ic)
// Pass 1 streams I once and leaves per-block partial sums; pass 2 (one thread per (i,c)) adds
// them in a fixed order and runs the reference's CG recurrence on the 4x4 system in registers.
// =============================================================================================
// With ENERGY the same sweep over I also evaluates the photometric energy of the depth that was just
// solved (k_energy_partial's sum, with the lighting / albedo / dz the system was built from) and takes the
// normals of that depth from z, zx, zy instead of reading N: the energy pass of outer iteration k and the
// lighting pass of iteration k+1 read I once instead of twice.
struct EnergyArgs {
    const float *s, *xx, *yy, *dz, *z, *zx, *zy;
    float fx, fy;
    int img_offset;
    float* part_e;
};

template <int V, int IB, bool ENERGY>
__global__ __launch_bounds__(256) void k_light_partial(const float* __restrict__ rho, const float* __restrict__ N,
                                                       const float* __restrict__ I, int P, int n_img, int C, int chunk,
                                                       float* __restrict__ part_atb, float* __restrict__ part_g,
                                                       EnergyArgs ea) {
    __shared__ float sm[4][IB * 4 + 10];
    __shared__ float sme[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int blk = blockIdx.x;
    const int p0 = blk * chunk;
    const int p1 = min(P, p0 + chunk);
    float e_acc = 0.f;
    for (int c = 0; c < C; ++c) {
        for (int b0 = 0; b0 < n_img; b0 += IB) {
            float acc[IB][4];
            float g[10];
#pragma unroll
            for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[ii][k] = 0.f;
#pragma unroll
            for (int t = 0; t < 10; ++t) g[t] = 0.f;
            for (int q = p0 + tid * V; q < p1; q += 256 * V) {
                const Vec<V> r = ldv<V>(rho + (size_t)c * P + q);
                Vec<V> nk[4];
                Vec<V> vxx, vyy, vz, vzx, vzy, vg;
                if constexpr (ENERGY) {
                    vxx = ldv<V>(ea.xx + q); vyy = ldv<V>(ea.yy + q);
                    vz = ldv<V>(ea.z + q); vzx = ldv<V>(ea.zx + q); vzy = ldv<V>(ea.zy + q);
                    const Vec<V> vdz = ldv<V>(ea.dz + q);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        float nrm;
                        perspective_normal(ea.fx, ea.fy, vz.v[e], vzx.v[e], vzy.v[e], vxx.v[e], vyy.v[e],
                                           nk[0].v[e], nk[1].v[e], nk[2].v[e], nrm);
                        nk[3].v[e] = 1.f;
                        vg.v[e] = r.v[e] / vdz.v[e];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) nk[k] = ldv<V>(N + (size_t)k * P + q);
                }
                float a[4][V];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < V; ++e) a[k][e] = r.v[e] * nk[k].v[e];       // dc.cu:381
                // All IB loads are issued back to back (no branch between them): images past the end of the
                // batch re-read the last image (cache hits) and their sums are simply not stored.
                Vec<V> iv[IB];
#pragma unroll
                for (int ii = 0; ii < IB; ++ii) iv[ii] = ldv<V>(I + ((size_t)min(b0 + ii, n_img - 1) * C + c) * P + q);
#pragma unroll
                for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int e = 0; e < V; ++e) acc[ii][k] = fmaf(a[k][e], iv[ii].v[e], acc[ii][k]);
                if constexpr (ENERGY) {
#pragma unroll
                    for (int ii = 0; ii < IB; ++ii) {
                        if (b0 + ii < n_img) {                                   // wave-uniform
                            const float* sv = ea.s + ((size_t)(ea.img_offset + b0 + ii) * C + c) * 4;
                            const float s2 = sv[2], s3 = sv[3];
                            const float fs0 = ea.fx * sv[0], fs1 = ea.fy * sv[1];
#pragma unroll
                            for (int e = 0; e < V; ++e) {
                                const float a1 = vg.v[e] * (fs0 - vxx.v[e] * s2);
                                const float a2 = vg.v[e] * (fs1 - vyy.v[e] * s2);
                                const float a3 = vg.v[e] * s2;
                                const float b = iv[ii].v[e] - r.v[e] * s3;
                                const float res = a1 * vzx.v[e] + a2 * vzy.v[e] - a3 * vz.v[e] - b;
                                e_acc = fmaf(res, res, e_acc);
                            }
                        }
                    }
                }
                if (b0 == 0) {
                    int t = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int l = k; l < 4; ++l) {
#pragma unroll
                            for (int e = 0; e < V; ++e) g[t] = fmaf(a[k][e], a[l][e], g[t]);
                            ++t;
                        }
                }
            }
#pragma unroll
            for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = wave_sum(acc[ii][k]);
                    if (lane == 0) sm[wv][ii * 4 + k] = v;
                }
            if (b0 == 0) {
#pragma unroll
                for (int t = 0; t < 10; ++t) {
                    const float v = wave_sum(g[t]);
                    if (lane == 0) sm[wv][IB * 4 + t] = v;
                }
            }
            __syncthreads();
            if (tid < IB * 4) {
                const int ii = tid >> 2, k = tid & 3;
                if (b0 + ii < n_img)
                    part_atb[(((size_t)blk * n_img + b0 + ii) * C + c) * 4 + k] = sm[0][tid] + sm[1][tid] + sm[2][tid] + sm[3][tid];
            } else if (b0 == 0 && tid < IB * 4 + 10) {
                part_g[((size_t)blk * C + c) * 10 + (tid - IB * 4)] = sm[0][tid] + sm[1][tid] + sm[2][tid] + sm[3][tid];
            }
            __syncthreads();
        }
    }
    if constexpr (ENERGY) {
        const float t = block_sum(e_acc, sme);
        if (tid == 0) ea.part_e[blk] = t;
    }
}

// The same sums with the images dealt to four BLOCKS per pixel range (image group g of every batch of 4*IBW images): the four
// waves of a block read 4 KiB of consecutive pixels of each plane, with 20 accumulators per thread instead of 80 (the kernel
// above holds 209 registers in its fused form, two waves per SIMD).  The four blocks of a pixel range are dispatched next to each other on the same XCD
// (block id -> (range, group) below), so that part of the geometry re-reads hit the L2 (PMC: 1.42 GB fetched per sweep against
// 1.16 GB touched; the sweep runs at 5.3 TB/s of fabric traffic).
template <int V, int IBW, bool ENERGY>
__global__ __launch_bounds__(256) void k_light_grouped(const float* __restrict__ rho, const float* __restrict__ N,
                                                         const float* __restrict__ I, int P, int n_img, int C, int chunk,
                                                         float* __restrict__ part_atb, float* __restrict__ part_g,
                                                         EnergyArgs ea) {
    __shared__ float sme[16];
    __shared__ float smr[4][IBW * 4 + 10];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blocks b, b+8, ... share an XCD: the four image groups of a pixel range are 8 apart in dispatch order
    const int xcd = blockIdx.x & 7, t8 = blockIdx.x >> 3;
    const int grp = t8 & 3;
    const int blk = (t8 >> 2) * 8 + xcd;                    // pixel range
    if (blk * chunk >= P) { if (ENERGY && tid == 0) ea.part_e[blockIdx.x] = 0.f; return; }
    const int p0 = blk * chunk;
    const int p1 = min(P, p0 + chunk);
    float e_acc = 0.f;
    for (int c = 0; c < C; ++c) {
        for (int b0 = 0; b0 < n_img; b0 += 4 * IBW) {
            const int ib = b0 + grp * IBW;                 // first image of this block (may be past the end: nothing stored)
            const bool gram = (b0 == 0 && grp == 0);
            float acc[IBW][4];
            float g[10];
#pragma unroll
            for (int ii = 0; ii < IBW; ++ii)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[ii][k] = 0.f;
#pragma unroll
            for (int t = 0; t < 10; ++t) g[t] = 0.f;
            for (int q = p0 + tid * V; q < p1; q += 256 * V) {
                const Vec<V> r = ldv<V>(rho + (size_t)c * P + q);
                Vec<V> nk[4];
                Vec<V> vxx, vyy, vz, vzx, vzy, vg;
                if constexpr (ENERGY) {
                    vxx = ldv<V>(ea.xx + q); vyy = ldv<V>(ea.yy + q);
                    vz = ldv<V>(ea.z + q); vzx = ldv<V>(ea.zx + q); vzy = ldv<V>(ea.zy + q);
                    const Vec<V> vdz = ldv<V>(ea.dz + q);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        float nrm;
                        perspective_normal(ea.fx, ea.fy, vz.v[e], vzx.v[e], vzy.v[e], vxx.v[e], vyy.v[e],
                                           nk[0].v[e], nk[1].v[e], nk[2].v[e], nrm);
                        nk[3].v[e] = 1.f;
                        vg.v[e] = r.v[e] / vdz.v[e];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) nk[k] = ldv<V>(N + (size_t)k * P + q);
                }
                float a[4][V];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < V; ++e) a[k][e] = r.v[e] * nk[k].v[e];       // dc.cu:381
                Vec<V> iv[IBW];                                      // images past the end re-read the last one
#pragma unroll
                for (int ii = 0; ii < IBW; ++ii) iv[ii] = ldv<V>(I + ((size_t)min(ib + ii, n_img - 1) * C + c) * P + q);
#pragma unroll
                for (int ii = 0; ii < IBW; ++ii)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int e = 0; e < V; ++e) acc[ii][k] = fmaf(a[k][e], iv[ii].v[e], acc[ii][k]);
                if constexpr (ENERGY) {
                    // residual a1 zx + a2 zy - a3 z - (I - rho s3) of k_energy_partial, factored by the lighting vector:
                    // (g fx zx) s0 + (g fy zy) s1 - g (xx zx + yy zy + z) s2 + rho s3 - I   (5 instead of 11 operations per image)
                    float E[3][V];
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        E[0][e] = vg.v[e] * (ea.fx * vzx.v[e]);
                        E[1][e] = vg.v[e] * (ea.fy * vzy.v[e]);
                        E[2][e] = -vg.v[e] * fmaf(vyy.v[e], vzy.v[e], fmaf(vxx.v[e], vzx.v[e], vz.v[e]));
                    }
#pragma unroll
                    for (int ii = 0; ii < IBW; ++ii) {
                        if (ib + ii < n_img) {                                   // wave-uniform
                            const float* sv = ea.s + ((size_t)(ea.img_offset + ib + ii) * C + c) * 4;
                            const float s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
#pragma unroll
                            for (int e = 0; e < V; ++e) {
                                const float res = fmaf(E[0][e], s0, fmaf(E[1][e], s1, fmaf(E[2][e], s2, fmaf(r.v[e], s3, -iv[ii].v[e]))));
                                e_acc = fmaf(res, res, e_acc);
                            }
                        }
                    }
                }
                if (gram) {                                                      // wave-uniform
                    int t = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int l = k; l < 4; ++l) {
#pragma unroll
                            for (int e = 0; e < V; ++e) g[t] = fmaf(a[k][e], a[l][e], g[t]);
                            ++t;
                        }
                }
            }
#pragma unroll
            for (int ii = 0; ii < IBW; ++ii)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = wave_sum(acc[ii][k]);
                    if (lane == 0) smr[wv][ii * 4 + k] = v;
                }
            if (gram) {
#pragma unroll
                for (int t = 0; t < 10; ++t) {
                    const float v = wave_sum(g[t]);
                    if (lane == 0) smr[wv][IBW * 4 + t] = v;
                }
            }
            __syncthreads();
            if (tid < IBW * 4) {
                const int ii = tid >> 2, k = tid & 3;
                if (ib + ii < n_img)
                    part_atb[(((size_t)blk * n_img + ib + ii) * C + c) * 4 + k] = smr[0][tid] + smr[1][tid] + smr[2][tid] + smr[3][tid];
            } else if (gram && tid < IBW * 4 + 10) {
                part_g[((size_t)blk * C + c) * 10 + (tid - IBW * 4)] = smr[0][tid] + smr[1][tid] + smr[2][tid] + smr[3][tid];
            }
            __syncthreads();
        }
    }
    if constexpr (ENERGY) {
        const float t = block_sum(e_acc, sme);
        if (tid == 0) ea.part_e[blockIdx.x] = t;
    }
}

// The fused energy + lighting sweep with the channels in the INNER loop (NCH = C known at compile time): the six geometry
// planes are loaded and the normal is formed once per pixel and image group instead of once per channel (k_light_grouped:
// 12 reads of the geometry per pixel, about a third of them from HBM).  Same arithmetic, same sums, same bits as
// k_light_grouped<V, IBW, true>; the Gram matrix of channel c is accumulated by image group c.
template <int V, int IBW, int NCH>
__global__ __launch_bounds__(256) void k_light_fused_ci(const float* __restrict__ rho, const float* __restrict__ I, int P, int n_img,
                                                          int chunk, float* __restrict__ part_atb, float* __restrict__ part_g,
                                                          EnergyArgs ea) {
    constexpr int C = NCH;
    __shared__ float sme[16];
    __shared__ float smr[4][NCH * IBW * 4 + 10];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, t8 = blockIdx.x >> 3;
    const int grp = t8 & 3;
    const int blk = (t8 >> 2) * 8 + xcd;                    // pixel range
    if (blk * chunk >= P) { if (tid == 0) ea.part_e[blockIdx.x] = 0.f; return; }
    const int p0 = blk * chunk;
    const int p1 = min(P, p0 + chunk);
    float e_acc = 0.f;
    for (int b0 = 0; b0 < n_img; b0 += 4 * IBW) {
        const int ib = b0 + grp * IBW;                     // first image of this block (may be past the end: nothing stored)
        const int gram_c = (b0 == 0 && grp < NCH) ? grp : -1;      // the channel whose Gram matrix this block accumulates
        float acc[NCH][IBW][4];
        float g[10];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int ii = 0; ii < IBW; ++ii)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[c][ii][k] = 0.f;
#pragma unroll
        for (int t = 0; t < 10; ++t) g[t] = 0.f;
        for (int q = p0 + tid * V; q < p1; q += 256 * V) {
            Vec<V> nk[3], T[3];
            const Vec<V> vdz = ldv<V>(ea.dz + q);
            {
                const Vec<V> vxx = ldv<V>(ea.xx + q), vyy = ldv<V>(ea.yy + q);
                const Vec<V> vz = ldv<V>(ea.z + q), vzx = ldv<V>(ea.zx + q), vzy = ldv<V>(ea.zy + q);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    float nrm;
                    perspective_normal(ea.fx, ea.fy, vz.v[e], vzx.v[e], vzy.v[e], vxx.v[e], vyy.v[e], nk[0].v[e], nk[1].v[e], nk[2].v[e], nrm);
                    T[0].v[e] = ea.fx * vzx.v[e];
                    T[1].v[e] = ea.fy * vzy.v[e];
                    T[2].v[e] = fmaf(vyy.v[e], vzy.v[e], fmaf(vxx.v[e], vzx.v[e], vz.v[e]));
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const Vec<V> r = ldv<V>(rho + (size_t)c * P + q);
                float a[4][V];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    a[0][e] = r.v[e] * nk[0].v[e]; a[1][e] = r.v[e] * nk[1].v[e]; a[2][e] = r.v[e] * nk[2].v[e];       // dc.cu:381
                    a[3][e] = r.v[e] * 1.f;
                }
                Vec<V> iv[IBW];                                      // images past the end re-read the last one
#pragma unroll
                for (int ii = 0; ii < IBW; ++ii) iv[ii] = ldv<V>(I + ((size_t)min(ib + ii, n_img - 1) * C + c) * P + q);
#pragma unroll
                for (int ii = 0; ii < IBW; ++ii)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int e = 0; e < V; ++e) acc[c][ii][k] = fmaf(a[k][e], iv[ii].v[e], acc[c][ii][k]);
                float E[3][V];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float vg = r.v[e] / vdz.v[e];
                    E[0][e] = vg * T[0].v[e];
                    E[1][e] = vg * T[1].v[e];
                    E[2][e] = -vg * T[2].v[e];
                }
#pragma unroll
                for (int ii = 0; ii < IBW; ++ii) {
                    if (ib + ii < n_img) {                                   // wave-uniform
                        const float* sv = ea.s + ((size_t)(ea.img_offset + ib + ii) * C + c) * 4;
                        const float s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const float res = fmaf(E[0][e], s0, fmaf(E[1][e], s1, fmaf(E[2][e], s2, fmaf(r.v[e], s3, -iv[ii].v[e]))));
                            e_acc = fmaf(res, res, e_acc);
                        }
                    }
                }
                if (c == gram_c) {                                           // wave-uniform
                    int t = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int l = k; l < 4; ++l) {
#pragma unroll
                            for (int e = 0; e < V; ++e) g[t] = fmaf(a[k][e], a[l][e], g[t]);
                            ++t;
                        }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int ii = 0; ii < IBW; ++ii)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = wave_sum(acc[c][ii][k]);
                    if (lane == 0) smr[wv][(c * IBW + ii) * 4 + k] = v;
                }
        if (gram_c >= 0) {
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                const float v = wave_sum(g[t]);
                if (lane == 0) smr[wv][NCH * IBW * 4 + t] = v;
            }
        }
        __syncthreads();
        if (tid < NCH * IBW * 4) {
            const int c = tid / (IBW * 4), ii = (tid / 4) % IBW, k = tid & 3;
            if (ib + ii < n_img)
                part_atb[(((size_t)blk * n_img + ib + ii) * C + c) * 4 + k] = smr[0][tid] + smr[1][tid] + smr[2][tid] + smr[3][tid];
        } else if (gram_c >= 0 && tid < NCH * IBW * 4 + 10) {
            part_g[((size_t)blk * C + gram_c) * 10 + (tid - NCH * IBW * 4)] = smr[0][tid] + smr[1][tid] + smr[2][tid] + smr[3][tid];
        }
        __syncthreads();
    }
    const float t = block_sum(e_acc, sme);
    if (tid == 0) ea.part_e[blockIdx.x] = t;
}

// one wave per (image, channel) of the WHOLE image set; non-local rows are zeroed when sharded.
// The 64 lanes add the per-block partial sums (fixed order, double), lane 0 runs the 4x4 CG.
__global__ __launch_bounds__(64) void k_light_solve(const float* __restrict__ part_atb, const float* __restrict__ part_g, int nblk,
                              int n_local, int C, int n_total, int img_offset, int zero_nonlocal,
                              float* __restrict__ s, int* __restrict__ iters_max, float tol, int max_iter) {
    const int t = blockIdx.x;
    const int lane = threadIdx.x;
    if (t >= n_total * C) return;
    const int i = t / C, c = t - i * C;
    const int li = i - img_offset;
    float* sv = s + (size_t)t * 4;
    if (li < 0 || li >= n_local) {
        if (zero_nonlocal && lane < 4) sv[lane] = 0.f;
        return;
    }
    double Gd[10], bd[4];
    for (int u = 0; u < 10; ++u) Gd[u] = 0.0;
    for (int k = 0; k < 4; ++k) bd[k] = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        for (int u = 0; u < 10; ++u) Gd[u] += (double)part_g[((size_t)b * C + c) * 10 + u];
        for (int k = 0; k < 4; ++k) bd[k] += (double)part_atb[(((size_t)b * n_local + li) * C + c) * 4 + k];
    }
    for (int u = 0; u < 10; ++u) Gd[u] = wave_sum(Gd[u]);
    for (int k = 0; k < 4; ++k) bd[k] = wave_sum(bd[k]);
    if (lane != 0) return;
    float A[4][4];
    {
        int u = 0;
        for (int k = 0; k < 4; ++k)
            for (int l = k; l < 4; ++l) { A[k][l] = (float)Gd[u]; A[l][k] = A[k][l]; ++u; }    // sgemm dc.cu:422
    }
    float x[4], r[4], p[4], w[4];
    for (int k = 0; k < 4; ++k) x[k] = sv[k];
    for (int k = 0; k < 4; ++k) {                                                          // sgemv dc.cu:423-424
        float acc = (float)bd[k];
        for (int l = 0; l < 4; ++l) acc -= A[k][l] * x[l];
        r[k] = acc;
    }
    // cuda_based_conjugate_gradient on the 4x4 system, dc.cu:251-275
    float r1 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
    float r0 = 0.f;
    int k = 0;
    while (r1 > tol * tol && k <= max_iter) {
        ++k;
        if (k == 1) {
            for (int u = 0; u < 4; ++u) p[u] = r[u];
        } else {
            const float beta = r1 / r0;
            for (int u = 0; u < 4; ++u) p[u] = beta * p[u];
            for (int u = 0; u < 4; ++u) p[u] = p[u] + r[u];
        }
        for (int u = 0; u < 4; ++u) w[u] = A[u][0] * p[0] + A[u][1] * p[1] + A[u][2] * p[2] + A[u][3] * p[3];
        const float dot = p[0] * w[0] + p[1] * w[1] + p[2] * w[2] + p[3] * w[3];
        const float alpha = r1 / dot;
        for (int u = 0; u < 4; ++u) x[u] = x[u] + alpha * p[u];
        for (int u = 0; u < 4; ++u) r[u] = r[u] - alpha * w[u];
        r0 = r1;
        r1 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
    }
    for (int u = 0; u < 4; ++u) sv[u] = x[u];
    atomicMax(iters_max, k);
}

struct LightPlan {
    int V, IB, chunk, nblk;
    int n_epart;               // energy partial sums the fused sweep leaves (one per launched block)
    float *part_atb, *part_g;
    int* d_it;
};
static int light_plan(srps_ctx* ctx, bool vec, int P, int n_local, int C, LightPlan& L, bool fused = false) {
    // images per register batch: the block re-reads rho and N once per batch, so one batch is best
    L.IB = 4;
    for (int cand : {4, 8, 12, 16, 20}) { L.IB = cand; if (n_local <= cand) break; }      // n_local > 20: batches of 20
    // the fused sweep holds 7 more planes per pixel: 2 pixels per thread keep it at 2 waves per SIMD
    // (334 us against 504 us at 2048^2, 20 images); alone the lighting sweep is faster with 4 (250 / 268 us)
    L.V = vec ? ((fused && !ctx->light_grouped) ? 2 : 4) : 1;
    if (ctx->light_grouped && L.V == 4) {
        // one round of blocks: the sweep keeps 3 (fused, 135 registers) or 5 (96 registers) blocks per CU resident; with
        // 1024 blocks the last third of the kernel ran at a third of the occupancy
        const int target = (ctx->light_blocks > 0 ? ctx->light_blocks : ctx->num_cus * (fused ? 3 : 5)) / 4;      // pixel ranges
        const int gran = 256 * L.V;                        // a block covers 256 V pixels per iteration
        L.chunk = std::max(gran, cdiv(cdiv(P, target), gran) * gran);
    } else {
        const int chunk = cdiv(P, 1024);
        L.chunk = std::max(256 * L.V, cdiv(chunk, 256 * L.V) * 256 * L.V);
    }
    L.nblk = cdiv(P, L.chunk);
    L.n_epart = (ctx->light_grouped && L.V == 4) ? cdiv(L.nblk, 8) * 32 : L.nblk;
    const size_t n_atb = (size_t)L.nblk * std::max(n_local, 1) * C * 4, n_g = (size_t)L.nblk * C * 10;
    SRPS_TRY(ensure(ctx->ws_light, (n_atb + n_g) * sizeof(float) + 64));
    L.part_atb = (float*)ctx->ws_light.p;
    L.part_g = L.part_atb + n_atb;
    L.d_it = (int*)(L.part_g + n_g);
    return SRPS_OK;
}
template <bool ENERGY>
static int light_partial_launch(srps_ctx* ctx, const LightPlan& L, const float* d_rho, const float* d_N, const float* d_I,
                                int P, int n_local, int C, const EnergyArgs& ea) {
    if (ctx->light_grouped && L.V == 4) {
        const int ibw = std::min(5, cdiv(n_local, 4));
        const int nb4 = cdiv(L.nblk, 8) * 8 * 4;            // four image groups per pixel range, ranges in sets of 8 (one per XCD)
        if (ENERGY && ctx->light_channel_inner && (C == 1 || C == 3)) {
#define SRPS_LCI(BB, CC) hipLaunchKernelGGL((k_light_fused_ci<4, BB, CC>), dim3(nb4), dim3(256), 0, ctx->stream, d_rho, d_I, P, n_local, L.chunk, L.part_atb, L.part_g, ea)
            if (C == 3) { switch (ibw) { case 1: SRPS_LCI(1, 3); break; case 2: SRPS_LCI(2, 3); break; case 3: SRPS_LCI(3, 3); break; case 4: SRPS_LCI(4, 3); break; default: SRPS_LCI(5, 3); } }
            else { switch (ibw) { case 1: SRPS_LCI(1, 1); break; case 2: SRPS_LCI(2, 1); break; case 3: SRPS_LCI(3, 1); break; case 4: SRPS_LCI(4, 1); break; default: SRPS_LCI(5, 1); } }
#undef SRPS_LCI
            SRPS_LAUNCH_CHECK();
            return SRPS_OK;
        }
#define SRPS_LGR(BB) hipLaunchKernelGGL((k_light_grouped<4, BB, ENERGY>), dim3(nb4), dim3(256), 0, ctx->stream, d_rho, d_N, d_I, P, n_local, C, L.chunk, L.part_atb, L.part_g, ea)
        switch (ibw) { case 1: SRPS_LGR(1); break; case 2: SRPS_LGR(2); break; case 3: SRPS_LGR(3); break; case 4: SRPS_LGR(4); break; default: SRPS_LGR(5); }
#undef SRPS_LGR
        SRPS_LAUNCH_CHECK();
        return SRPS_OK;
    }
#define SRPS_LIGHT(VV, BB) hipLaunchKernelGGL((k_light_partial<VV, BB, ENERGY>), dim3(L.nblk), dim3(256), 0, ctx->stream, d_rho, d_N, d_I, P, n_local, C, L.chunk, L.part_atb, L.part_g, ea)
    if (L.V == 4) { if constexpr (!ENERGY) switch (L.IB) { case 4: SRPS_LIGHT(4, 4); break; case 8: SRPS_LIGHT(4, 8); break; case 12: SRPS_LIGHT(4, 12); break; case 16: SRPS_LIGHT(4, 16); break; default: SRPS_LIGHT(4, 20); } }
    else if (L.V == 2) { if constexpr (ENERGY) switch (L.IB) { case 4: SRPS_LIGHT(2, 4); break; case 8: SRPS_LIGHT(2, 8); break; case 12: SRPS_LIGHT(2, 12); break; case 16: SRPS_LIGHT(2, 16); break; default: SRPS_LIGHT(2, 20); } }
    else { switch (L.IB) { case 4: SRPS_LIGHT(1, 4); break; case 8: SRPS_LIGHT(1, 8); break; case 12: SRPS_LIGHT(1, 12); break; case 16: SRPS_LIGHT(1, 16); break; default: SRPS_LIGHT(1, 20); } }
#undef SRPS_LIGHT
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// use_cache: the partial sums left by energy_light_fused for exactly these arrays are still in ws_light
int lighting(srps_ctx* ctx, float* d_s, const float* d_rho, const float* d_N, const float* d_I, int P,
             int n_local, int C, int n_total, int img_offset, bool zero_nonlocal, bool use_cache) {
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_rho | (uintptr_t)d_N | (uintptr_t)d_I) % 16 == 0);
    LightPlan L;
    const bool cached = use_cache && ctx->light_cache_valid && ctx->light_cache_normals;
    ctx->light_cache_valid = false;
    if (cached) {
        L = LightPlan{ctx->light_cache_V, 0, 0, ctx->light_cache_nblk, 0, nullptr, nullptr, nullptr};
        const size_t n_atb = (size_t)L.nblk * std::max(n_local, 1) * C * 4, n_g = (size_t)L.nblk * C * 10;
        L.part_atb = (float*)ctx->ws_light.p;
        L.part_g = L.part_atb + n_atb;
        L.d_it = (int*)(L.part_g + n_g);
    } else {
        SRPS_TRY(light_plan(ctx, vec, P, n_local, C, L));
    }
    SRPS_HIP(hipMemsetAsync(L.d_it, 0, sizeof(int), ctx->stream));
    if (n_local > 0 && !cached) SRPS_TRY(light_partial_launch<false>(ctx, L, d_rho, d_N, d_I, P, n_local, C, EnergyArgs{}));
    const int nt = n_total * C;
    hipLaunchKernelGGL(k_light_solve, dim3(nt), dim3(64), 0, ctx->stream, L.part_atb, L.part_g, L.nblk, n_local, C,
                       n_total, img_offset, zero_nonlocal ? 1 : 0, d_s, L.d_it, ctx->cg_tol, ctx->cg_max_iter);
    SRPS_LAUNCH_CHECK();
    SRPS_HIP(hipMemcpyAsync(ctx->h_pinned + 8, L.d_it, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return SRPS_OK;
}

// Photometric energy of the depth just solved (k_energy_partial's quantity) and, in the same sweep over I, the
// lighting partial sums of the next outer iteration (normals of the new depth computed in registers). The sums
// stay in ws_light; lighting(..., use_cache) consumes them.
int energy_light_fused(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_I, const float* d_xx,
                       const float* d_yy, const float* d_dz, const float* d_z, const float* d_zx, const float* d_zy,
                       float fx, float fy, int P, int n_local, int C, int img_offset, float* d_out) {
    Grid& G = ctx->grid;
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_rho | (uintptr_t)d_I | (uintptr_t)d_xx | (uintptr_t)d_yy |
                                       (uintptr_t)d_dz | (uintptr_t)d_z | (uintptr_t)d_zx | (uintptr_t)d_zy) % 16 == 0);
    LightPlan L;
    SRPS_TRY(light_plan(ctx, vec, P, n_local, C, L, /*fused=*/true));
    EnergyArgs ea{d_s, d_xx, d_yy, d_dz, d_z, d_zx, d_zy, fx, fy, img_offset, G.d_misc_part};
    SRPS_TRY(light_partial_launch<true>(ctx, L, d_rho, nullptr, d_I, P, n_local, C, ea));
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, ctx->stream, G.d_misc_part, L.n_epart, d_out);
    SRPS_LAUNCH_CHECK();
    ctx->light_cache_valid = true;
    ctx->light_cache_normals = false;
    ctx->light_cache_V = L.V;
    ctx->light_cache_nblk = L.nblk;
    return SRPS_OK;
}

// =============================================================================================
// albedo (reference: devicecalls.cu:447-548)
//   sh_i[p] = N[:,p] . s_ic  (sgemm dc.cu:507);  num = sum_i sh_i I_ic, den = sum_i sh_i^2
//   (the diagonal A'A and A'b of dc.cu:395-406);  then the reference's global CG on the
//   diagonal system from the warm start rho_c (dc.cu:540), or its fixed point num/den.
// =============================================================================================
template <int V>
__global__ __launch_bounds__(256) void k_albedo_numden(const float* __restrict__ s, const float* __restrict__ N,
                                                       const float* __restrict__ I, int P, int n_local, int C,
                                                       int s_img_offset, float* __restrict__ num, float* __restrict__ den) {
    const int q = (blockIdx.x * 256 + threadIdx.x) * V;
    if (q >= P) return;
    Vec<V> nk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) nk[k] = ldv<V>(N + (size_t)k * P + q);
    for (int c = 0; c < C; ++c) {
        Vec<V> nu, de;
#pragma unroll
        for (int e = 0; e < V; ++e) { nu.v[e] = 0.f; de.v[e] = 0.f; }
#pragma unroll 4
        for (int i = 0; i < n_local; ++i) {
            const float* sv = s + ((size_t)(s_img_offset + i) * C + c) * 4;      // uniform -> scalar loads
            const float s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
            const Vec<V> iv = ldv<V>(I + ((size_t)i * C + c) * P + q);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float sh = nk[0].v[e] * s0 + nk[1].v[e] * s1 + nk[2].v[e] * s2 + nk[3].v[e] * s3;
                nu.v[e] = fmaf(sh, iv.v[e], nu.v[e]);
                de.v[e] = fmaf(sh, sh, de.v[e]);
            }
        }
        stv<V>(num + (size_t)c * P + q, nu);
        stv<V>(den + (size_t)c * P + q, de);
    }
}

int albedo_numden(srps_ctx* ctx, const float* d_s, const float* d_N, const float* d_I, int P, int n_local,
                  int C, int s_img_offset, float* d_numden) {
    float* num = d_numden;
    float* den = d_numden + (size_t)C * P;
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_N | (uintptr_t)d_I | (uintptr_t)d_numden) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL((k_albedo_numden<4>), dim3(cdiv(P, 1024)), dim3(256), 0, ctx->stream, d_s, d_N, d_I, P, n_local, C, s_img_offset, num, den);
    else
        hipLaunchKernelGGL((k_albedo_numden<1>), dim3(cdiv(P, 256)), dim3(256), 0, ctx->stream, d_s, d_N, d_I, P, n_local, C, s_img_offset, num, den);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

__global__ void k_albedo_closed(float* __restrict__ rho, const float* __restrict__ num, const float* __restrict__ den, size_t n) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const float d = den[t];
        if (d > 0.f) rho[t] = num[t] / d;
    }
}

struct DcgScal {
    float r0;
    int iters;
    int active;
    int pad;
};

// r = num - den*rho (dc.cu:404-405); rr_part[c][0][blk] = sum r^2
template <int V>
__global__ __launch_bounds__(256) void k_dcg_init(const float* __restrict__ rho, const float* __restrict__ num,
                                                  const float* __restrict__ den, int P, float* __restrict__ r,
                                                  float* __restrict__ rr_part, int nb, DcgScal* __restrict__ scal) {
    __shared__ float sm[16];
    const int c = blockIdx.y;
    const size_t base = (size_t)c * P;
    float acc = 0.f;
    for (int p = (blockIdx.x * 256 + threadIdx.x) * V; p < P; p += nb * 256 * V) {
        const Vec<V> vn = ldv<V>(num + base + p), vd = ldv<V>(den + base + p), vx = ldv<V>(rho + base + p);
        Vec<V> vr;
#pragma unroll
        for (int e = 0; e < V; ++e) { vr.v[e] = vn.v[e] - vd.v[e] * vx.v[e]; acc = fmaf(vr.v[e], vr.v[e], acc); }
        stv<V>(r + base + p, vr);
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        rr_part[((size_t)c * 2 + 0) * nb + blockIdx.x] = t;
        if (blockIdx.x == 0) { scal[c].r0 = 0.f; scal[c].iters = 0; scal[c].active = 1; }
    }
}

// first half of CG step k: p = beta p + r ; partial p.(d p)
template <int V>
__global__ __launch_bounds__(256) void k_dcg_a(int k, const float* __restrict__ den, const float* __restrict__ r,
                                               float* __restrict__ p, int P, const float* __restrict__ rr_part,
                                               float* __restrict__ pw_part, int nb, DcgScal* __restrict__ scal, float tol2) {
    __shared__ float sm[16];
    __shared__ double smd[4];
    const int c = blockIdx.y;
    const float r1 = (float)sum_partials(rr_part + ((size_t)c * 2 + ((k - 1) & 1)) * nb, nb, smd);
    if (!(r1 > tol2)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[c].active = 0;
        return;
    }
    const float beta = (k == 1) ? 0.f : r1 / scal[c].r0;
    const size_t base = (size_t)c * P;
    float acc = 0.f;
    for (int q = (blockIdx.x * 256 + threadIdx.x) * V; q < P; q += nb * 256 * V) {
        const Vec<V> vr = ldv<V>(r + base + q), vd = ldv<V>(den + base + q);
        Vec<V> vp;
        if (k == 1) vp = vr;
        else {
            vp = ldv<V>(p + base + q);
#pragma unroll
            for (int e = 0; e < V; ++e) vp.v[e] = scal_then_axpy(beta, vp.v[e], vr.v[e]);             // dc.cu:263-264
        }
        stv<V>(p + base + q, vp);
#pragma unroll
        for (int e = 0; e < V; ++e) acc = fmaf(vp.v[e], vd.v[e] * vp.v[e], acc);
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) pw_part[(size_t)c * nb + blockIdx.x] = t;
}

// second half: alpha = r1 / p.w ; x += alpha p ; r -= alpha w ; partial r.r
template <int V>
__global__ __launch_bounds__(256) void k_dcg_b(int k, const float* __restrict__ den, float* __restrict__ r,
                                               const float* __restrict__ p, float* __restrict__ x, int P,
                                               float* __restrict__ rr_part, const float* __restrict__ pw_part, int nb,
                                               DcgScal* __restrict__ scal, float tol2) {
    __shared__ float sm[16];
    __shared__ double smd[4];
    const int c = blockIdx.y;
    const float* rr_old = rr_part + ((size_t)c * 2 + ((k - 1) & 1)) * nb;
    float* rr_new = rr_part + ((size_t)c * 2 + (k & 1)) * nb;
    const float r1 = (float)sum_partials(rr_old, nb, smd);
    if (!(r1 > tol2)) {
        if (threadIdx.x == 0) rr_new[blockIdx.x] = rr_old[blockIdx.x];
        return;
    }
    const float dot = (float)sum_partials(pw_part + (size_t)c * nb, nb, smd);
    const float alpha = r1 / dot;
    const size_t base = (size_t)c * P;
    float acc = 0.f;
    for (int q = (blockIdx.x * 256 + threadIdx.x) * V; q < P; q += nb * 256 * V) {
        const Vec<V> vp = ldv<V>(p + base + q), vd = ldv<V>(den + base + q);
        Vec<V> vx = ldv<V>(x + base + q), vr = ldv<V>(r + base + q);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float w = vd.v[e] * vp.v[e];
            vx.v[e] = fmaf(alpha, vp.v[e], vx.v[e]);
            vr.v[e] = fmaf(-alpha, w, vr.v[e]);
            acc = fmaf(vr.v[e], vr.v[e], acc);
        }
        stv<V>(x + base + q, vx);
        stv<V>(r + base + q, vr);
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        rr_new[blockIdx.x] = t;
        if (blockIdx.x == 0) { scal[c].r0 = r1; scal[c].iters = k; scal[c].active = 1; }
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the same CG for masks that fit the register file (P <= CUs * 2048 * 10, i.e. up
// to 5.2 M pixels on 256 CUs): one cooperative launch, one block of 512 threads per CU, every thread
// keeps x, r, p and the diagonal of its 4*NV pixels in registers for the whole solve of a channel and
// the two dot products of a step are grid-wide reductions (grid_sum below; every block adds the per-block
// partial sums in the same fixed order, in double). HBM traffic: 16 B per pixel and channel
// instead of 16 + 40 B per CG step. The element-wise arithmetic is that of k_dcg_init / _a / _b.
// ---------------------------------------------------------------------------------------------
struct F4 {
    float e[4];
};
template <int NV, int BT, bool ONE>
__global__ __launch_bounds__(BT) void k_dcg_persistent(float* __restrict__ rho, const float* __restrict__ num,
                                                         const float* __restrict__ den, int P, int C,
                                                         unsigned long long* ent /* [2][gridDim.x], zeroed */,
                                                         unsigned long long* ent3 /* [2][3][gridDim.x], zeroed */,
                                                         DcgScal* __restrict__ scal, float tol2, int max_iter) {
    __shared__ float sm[40];
    const int nb = gridDim.x, tid = threadIdx.x;
    unsigned gen = 0;                  // generations start at 1: the entries are zeroed before the launch
    for (int c = 0; c < C; ++c) {
        const size_t base = (size_t)c * P;
        F4 x[NV], r[NV], p[NV], d[NV];
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const size_t q = ((size_t)(j * nb + blockIdx.x) * BT + tid) * 4;
            if (q < (size_t)P) {
                const Vec<4> vn = ldv<4>(num + base + q), vd = ldv<4>(den + base + q), vx = ldv<4>(rho + base + q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[j].e[e] = vx.v[e]; d[j].e[e] = vd.v[e];
                    r[j].e[e] = vn.v[e] - vd.v[e] * vx.v[e];                      // dc.cu:404-405
                    acc = fmaf(r[j].e[e], r[j].e[e], acc);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[j].e[e] = 0.f; d[j].e[e] = 0.f; r[j].e[e] = 0.f; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) p[j].e[e] = 0.f;
        }
        float r1, r0 = 0.f;
        int k = 0;
        if constexpr (ONE) {
            // One grid-wide wait per step: p.(D p) of the NEXT direction p' = r' + beta p is
            //   r'.D r' + 2 beta r'.D p + beta^2 p.D p,
            // and the first two products can be summed together with r'.r' before beta is known.  r.r, the quantity the
            // stop test looks at, is still summed directly; the solve converges (11 - 15 steps to 1e-9), so the rounding of
            // the predicted p.(D p) (relative 1e-6) does not reach the result.
            float a_rdr = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { p[j].e[e] = r[j].e[e]; a_rdr = fmaf(r[j].e[e], d[j].e[e] * r[j].e[e], a_rdr); }      // k = 1: p = r
            double s_rr, s_pw, s_x;
            ++gen;
            grid_sum3_publish(acc, a_rdr, 0.f, ent3, gen);
            grid_sum3_collect(ent3, gen, s_rr, s_pw, s_x);
            r1 = (float)s_rr;
            double pw = s_pw;
            while (r1 > tol2 && k <= max_iter) {                                  // dc.cu:252
                ++k;
                const float alpha = r1 / (float)pw;
                float a_rr = 0.f, a_rdp = 0.f;
                a_rdr = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float w = d[j].e[e] * p[j].e[e];
                        x[j].e[e] = fmaf(alpha, p[j].e[e], x[j].e[e]);
                        r[j].e[e] = fmaf(-alpha, w, r[j].e[e]);
                        a_rr = fmaf(r[j].e[e], r[j].e[e], a_rr);
                        a_rdr = fmaf(r[j].e[e], d[j].e[e] * r[j].e[e], a_rdr);
                        a_rdp = fmaf(r[j].e[e], w, a_rdp);
                    }
                double s_rdr, s_rdp;
                ++gen;
                grid_sum3_publish(a_rr, a_rdr, a_rdp, ent3, gen);
                grid_sum3_collect(ent3, gen, s_rr, s_rdr, s_rdp);
                r0 = r1;
                r1 = (float)s_rr;
                const float beta = r1 / r0;
                const double t_sq = s_rdr + (double)beta * (double)beta * pw;      // >= |2 beta r.Dp| (Cauchy-Schwarz)
                pw = t_sq + 2.0 * (double)beta * s_rdp;
                float a_pdp = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        p[j].e[e] = scal_then_axpy(beta, p[j].e[e], r[j].e[e]);
                        a_pdp = fmaf(p[j].e[e], d[j].e[e] * p[j].e[e], a_pdp);
                    }
                // guard: when the three terms cancel two digits the product is summed directly (one more wait; every block
                // holds the same numbers, so the decision is uniform)
                if (!(pw > 1e-2 * t_sq) && r1 > tol2) pw = (double)grid_sum(a_pdp, ent, ++gen, sm);
            }
        } else {
        r1 = grid_sum(acc, ent, ++gen, sm);
        while (r1 > tol2 && k <= max_iter) {                                      // dc.cu:252
            ++k;
            const float beta = (k == 1) ? 0.f : r1 / r0;
            acc = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[j].e[e] = (k == 1) ? r[j].e[e] : scal_then_axpy(beta, p[j].e[e], r[j].e[e]);
                    acc = fmaf(p[j].e[e], d[j].e[e] * p[j].e[e], acc);
                }
            const float dot = grid_sum(acc, ent, ++gen, sm);
            const float alpha = r1 / dot;
            acc = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float w = d[j].e[e] * p[j].e[e];
                    x[j].e[e] = fmaf(alpha, p[j].e[e], x[j].e[e]);
                    r[j].e[e] = fmaf(-alpha, w, r[j].e[e]);
                    acc = fmaf(r[j].e[e], r[j].e[e], acc);
                }
            r0 = r1;
            r1 = grid_sum(acc, ent, ++gen, sm);
        }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const size_t q = ((size_t)(j * nb + blockIdx.x) * BT + tid) * 4;
            if (q < (size_t)P) {
                Vec<4> vx;
#pragma unroll
                for (int e = 0; e < 4; ++e) vx.v[e] = x[j].e[e];
                stv<4>(rho + base + q, vx);
            }
        }
        if (blockIdx.x == 0 && tid == 0) { scal[c].r0 = r0; scal[c].iters = k; scal[c].active = (r1 > tol2) ? 1 : 0; }
    }
}

// 0 when the persistent form cannot be used (mask too large for the register file, unaligned arrays)
// blocks of 512 threads (8 waves: cheaper block barriers in the grid-wide sums than 16 waves) with 2, 4 or 8 float4 per
// thread and array; 10 float4 (160 of 256 registers) for masks up to 5.2 M pixels
static int dcg_persistent_plan(srps_ctx* ctx, int P, bool vec, int& NV, int& nb) {
    if (!vec || !ctx->albedo_persistent) return 0;
    const int cus = ctx->num_cus;
    for (int cand : {2, 4, 8, 10}) {
        if ((long long)cand * cus * 2048 >= P) {
            NV = cand;
            nb = cdiv(P, cand * 2048);
            return 1;
        }
    }
    return 0;
}

// after a stream synchronisation: move the counts of the last persistent albedo solve out of the pinned buffer
void albedo_iters_collect(srps_ctx* ctx) {
    if (ctx->albedo_iters_pending <= 0) return;
    const DcgScal* hs = (const DcgScal*)(ctx->h_pinned + 16);
    for (int c = 0; c < ctx->albedo_iters_pending; ++c) ctx->last_albedo_iters[c] = hs[c].iters;
    ctx->albedo_iters_pending = 0;
}

int albedo_finish(srps_ctx* ctx, float* d_rho, const float* d_numden, int P, int C) {
    const float* num = d_numden;
    const float* den = d_numden + (size_t)C * P;
    SRPS_REQUIRE(C <= 8, SRPS_ERR_UNSUPPORTED, "albedo: at most 8 channels");
    if (ctx->albedo_mode == SRPS_ALBEDO_CLOSED_FORM) {
        const size_t n = (size_t)C * P;
        hipLaunchKernelGGL(k_albedo_closed, dim3(std::min(cdiv((long long)n, 256), 4096)), dim3(256), 0, ctx->stream, d_rho, num, den, n);
        SRPS_LAUNCH_CHECK();
        for (int c = 0; c < C; ++c) ctx->last_albedo_iters[c] = 0;
        return SRPS_OK;
    }
    ctx->albedo_iters_pending = 0;
    const int nb = std::max(1, std::min(cdiv(P, 256 * 4), 512));
    const size_t nv = (size_t)C * P;
    const size_t bytes = (2 * nv + (size_t)C * 3 * nb) * sizeof(float) + 8 * sizeof(DcgScal) + (2 + 6) * 1024 * sizeof(unsigned long long) + 256;
    SRPS_TRY(ensure(ctx->ws_albedo, bytes));
    float* r = (float*)ctx->ws_albedo.p;
    float* p = r + nv;
    float* rr_part = p + nv;                    // [C][2][nb]
    float* pw_part = rr_part + (size_t)C * 2 * nb;
    DcgScal* scal = (DcgScal*)(pw_part + (size_t)C * nb);
    const float tol2 = ctx->cg_tol * ctx->cg_tol;
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_rho | (uintptr_t)d_numden | (uintptr_t)r) % 16 == 0);
    DcgScal* hs = (DcgScal*)(ctx->h_pinned + 16);
    int pNV = 0, pnb = 0;
    if (dcg_persistent_plan(ctx, P, vec, pNV, pnb)) {
        // [2][pnb] behind the C <= 8 scalar records, 8-byte aligned
        unsigned long long* ent = (unsigned long long*)(((uintptr_t)(scal + 8) + 7) & ~(uintptr_t)7);
        unsigned long long* ent3 = ent + 2 * 1024;
        SRPS_HIP(hipMemsetAsync(ent, 0, (2 * 1024 + 6 * (size_t)pnb) * sizeof(unsigned long long), ctx->stream));
        float tol2v = tol2;
        int maxit = ctx->cg_max_iter, Pv = P, Cv = C;
        float* rho_v = d_rho;
        void* args[] = {&rho_v, (void*)&num, (void*)&den, &Pv, &Cv, &ent, &ent3, &scal, &tol2v, &maxit};
#define SRPS_DCG(NVV) (ctx->albedo_one_sync ? (const void*)k_dcg_persistent<NVV, 512, true> : (const void*)k_dcg_persistent<NVV, 512, false>)
        const void* fn = pNV == 2 ? SRPS_DCG(2) : pNV == 4 ? SRPS_DCG(4) : pNV == 8 ? SRPS_DCG(8) : SRPS_DCG(10);
#undef SRPS_DCG
        const int lrc = launch_persistent(ctx, fn, pnb, 512, args, 0);
        if (lrc == SRPS_ERR_UNSUPPORTED) ctx->albedo_persistent = 0;     // fall through to the streaming form below
        else {
            SRPS_TRY(lrc);
            // no host synchronisation here: the iteration counts are picked up from the pinned buffer the next time the
            // host waits for the stream anyway (albedo_iters_collect)
            SRPS_HIP(hipMemcpyAsync(hs, scal, C * sizeof(DcgScal), hipMemcpyDeviceToHost, ctx->stream));
            ctx->albedo_iters_pending = C;
            return SRPS_OK;
        }
    }
    if (vec) hipLaunchKernelGGL(k_dcg_init<4>, dim3(nb, C), dim3(256), 0, ctx->stream, d_rho, num, den, P, r, rr_part, nb, scal);
    else hipLaunchKernelGGL(k_dcg_init<1>, dim3(nb, C), dim3(256), 0, ctx->stream, d_rho, num, den, P, r, rr_part, nb, scal);
    SRPS_LAUNCH_CHECK();
    const int kmax = ctx->cg_max_iter + 1;         // "k <= max_iter" => up to max_iter+1 steps (dc.cu:252)
    for (int k = 1; k <= kmax; ++k) {
        if (vec) {
            hipLaunchKernelGGL(k_dcg_a<4>, dim3(nb, C), dim3(256), 0, ctx->stream, k, den, r, p, P, rr_part, pw_part, nb, scal, tol2);
            hipLaunchKernelGGL(k_dcg_b<4>, dim3(nb, C), dim3(256), 0, ctx->stream, k, den, r, p, d_rho, P, rr_part, pw_part, nb, scal, tol2);
        } else {
            hipLaunchKernelGGL(k_dcg_a<1>, dim3(nb, C), dim3(256), 0, ctx->stream, k, den, r, p, P, rr_part, pw_part, nb, scal, tol2);
            hipLaunchKernelGGL(k_dcg_b<1>, dim3(nb, C), dim3(256), 0, ctx->stream, k, den, r, p, d_rho, P, rr_part, pw_part, nb, scal, tol2);
        }
        if ((k % 8) == 0 || k == kmax) {
            SRPS_LAUNCH_CHECK();
            SRPS_HIP(hipMemcpyAsync(hs, scal, C * sizeof(DcgScal), hipMemcpyDeviceToHost, ctx->stream));
            SRPS_HIP(hipStreamSynchronize(ctx->stream));
            bool any = false;
            for (int c = 0; c < C; ++c) any |= (hs[c].active != 0) && (hs[c].iters == k);
            if (!any) break;
        }
    }
    SRPS_HIP(hipMemcpyAsync(hs, scal, C * sizeof(DcgScal), hipMemcpyDeviceToHost, ctx->stream));
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < C; ++c) ctx->last_albedo_iters[c] = hs[c].iters;
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
                                                        float* __restrict__ M, float* __restrict__ Q, float* __restrict__ Gp) {
    const int q = (blockIdx.x * 256 + threadIdx.x) * V;
    if (q >= P) return;
    const Vec<V> vdz = ldv<V>(dz + q), vxx = ldv<V>(xx + q), vyy = ldv<V>(yy + q);
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
#pragma unroll
            for (int e = 0; e < V; ++e) Gp[(size_t)c * plane + gofp[q + e]] = g[e] * g[e];
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
            const Vec<V> iv = ldv<V>(I + ((size_t)li * C + c) * P + q);
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
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int go = gofp[q + e];
        if (M) {
#pragma unroll
            for (int t = 0; t < 6; ++t) M[(size_t)t * plane + go] = m[t][e];
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) Q[(size_t)t * plane + go] = qq[t][e];
    }
}

// Per-channel constants of the tensor-recompute form.  With a_i = fx s_i0, a'_i = fy s_i1, b_i = s_i2
// (image i, channel c) and sums S.. over the images:
//   sum_i t_i t_i' ,  t_i = (a_i - xx b_i, a'_i - yy b_i, -b_i)
//     = [[Sbb dx^2 + R00, Sbb dx dy + R01, Sbb dx], [., Sbb dy^2 + R11, Sbb dy], [., ., Sbb]]
//   dx = xx - Sab/Sbb, dy = yy - Sa'b/Sbb, R00 = Saa - Sab^2/Sbb, R01 = Saa' - Sab Sa'b/Sbb, R11 = Sa'a' - Sa'b^2/Sbb
// (completed squares: every term is non-negative, no cancellation).  Evaluated in double.
__global__ void k_tensor_consts(const float* __restrict__ s, int n_total, int C, float fx, float fy, float* __restrict__ out) {
    const int c = threadIdx.x;
    if (c >= C) return;
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

int depth_assemble(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_I, const float* d_xx,
                   const float* d_yy, const float* d_dz, float fx, float fy, int P, int n_local, int C,
                   int n_total, int img_offset, float cx, float cy) {
    Grid& G = ctx->grid;
    // tensor-recompute form needs the principal point (xx = j - cx, yy = i - cy are rebuilt in the kernel)
    const bool rec = ctx->tensor_recompute && (C == 1 || C == 3) && cx == cx && cy == cy;
    float* Gp = nullptr;
    if (rec) {
        if (G.G_planes < (size_t)C) {
            if (G.d_G) SRPS_HIP(hipFree(G.d_G));
            G.d_G = nullptr; G.G_planes = 0;
            SRPS_HIP(hipMalloc((void**)&G.d_G, (size_t)C * G.plane * sizeof(float)));
            SRPS_HIP(hipMemsetAsync(G.d_G, 0, (size_t)C * G.plane * sizeof(float), ctx->stream));     // zero outside the mask
            G.G_planes = C;
        }
        Gp = G.d_G;
        hipLaunchKernelGGL(k_tensor_consts, dim3(1), dim3(64), 0, ctx->stream, d_s, n_total, C, fx, fy, G.d_tconsts);
        G.tensor_channels = C; G.cx = cx; G.cy = cy;
    } else {
        G.tensor_channels = 0;
    }
    // the stored tensor is only needed by the simple operator kernel and by the stored-tensor marching form
    float* Mp = (rec && use_march(ctx) && !ctx->keep_stored_tensor) ? nullptr : G.d_M;
    G.M_valid = Mp != nullptr;
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_rho | (uintptr_t)d_I | (uintptr_t)d_xx | (uintptr_t)d_yy | (uintptr_t)d_dz) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL((k_depth_assemble<4>), dim3(cdiv(P, 1024)), dim3(256), 0, ctx->stream, d_s, d_rho, d_I, d_xx, d_yy, d_dz,
                           fx, fy, P, n_local, C, n_total, img_offset, G.d_gofp, G.plane, Mp, G.d_q, Gp);
    else
        hipLaunchKernelGGL((k_depth_assemble<1>), dim3(cdiv(P, 256)), dim3(256), 0, ctx->stream, d_s, d_rho, d_I, d_xx, d_yy, d_dz,
                           fx, fy, P, n_local, C, n_total, img_offset, G.d_gofp, G.plane, Mp, G.d_q, Gp);
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
                const Vec<V> iv = ldv<V>(I + ((size_t)li * C + c) * P + q);
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
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, ctx->stream, part, nb, d_out);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

}  // namespace srps
