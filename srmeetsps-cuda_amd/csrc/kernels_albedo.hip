// kernels_albedo.hip -- albedo estimation (devicecalls.cu:447-548): the per-pixel sums over the images, then the reference's
// CG on the diagonal system: persistent (in registers, one launch) for masks up to 5.2 M pixels, one kernel per half step
// otherwise, or the fixed point num / den.
#include "srps_internal.h"
#include "device_utils.h"

namespace srps {

// =============================================================================================
// albedo (reference: devicecalls.cu:447-548)
//   sh_i[p] = N[:,p] . s_ic  (sgemm dc.cu:507);  num = sum_i sh_i I_ic, den = sum_i sh_i^2
//   (the diagonal A'A and A'b of dc.cu:395-406);  then the reference's global CG on the
//   diagonal system from the warm start rho_c (dc.cu:540), or its fixed point num/den.
// =============================================================================================
// SUMS: the same sweep also leaves, per channel, the three image sums the depth right-hand side is made of,
//   SA = sum_i (fx s_i0) I_i,  SA' = sum_i (fy s_i1) I_i,  SB = sum_i s_i2 I_i        (ssum[c][3][P]):
//   q = sum_{c,i} g tau_i (I_i - rho s_i3), tau_i = (fx s_i0 - xx s_i2, fy s_i1 - yy s_i2, -s_i2)      (dc.cu:588-610)
//     = sum_c g [ (SA - xx SB, SA' - yy SB, -SB) - rho (CA - xx CB, CA' - yy CB, -CB) ],  C. = the same sums of s_i3,
// so that the depth assembly after the albedo step needs no second pass over I (k_depth_from_sums).
// sh = N . s with every multiply-add spelled out: the three loops of the sweep below (and the ranks of a sharded run, which run
// different ones of them for the same image) must produce the same bits
__device__ __forceinline__ float shading(float n0, float n1, float n2, float n3, float s0, float s1, float s2, float s3) {
    return __builtin_fmaf(n3, s3, __builtin_fmaf(n2, s2, __builtin_fmaf(n1, s1, n0 * s0)));
}

// SHARD (a context that holds images [s_img_offset, s_img_offset + n_local) of n_total): den = sum_i sh_i^2 does not involve the
// images -- it is formed over ALL images here, in the single-GPU order (the same bits on every rank and as on one GPU), and only
// num travels through the all-reduce (C P floats instead of 2 C P).
template <int V, bool SUMS, bool U8 = false, bool SHARD = false>
__global__ __launch_bounds__(256) void k_albedo_numden(const float* __restrict__ s, const float* __restrict__ N,
                                                       const float* __restrict__ I, const unsigned char* __restrict__ I8, int P, int n_local, int C,
                                                       int s_img_offset, float* __restrict__ num, float* __restrict__ den,
                                                       float fx, float fy, float* __restrict__ ssum, int n_total, int blk0) {
    const int q = ((blockIdx.x + blk0) * 256 + threadIdx.x) * V;      // blk0: a launch that covers a range of the pixels only (overlapped exchange)
    if (q >= P) return;
    Vec<V> nk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) nk[k] = ldv<V>(N + (size_t)k * P + q);
    for (int c = 0; c < C; ++c) {
        Vec<V> nu, de, sa, sap, sb;
#pragma unroll
        for (int e = 0; e < V; ++e) { nu.v[e] = 0.f; de.v[e] = 0.f; sa.v[e] = 0.f; sap.v[e] = 0.f; sb.v[e] = 0.f; }
        if (SHARD) {                                                             // images before this rank's: den only
            for (int i = 0; i < s_img_offset; ++i) {
                const float* sv = s + ((size_t)i * C + c) * 4;
                const float s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float sh = shading(nk[0].v[e], nk[1].v[e], nk[2].v[e], nk[3].v[e], s0, s1, s2, s3);
                    de.v[e] = fmaf(sh, sh, de.v[e]);
                }
            }
        }
#pragma unroll 4
        for (int i = 0; i < n_local; ++i) {
            const float* sv = s + ((size_t)(s_img_offset + i) * C + c) * 4;      // uniform -> scalar loads
            const float s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
            const float fs0 = fx * s0, fs1 = fy * s1;
            const Vec<V> iv = ld_img<V, U8>(I, I8, (size_t)i * C + c, P, q, n_local * C);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float sh = shading(nk[0].v[e], nk[1].v[e], nk[2].v[e], nk[3].v[e], s0, s1, s2, s3);
                nu.v[e] = fmaf(sh, iv.v[e], nu.v[e]);
                de.v[e] = fmaf(sh, sh, de.v[e]);
                if (SUMS) {
                    sa.v[e] = fmaf(fs0, iv.v[e], sa.v[e]);
                    sap.v[e] = fmaf(fs1, iv.v[e], sap.v[e]);
                    sb.v[e] = fmaf(s2, iv.v[e], sb.v[e]);
                }
            }
        }
        if (SHARD) {                                                             // images after this rank's
            for (int i = s_img_offset + n_local; i < n_total; ++i) {
                const float* sv = s + ((size_t)i * C + c) * 4;
                const float s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float sh = shading(nk[0].v[e], nk[1].v[e], nk[2].v[e], nk[3].v[e], s0, s1, s2, s3);
                    de.v[e] = fmaf(sh, sh, de.v[e]);
                }
            }
        }
        stv_stream<V>(num + (size_t)c * P + q, nu);
        stv_stream<V>(den + (size_t)c * P + q, de);
        if (SUMS) {
            stv_stream<V>(ssum + ((size_t)c * 3 + 0) * P + q, sa);
            stv_stream<V>(ssum + ((size_t)c * 3 + 1) * P + q, sap);
            stv_stream<V>(ssum + ((size_t)c * 3 + 2) * P + q, sb);
        }
    }
}

// SRPS_ALBEDO_FUSED: the albedo's fixed point AND the depth system in the one sweep over the images.  Per pixel and channel the
// sweep has num, den and the three image sums in registers anyway; rho = num / den (k_albedo_closed), g = (rho / dz)^2 and the
// right-hand side q (k_depth_from_sums, the same expressions in the same order: the same bits as the unfused closed form) follow
// without num, den and the nine sum planes ever being stored -- 250 MB less written, 250 MB less read, two kernels fewer per pass.
#ifndef SRPS_ALBEDO_UNROLL
#define SRPS_ALBEDO_UNROLL 4      // image loads in flight per thread (round 5 swept 1 .. 10: flat between 2 and 4, slower outside; profiles/r05_ab_albedo_sweep.jsonl)
#endif
template <int V, bool U8>
__global__ __launch_bounds__(256) void k_albedo_fused(const float* __restrict__ s, const float* __restrict__ N, const float* __restrict__ I,
                                                      const unsigned char* __restrict__ I8, int P, int n_img, int C, float* __restrict__ rho,
                                                      const float* __restrict__ qc, const float* __restrict__ xx, const float* __restrict__ yy,
                                                      const float* __restrict__ dz, float fx, float fy, const int* __restrict__ gofp, size_t plane,
                                                      float* __restrict__ Q, float* __restrict__ Gp, int n3_one) {
    const int q = (blockIdx.x * 256 + threadIdx.x) * V;
    if (q >= P) return;
    Vec<V> nk[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) nk[k] = ldv<V>(N + (size_t)k * P + q);
    // N3 == 1 (dc.cu:175): the pipeline's own normals hold a plane of ones there, which is then not read (kernel-uniform; 1.f * s3 is exact:
    // the same bits) -- 16.8 MB less per sweep at 2048^2.  A caller's array (srps_set / a pointer handed out) is read as it is.
    if (n3_one) {
#pragma unroll
        for (int e = 0; e < V; ++e) nk[3].v[e] = 1.f;
    } else nk[3] = ldv<V>(N + (size_t)3 * P + q);
    const Vec<V> vdz = ldv<V>(dz + q), vxx = ldv<V>(xx + q), vyy = ldv<V>(yy + q);
    const GridIdx<V> gi = grid_idx<V>(gofp, q);
    float qq[3][V];
#pragma unroll
    for (int e = 0; e < V; ++e) qq[0][e] = qq[1][e] = qq[2][e] = 0.f;
    for (int c = 0; c < C; ++c) {
        Vec<V> nu, de, sa, sap, sb;
#pragma unroll
        for (int e = 0; e < V; ++e) { nu.v[e] = 0.f; de.v[e] = 0.f; sa.v[e] = 0.f; sap.v[e] = 0.f; sb.v[e] = 0.f; }
#pragma unroll SRPS_ALBEDO_UNROLL
        for (int i = 0; i < n_img; ++i) {                                        // k_albedo_numden<V, true>: the same loop
            const float* sv = s + ((size_t)i * C + c) * 4;
            const float s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
            const float fs0 = fx * s0, fs1 = fy * s1;
            const Vec<V> iv = ld_img<V, U8>(I, I8, (size_t)i * C + c, P, q, n_img * C);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float sh = shading(nk[0].v[e], nk[1].v[e], nk[2].v[e], nk[3].v[e], s0, s1, s2, s3);
                nu.v[e] = fmaf(sh, iv.v[e], nu.v[e]);
                de.v[e] = fmaf(sh, sh, de.v[e]);
                sa.v[e] = fmaf(fs0, iv.v[e], sa.v[e]);
                sap.v[e] = fmaf(fs1, iv.v[e], sap.v[e]);
                sb.v[e] = fmaf(s2, iv.v[e], sb.v[e]);
            }
        }
        // k_albedo_closed: pixels with a zero denominator keep their value -- the old albedo is read only by a thread that has such a pixel
        // (no image lights it: rare; 50 MB less read per sweep at 2048^2 x 3 channels)
        bool keep = false;
#pragma unroll
        for (int e = 0; e < V; ++e) keep = keep || !(de.v[e] > 0.f);
        Vec<V> vr;
        if (keep) vr = ldv<V>(rho + (size_t)c * P + q);
#pragma unroll
        for (int e = 0; e < V; ++e)
            if (de.v[e] > 0.f) vr.v[e] = nu.v[e] / de.v[e];
        stv<V>(rho + (size_t)c * P + q, vr);
        const float ca = qc[c * 4 + 0], cap = qc[c * 4 + 1], cb = qc[c * 4 + 2];
        float g[V], g2[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { g[e] = vr.v[e] / vdz.v[e]; g2[e] = g[e] * g[e]; }
        scatter_store<V>(Gp + (size_t)c * plane, gi, g2);
#pragma unroll
        for (int e = 0; e < V; ++e) {                                            // k_depth_from_sums: the same expressions
            const float t0 = fmaf(-vxx.v[e], sb.v[e], sa.v[e]), t1 = fmaf(-vyy.v[e], sb.v[e], sap.v[e]), t2 = -sb.v[e];
            const float u0 = fmaf(-vxx.v[e], cb, ca), u1 = fmaf(-vyy.v[e], cb, cap), u2 = -cb;
            qq[0][e] = fmaf(g[e], fmaf(-vr.v[e], u0, t0), qq[0][e]);
            qq[1][e] = fmaf(g[e], fmaf(-vr.v[e], u1, t1), qq[1][e]);
            qq[2][e] = fmaf(g[e], fmaf(-vr.v[e], u2, t2), qq[2][e]);
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) scatter_store<V>(Q + (size_t)t * plane, gi, qq[t]);
}
int albedo_fused(srps_ctx* ctx, const float* d_s, const float* d_N, const float* d_I, int P, int n_img, int C, float* d_rho, const float* d_qc,
                 const float* d_xx, const float* d_yy, const float* d_dz, float fx, float fy) {
    Grid& G = ctx->grid;
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_N | (uintptr_t)d_I | (uintptr_t)d_rho | (uintptr_t)d_xx | (uintptr_t)d_yy | (uintptr_t)d_dz) % 16 == 0);
    const unsigned char* d_I8 = vec ? image_store_bytes(ctx, d_I) : nullptr;
#define SRPS_AF(VV, UU, IMG) hipLaunchKernelGGL((k_albedo_fused<VV, UU>), dim3(cdiv(P, 256 * VV)), dim3(256), 0, ctx->stream, d_s, d_N, IMG, d_I8, P, n_img, C, d_rho, d_qc, \
                                           d_xx, d_yy, d_dz, fx, fy, G.d_gofp, G.plane, G.d_q, G.d_G, (ctx->n3_one && d_N == ctx->Nrm) ? 1 : 0)
    if (vec && d_I8) SRPS_AF(4, true, d_I);
    else if (vec) SRPS_AF(4, false, d_I);
    else SRPS_AF(1, false, d_I);
#undef SRPS_AF
    SRPS_LAUNCH_CHECK();
    for (int c = 0; c < C; ++c) ctx->last_albedo_iters[c] = 0;
    return SRPS_OK;
}

// ssum != null: also the image sums of the depth right-hand side (fx, fy needed).  n_total > n_local: the context holds a shard
// (see SHARD above): d_numden receives this rank's part of num and the complete den.
int albedo_numden(srps_ctx* ctx, const float* d_s, const float* d_N, const float* d_I, int P, int n_local,
                  int C, int s_img_offset, float* d_numden, float fx, float fy, float* d_ssum, int n_total, int q0, int q1) {
    float* num = d_numden;
    float* den = d_numden + (size_t)C * P;
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_N | (uintptr_t)d_I | (uintptr_t)d_numden | (uintptr_t)d_ssum) % 16 == 0);
    const unsigned char* d_I8 = vec ? image_store_bytes(ctx, d_I) : nullptr;      // the context's images as bytes, when they are held that way
    const bool shard = n_total > n_local;
    // [q0, q1): the pixels of this launch (q1 <= 0: all; q0 a multiple of 1024 -- whole blocks of either vector width)
    const bool part = q1 > 0;
    auto blocks = [&](int per) { return part ? cdiv(std::min(q1, P) - q0, per) : cdiv(P, per); };
    auto first = [&](int per) { return part ? q0 / per : 0; };
#define SRPS_NUMDEN(VV, SS, UU, HH, IMG) hipLaunchKernelGGL((k_albedo_numden<VV, SS, UU, HH>), dim3(blocks(256 * VV)), dim3(256), 0, ctx->stream, d_s, d_N, IMG, d_I8, P, n_local, C, s_img_offset, num, den, fx, fy, d_ssum, n_total, first(256 * VV))
#define SRPS_NUMDEN_S(VV, UU, HH, IMG) do { if (d_ssum) SRPS_NUMDEN(VV, true, UU, HH, IMG); else SRPS_NUMDEN(VV, false, UU, HH, IMG); } while (0)
    if (shard) {
        if (vec && d_I8) SRPS_NUMDEN_S(4, true, true, d_I);
        else if (vec) SRPS_NUMDEN_S(4, false, true, d_I);
        else SRPS_NUMDEN_S(1, false, true, d_I);
    } else {
        if (vec && d_I8) SRPS_NUMDEN_S(4, true, false, d_I);
        else if (vec) SRPS_NUMDEN_S(4, false, false, d_I);
        else SRPS_NUMDEN_S(1, false, false, d_I);
    }
#undef SRPS_NUMDEN_S
#undef SRPS_NUMDEN
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
                                                         unsigned long long* ent3 /* [2][256] 16-byte granules, zeroed */,
                                                         DcgScal* __restrict__ scal, float tol2, int max_iter,
                                                         int* status /* CgScalars::abort_flags.. */, unsigned long long spin_ticks) {
    __shared__ float sm[40];
    const int nb = gridDim.x, tid = threadIdx.x;
    unsigned gen = 0;                  // generations start at 1: the entries are zeroed before the launch
    spin_guard_init(spin_ticks, status, ABORT_ALBEDO);      // bounded waits: device_utils.h
    __syncthreads();
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
            grid_sum3_publish<BT / 64>(acc, a_rdr, 0.f, ent3, gen);
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
                grid_sum3_publish<BT / 64>(a_rr, a_rdr, a_rdp, ent3, gen);
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
        if (spin_block_dead()) break;          // a wait gave up: this channel (and the ones after it) keep their albedo
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

// The three colour channels advanced together (C == 3, masks that leave room for 3 x 4 arrays in the register file):
// the channels are independent solves, so their steps can share the grid-wide wait -- one exchange of three granules
// per block and step instead of one per channel and step; a solve then waits as often as its slowest channel iterates
// (11 - 15 times) instead of the sum over the channels.  Per channel the arithmetic, the order of every sum and the stop
// test are those of k_dcg_persistent<NV, BT, true>: same bits.
template <int NV, int BT>
__global__ __launch_bounds__(BT) void k_dcg_persistent3(float* __restrict__ rho, const float* __restrict__ num,
                                                          const float* __restrict__ den, int P,
                                                          unsigned long long* ent /* [2][gridDim.x], zeroed */,
                                                          unsigned long long* ent9 /* [2][3][256] 16-byte granules, zeroed */,
                                                          DcgScal* __restrict__ scal, float tol2, int max_iter,
                                                          int* status /* CgScalars::abort_flags.. */, unsigned long long spin_ticks) {
    __shared__ float sm[40];
    const int nb = gridDim.x, tid = threadIdx.x;
    unsigned gen = 0, gen1 = 0;        // generations of the nine-value and of the single-value exchange
    spin_guard_init(spin_ticks, status, ABORT_ALBEDO);      // bounded waits: device_utils.h
    __syncthreads();
    F4 x[3][NV], r[3][NV], p[3][NV], d[3][NV];
    float part[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t base = (size_t)c * P;
        float acc = 0.f, a_rdr = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const size_t q = ((size_t)(j * nb + blockIdx.x) * BT + tid) * 4;
            if (q < (size_t)P) {
                const Vec<4> vn = ldv<4>(num + base + q), vd = ldv<4>(den + base + q), vx = ldv<4>(rho + base + q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[c][j].e[e] = vx.v[e]; d[c][j].e[e] = vd.v[e];
                    r[c][j].e[e] = vn.v[e] - vd.v[e] * vx.v[e];                   // dc.cu:404-405
                    acc = fmaf(r[c][j].e[e], r[c][j].e[e], acc);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[c][j].e[e] = 0.f; d[c][j].e[e] = 0.f; r[c][j].e[e] = 0.f; }
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { p[c][j].e[e] = r[c][j].e[e]; a_rdr = fmaf(r[c][j].e[e], d[c][j].e[e] * r[c][j].e[e], a_rdr); }   // k = 1: p = r
        part[c][0] = acc; part[c][1] = a_rdr; part[c][2] = 0.f;
    }
    double tot[3][3];
    ++gen;
    grid_sum9_publish<BT / 64>(part, ent9, gen);
    grid_sum9_collect(ent9, gen, tot);
    float r1[3], r0[3] = {0.f, 0.f, 0.f};
    double pw[3];
    int k[3] = {0, 0, 0};
    bool act[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { r1[c] = (float)tot[c][0]; pw[c] = tot[c][1]; act[c] = r1[c] > tol2 && k[c] <= max_iter; }
    while (act[0] || act[1] || act[2]) {                                          // dc.cu:252, per channel
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            part[c][0] = 0.f; part[c][1] = 0.f; part[c][2] = 0.f;
            if (act[c]) {
                ++k[c];
                const float alpha = r1[c] / (float)pw[c];
                float a_rr = 0.f, a_rdr = 0.f, a_rdp = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float w = d[c][j].e[e] * p[c][j].e[e];
                        x[c][j].e[e] = fmaf(alpha, p[c][j].e[e], x[c][j].e[e]);
                        r[c][j].e[e] = fmaf(-alpha, w, r[c][j].e[e]);
                        a_rr = fmaf(r[c][j].e[e], r[c][j].e[e], a_rr);
                        a_rdr = fmaf(r[c][j].e[e], d[c][j].e[e] * r[c][j].e[e], a_rdr);
                        a_rdp = fmaf(r[c][j].e[e], w, a_rdp);
                    }
                part[c][0] = a_rr; part[c][1] = a_rdr; part[c][2] = a_rdp;
            }
        }
        ++gen;
        grid_sum9_publish<BT / 64>(part, ent9, gen);
        grid_sum9_collect(ent9, gen, tot);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (act[c]) {
                r0[c] = r1[c];
                r1[c] = (float)tot[c][0];
                const float beta = r1[c] / r0[c];
                const double t_sq = tot[c][1] + (double)beta * (double)beta * pw[c];      // >= |2 beta r.Dp| (Cauchy-Schwarz)
                pw[c] = t_sq + 2.0 * (double)beta * tot[c][2];
                float a_pdp = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        p[c][j].e[e] = scal_then_axpy(beta, p[c][j].e[e], r[c][j].e[e]);
                        a_pdp = fmaf(p[c][j].e[e], d[c][j].e[e] * p[c][j].e[e], a_pdp);
                    }
                // the same guard as in the one-channel form; every block holds the same numbers, so the decision is uniform
                if (!(pw[c] > 1e-2 * t_sq) && r1[c] > tol2) pw[c] = (double)grid_sum(a_pdp, ent, ++gen1, sm);
                act[c] = r1[c] > tol2 && k[c] <= max_iter;
            }
        }
    }
    if (spin_block_dead()) return;             // a wait gave up: the albedo stays as it was
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t base = (size_t)c * P;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const size_t q = ((size_t)(j * nb + blockIdx.x) * BT + tid) * 4;
            if (q < (size_t)P) {
                Vec<4> vx;
#pragma unroll
                for (int e = 0; e < 4; ++e) vx.v[e] = x[c][j].e[e];
                stv<4>(rho + base + q, vx);
            }
        }
        if (blockIdx.x == 0 && tid == 0) { scal[c].r0 = r0[c]; scal[c].iters = k[c]; scal[c].active = (r1[c] > tol2) ? 1 : 0; }
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

int albedo_finish(srps_ctx* ctx, float* d_rho, const float* d_numden, int P, int C, bool pipeline) {
    const float* num = d_numden;
    const float* den = d_numden + (size_t)C * P;
    SRPS_REQUIRE(C <= 8, SRPS_ERR_UNSUPPORTED, "albedo: at most 8 channels");
    if (ctx->albedo_mode == SRPS_ALBEDO_CLOSED_FORM || ctx->albedo_mode == SRPS_ALBEDO_FUSED ||       // FUSED where the fused sweep did not apply
        (ctx->albedo_mode == SRPS_ALBEDO_AUTO && pipeline)) {
        const size_t n = (size_t)C * P;
        hipLaunchKernelGGL(k_albedo_closed, dim3(std::min(cdiv((long long)n, 256), 4096)), dim3(256), 0, ctx->stream, d_rho, num, den, n);
        SRPS_LAUNCH_CHECK();
        for (int c = 0; c < C; ++c) ctx->last_albedo_iters[c] = 0;
        return SRPS_OK;
    }
    ctx->albedo_iters_pending = 0;
    const int nb = std::max(1, std::min(cdiv(P, 256 * 4), 512));
    const size_t nv = (size_t)C * P;
    static_assert(SRPS_G3_REPLICAS <= 3, "the granule area below holds three triples (grid_sum9) or SRPS_G3_REPLICAS copies of one (grid_sum3)");
    const size_t bytes = (2 * nv + (size_t)C * 3 * nb) * sizeof(float) + 8 * sizeof(DcgScal) + 2 * 1024 * sizeof(unsigned long long) + 3 * 2 * 1024 * SRPS_G3_STRIDE + 256;
    SRPS_TRY(ensure(ctx->ws_albedo, bytes));
    float* r = (float*)ctx->ws_albedo.p;
    float* p = r + nv;
    float* rr_part = p + nv;                    // [C][2][nb]
    float* pw_part = rr_part + (size_t)C * 2 * nb;
    DcgScal* scal = (DcgScal*)(ctx->d_report + 16);             // [8] records, part of the context's report record
    float* after_parts = pw_part + (size_t)C * nb;
    const float tol2 = ctx->cg_tol * ctx->cg_tol;
    const bool vec = (P % 4 == 0) && (((uintptr_t)d_rho | (uintptr_t)d_numden | (uintptr_t)r) % 16 == 0);
    DcgScal* hs = (DcgScal*)(ctx->h_pinned + 16);
    int pNV = 0, pnb = 0;
    if (dcg_persistent_plan(ctx, P, vec, pNV, pnb)) {
        // [2][pnb] behind the C <= 8 scalar records, 16-byte aligned (ent3 holds 16-byte granules, [2][256])
        unsigned long long* ent = (unsigned long long*)(((uintptr_t)after_parts + 15) & ~(uintptr_t)15);
        unsigned long long* ent3 = ent + 2 * 1024;
        SRPS_HIP(hipMemsetAsync(ent, 0, (2 * 1024 + 3 * 2 * (SRPS_G3_STRIDE / 8) * (size_t)((pnb + 255) & ~255)) * sizeof(unsigned long long), ctx->stream));
        float tol2v = tol2;
        int maxit = ctx->cg_max_iter, Pv = P, Cv = C;
        float* rho_v = d_rho;
        int* status = (int*)(ctx->d_report + 64) + 5;                         // CgScalars::abort_flags of the report record
        unsigned long long spin_ticks = (unsigned long long)ctx->spin_budget_ms * 100000ull;
        void* args[] = {&rho_v, (void*)&num, (void*)&den, &Pv, &Cv, &ent, &ent3, &scal, &tol2v, &maxit, &status, &spin_ticks};
#define SRPS_DCG(NVV) (ctx->albedo_one_sync ? (const void*)k_dcg_persistent<NVV, 512, true> : (const void*)k_dcg_persistent<NVV, 512, false>)
        const void* fn = pNV == 2 ? SRPS_DCG(2) : pNV == 4 ? SRPS_DCG(4) : pNV == 8 ? SRPS_DCG(8) : SRPS_DCG(10);
#undef SRPS_DCG
        // three channels of a mask that leaves room for them in the register file: one launch, shared waits
        const bool together = C == 3 && ctx->albedo_one_sync && ctx->albedo_channels_together && pNV <= 2;
        void* args3[] = {&rho_v, (void*)&num, (void*)&den, &Pv, &ent, &ent3, &scal, &tol2v, &maxit, &status, &spin_ticks};
        if (together) fn = (const void*)k_dcg_persistent3<2, 512>;
        const int lrc = launch_persistent(ctx, fn, pnb, 512, together ? args3 : args, 0);
        if (lrc == SRPS_ERR_UNSUPPORTED) ctx->albedo_persistent = 0;     // fall through to the streaming form below
        else {
            SRPS_TRY(lrc);
            // no host synchronisation here: the iteration counts are picked up from the pinned buffer the next time the
            // host waits for the stream anyway (albedo_iters_collect)
            ctx->report_pending = true;
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

}  // namespace srps
