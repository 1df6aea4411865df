// kernels_cg.hip -- the depth solve on the bounding-box grid: scatter/gather between the compact
// (reference) layout and the grid, masked gradients, right-hand side, the matrix-free operator
//      A_ x = KT'(KT x) + lambda * ( Dx' u + Dy' v + w ),   (u,v,w) = M (Dx x, Dy x, x)
// (matrix-free form of devicecalls.cu:668-736) and the reference's CG recurrence
// (devicecalls.cu:229-279) with device-resident scalars.
//
// Grid layout: plane[(j - j_lo + PAD) * Hs + (i - i_lo + PAD)], zero outside the mask; a
// structure byte per pixel (F_* in srps_internal.h) encodes the rows of Dx, Dy and KT.
#include "srps_internal.h"
#include "device_utils.h"

namespace srps {

// ---------------------------------------------------------------------------------------------
// layout conversion
// ---------------------------------------------------------------------------------------------
// four consecutive compact pixels per thread: one 16-byte access on the grid side wherever they lie in one column segment
// (device_utils.h grid_idx), which is nearly everywhere
template <int V>
__global__ void k_scatter(const float* __restrict__ compact, const int* __restrict__ gofp, int P, float* __restrict__ plane) {
    for (int p = (blockIdx.x * blockDim.x + threadIdx.x) * V; p < P; p += gridDim.x * blockDim.x * V) {
        const GridIdx<V> gi = grid_idx<V>(gofp, p);
        const Vec<V> v = ldv<V>(compact + p);
        scatter_store<V>(plane, gi, v.v);
    }
}
template <int V>
__global__ void k_gather(const float* __restrict__ plane, const int* __restrict__ gofp, int P, float* __restrict__ compact) {
    for (int p = (blockIdx.x * blockDim.x + threadIdx.x) * V; p < P; p += gridDim.x * blockDim.x * V) {
        const GridIdx<V> gi = grid_idx<V>(gofp, p);
        Vec<V> v;
        if (V == 4 && gi.vec) v = ldv<V>(plane + gi.go[0]);
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) v.v[e] = plane[gi.go[e]];
        }
        stv<V>(compact + p, v);
    }
}
int grid_scatter(srps_ctx* ctx, const float* d_compact, float* d_plane) {
    Grid& G = ctx->grid;
    if (G.P % 4 == 0 && (uintptr_t)d_compact % 16 == 0)
        hipLaunchKernelGGL(k_scatter<4>, dim3(std::min(cdiv(G.P, 1024), 4096)), dim3(256), 0, ctx->stream, d_compact, G.d_gofp, G.P, d_plane);
    else
        hipLaunchKernelGGL(k_scatter<1>, dim3(std::min(cdiv(G.P, 256), 4096)), dim3(256), 0, ctx->stream, d_compact, G.d_gofp, G.P, d_plane);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}
int grid_gather(srps_ctx* ctx, const float* d_plane, float* d_compact) {
    Grid& G = ctx->grid;
    if (G.P % 4 == 0 && (uintptr_t)d_compact % 16 == 0)
        hipLaunchKernelGGL(k_gather<4>, dim3(std::min(cdiv(G.P, 1024), 4096)), dim3(256), 0, ctx->stream, d_plane, G.d_gofp, G.P, d_compact);
    else
        hipLaunchKernelGGL(k_gather<1>, dim3(std::min(cdiv(G.P, 256), 4096)), dim3(256), 0, ctx->stream, d_plane, G.d_gofp, G.P, d_compact);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// ---------------------------------------------------------------------------------------------
// zx = Dx z, zy = Dy z  (rows of make_gradient, SRPS.cu:29-47) -> compact outputs
// ---------------------------------------------------------------------------------------------
__global__ void k_gradient(const float* __restrict__ zg, const uint8_t* __restrict__ flags, const int* __restrict__ gofp,
                           int P, int Hs, float* __restrict__ zx, float* __restrict__ zy, float* __restrict__ zc /* may be null */) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int g = gofp[p];
        const uint8_t fl = flags[g];
        const float xc = zg[g];
        if (zc) zc[p] = xc;                                 // the gather of the plane into the compact layout, in the same pass
        float gx = 0.f, gy = 0.f;
        if (fl & F_FX) gx = zg[g + Hs] - xc; else if (fl & F_BX) gx = xc - zg[g - Hs];
        if (fl & F_FY) gy = zg[g + 1] - xc; else if (fl & F_BY) gy = xc - zg[g - 1];
        zx[p] = gx;
        zy[p] = gy;
    }
}
// Four masked pixels per thread where they are four consecutive rows of one column starting at a multiple of four (every group of a
// full-frame mask; most groups inside an object): 16-byte loads of the index, the plane and its two neighbour columns, 16-byte
// stores.  The differences are those of k_gradient, pixel by pixel (which also serves the groups that are not of that kind).
__global__ __launch_bounds__(256) void k_gradient4(const float* __restrict__ zg, const uint8_t* __restrict__ flags, const int* __restrict__ gofp,
                                                   int P, int Hs, float* __restrict__ zx, float* __restrict__ zy, float* __restrict__ zc /* may be null */) {
    const int n4 = P >> 2;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += gridDim.x * blockDim.x) {
        const int4 g4 = reinterpret_cast<const int4*>(gofp)[q];
        float c[4], gx[4], gy[4];
        if (g4.w - g4.x == 3 && (g4.x & 3) == 0) {
            const int g = g4.x;
            const unsigned fw = *reinterpret_cast<const unsigned*>(flags + g);
            const float4 xc = *reinterpret_cast<const float4*>(zg + g);
            c[0] = xc.x; c[1] = xc.y; c[2] = xc.z; c[3] = xc.w;
            float r[4] = {0.f, 0.f, 0.f, 0.f}, l[4] = {0.f, 0.f, 0.f, 0.f};
            if (fw & (0x01010101u * F_FX)) { const float4 t = *reinterpret_cast<const float4*>(zg + g + Hs); r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
            if (fw & (0x01010101u * F_BX)) { const float4 t = *reinterpret_cast<const float4*>(zg + g - Hs); l[0] = t.x; l[1] = t.y; l[2] = t.z; l[3] = t.w; }
            const float dn = (fw & ((unsigned)F_FY << 24)) ? zg[g + 4] : 0.f, up = (fw & (unsigned)F_BY) ? zg[g - 1] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned fl = (fw >> (8 * e)) & 0xffu;
                gx[e] = 0.f; gy[e] = 0.f;
                if (fl & F_FX) gx[e] = r[e] - c[e]; else if (fl & F_BX) gx[e] = c[e] - l[e];
                if (fl & F_FY) gy[e] = (e < 3 ? c[e < 3 ? e + 1 : 3] : dn) - c[e]; else if (fl & F_BY) gy[e] = c[e] - (e > 0 ? c[e > 0 ? e - 1 : 0] : up);
            }
        } else {
            const int gs[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int g = gs[e];
                const uint8_t fl = flags[g];
                c[e] = zg[g]; gx[e] = 0.f; gy[e] = 0.f;
                if (fl & F_FX) gx[e] = zg[g + Hs] - c[e]; else if (fl & F_BX) gx[e] = c[e] - zg[g - Hs];
                if (fl & F_FY) gy[e] = zg[g + 1] - c[e]; else if (fl & F_BY) gy[e] = c[e] - zg[g - 1];
            }
        }
        if (zc) reinterpret_cast<float4*>(zc)[q] = make_float4(c[0], c[1], c[2], c[3]);
        reinterpret_cast<float4*>(zx)[q] = make_float4(gx[0], gx[1], gx[2], gx[3]);
        reinterpret_cast<float4*>(zy)[q] = make_float4(gy[0], gy[1], gy[2], gy[3]);
    }
}
int grid_gradient(srps_ctx* ctx, const float* d_plane, float* d_zx, float* d_zy, float* d_compact) {
    Grid& G = ctx->grid;
    const bool al16 = ((uintptr_t)d_zx | (uintptr_t)d_zy | (uintptr_t)d_compact | (uintptr_t)d_plane | (uintptr_t)G.d_gofp) % 16 == 0;
    if (G.P % 4 == 0 && G.Hs % 4 == 0 && al16)
        hipLaunchKernelGGL(k_gradient4, dim3(std::min(cdiv(G.P / 4, 256), 4096)), dim3(256), 0, ctx->stream, d_plane, G.d_flags, G.d_gofp, G.P, G.Hs, d_zx, d_zy, d_compact);
    else
        hipLaunchKernelGGL(k_gradient, dim3(std::min(cdiv(G.P, 256), 4096)), dim3(256), 0, ctx->stream, d_plane, G.d_flags, G.d_gofp, G.P, G.Hs, d_zx, d_zy, d_compact);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// ---------------------------------------------------------------------------------------------
// rhs = KT' z0s + lambda (Dx' q0 + Dy' q1 + q2)        (devicecalls.cu:743-745, matrix-free)
// ---------------------------------------------------------------------------------------------
__global__ void k_rhs(const float* __restrict__ Q, const uint8_t* __restrict__ flags, const int* __restrict__ lr_index,
                      const float* __restrict__ z0s, int Hg, int Wg, int Hs, size_t plane, int sf, int Hl, float lambda,
                      float* __restrict__ rhs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= Hg || j >= Wg) return;
    const int g = (j + PAD) * Hs + i + PAD;
    const uint8_t fl = flags[g];
    if (!(fl & F_MASK)) return;
    const float* q0 = Q;
    const float* q1 = Q + plane;
    const float* q2 = Q + 2 * plane;
    float acc = q2[g];
    if (fl & F_FX) acc -= q0[g]; else if (fl & F_BX) acc += q0[g];
    if (flags[g - Hs] & F_FX) acc += q0[g - Hs];
    if (flags[g + Hs] & F_BX) acc -= q0[g + Hs];
    if (fl & F_FY) acc -= q1[g]; else if (fl & F_BY) acc += q1[g];
    if (flags[g - 1] & F_FY) acc += q1[g - 1];
    if (flags[g + 1] & F_BY) acc -= q1[g + 1];
    acc *= lambda;
    if (fl & F_KB) {
        const int lr = lr_index[(j / sf) * Hl + (i / sf)];
        acc += z0s[lr] / (float)(sf * sf);                       // KT value 1/sf^2, SRPS.cu:188
    }
    rhs[g] = acc;
}
// the same, four rows per thread (float4 loads; sf in {1, 2, 4}: a thread's rows lie in one LR block row group)
template <int SF>
__global__ __launch_bounds__(256) void k_rhs4(const float* __restrict__ Q, const uint8_t* __restrict__ flags, const int* __restrict__ lr_index,
                                              const float* __restrict__ z0s, int Hg, int Wg, int Hs, size_t plane, int Hl, float lambda,
                                              float* __restrict__ rhs) {
    const int i = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;          // first of the thread's four rows
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= Hg || j >= Wg) return;
    const size_t g = (size_t)(j + PAD) * Hs + i + PAD;                 // multiple of 4: aligned float4 / word
    const unsigned fw = *reinterpret_cast<const unsigned*>(flags + g);
    if ((fw & 0x01010101u) == 0u) return;                              // no masked pixel among the four (rhs stays 0)
    const unsigned fl_l = *reinterpret_cast<const unsigned*>(flags + g - Hs), fl_r = *reinterpret_cast<const unsigned*>(flags + g + Hs);
    const unsigned f_up = flags[g - 1], f_dn = flags[g + 4];
    const float* q0 = Q;
    const float* q1 = Q + plane;
    const float* q2 = Q + 2 * plane;
    const float4 a0 = *reinterpret_cast<const float4*>(q0 + g), al = *reinterpret_cast<const float4*>(q0 + g - Hs), ar = *reinterpret_cast<const float4*>(q0 + g + Hs);
    const float4 b0 = *reinterpret_cast<const float4*>(q1 + g), c0 = *reinterpret_cast<const float4*>(q2 + g);
    const float b_up = q1[g - 1], b_dn = q1[g + 4];
    const float q0c[4] = {a0.x, a0.y, a0.z, a0.w}, q0l[4] = {al.x, al.y, al.z, al.w}, q0r[4] = {ar.x, ar.y, ar.z, ar.w};
    const float q1c[6] = {b_up, b0.x, b0.y, b0.z, b0.w, b_dn};
    const float q2c[4] = {c0.x, c0.y, c0.z, c0.w};
    float kt[4] = {0.f, 0.f, 0.f, 0.f};
    {
#pragma unroll
        for (int e = 0; e < 4; e += SF) {                                // one LR block per SF rows
            if ((fw >> (8 * e)) & F_KB) {
                const float v = z0s[lr_index[(j / SF) * Hl + (i + e) / SF]] / (float)(SF * SF);      // KT value 1/sf^2, SRPS.cu:188
#pragma unroll
                for (int d = 0; d < SF; ++d) kt[e + d] = v;
            }
        }
    }
    float out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned fl = (fw >> (8 * e)) & 0xffu;
        const unsigned fu = (e == 0) ? f_up : ((fw >> (8 * (e - 1))) & 0xffu);
        const unsigned fd = (e == 3) ? f_dn : ((fw >> (8 * (e + 1))) & 0xffu);
        float acc = q2c[e];
        if (fl & F_FX) acc -= q0c[e]; else if (fl & F_BX) acc += q0c[e];
        if ((fl_l >> (8 * e)) & F_FX) acc += q0l[e];
        if ((fl_r >> (8 * e)) & F_BX) acc -= q0r[e];
        if (fl & F_FY) acc -= q1c[e + 1]; else if (fl & F_BY) acc += q1c[e + 1];
        if (fu & F_FY) acc += q1c[e];
        if (fd & F_BY) acc -= q1c[e + 2];
        acc *= lambda;
        if (fl & F_KB) acc += kt[e];
        out[e] = (fl & F_MASK) ? acc : 0.f;
    }
    *reinterpret_cast<float4*>(rhs + g) = make_float4(out[0], out[1], out[2], out[3]);
}

int grid_rhs(srps_ctx* ctx, const float* d_z0s) {
    Grid& G = ctx->grid;
    if (G.sf == 1 || G.sf == 2 || G.sf == 4) {
        dim3 grd(cdiv(G.Hg, 256), cdiv(G.Wg, 4));
#define SRPS_RHS4(SFV) hipLaunchKernelGGL((k_rhs4<SFV>), grd, dim3(256), 0, ctx->stream, G.d_q, G.d_flags, G.d_lr_index, d_z0s, G.Hg, G.Wg, G.Hs, G.plane, G.Hl, ctx->lambda, G.d_r)
        if (G.sf == 1) SRPS_RHS4(1); else if (G.sf == 2) SRPS_RHS4(2); else SRPS_RHS4(4);
#undef SRPS_RHS4
        SRPS_LAUNCH_CHECK();
        return SRPS_OK;
    }
    dim3 blk(64, 4), grd(cdiv(G.Hg, 64), cdiv(G.Wg, 4));
    hipLaunchKernelGGL(k_rhs, grd, blk, 0, ctx->stream, G.d_q, G.d_flags, G.d_lr_index, d_z0s, G.Hg, G.Wg, G.Hs, G.plane, G.sf, G.Hl, ctx->lambda, G.d_r);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// ---------------------------------------------------------------------------------------------
// operator, simple form: one thread per grid pixel, neighbours gathered from global memory.
// Works for every sf; it is the cross-check of (and the fall-back for) the marching kernel.
// MODE 0: out = A_ x                       (x = xin)
// MODE 1: r = r - A_ x ; partial r.r       (residual before CG, devicecalls.cu:758)
// MODE 2: CG step k: x := p_new = beta p + r (p_new = r when k == 1), out = A_ p_new,
//         p_out = p_new, partial p_new.out (devicecalls.cu:256-268)
// ---------------------------------------------------------------------------------------------
template <int MODE>
struct XRead {
    const float* x;
    const float* p;
    const float* r;
    float beta;
    int first;
    __device__ __forceinline__ float operator()(int g) const {
        if (MODE != 2) return x[g];
        if (first) return r[g];
        return scal_then_axpy(beta, p[g], r[g]);
    }
};

struct ApplyArgs {
    const float* M;
    const uint8_t* flags;
    const float* xin;      // MODE 0/1
    const float* p_in;     // MODE 2
    float* p_out;          // MODE 2
    float* r;              // MODE 1 (in/out), MODE 2 (in)
    float* out;            // MODE 0/2
    const float* rr_part;  // MODE 2: partials of r.r from the previous update kernel
    int n_rr;
    float* part_out;       // one partial per block
    CgScalars* scal;
    int Hg, Wg, Hs, sf;
    size_t plane;
    float lambda, inv_sf4, tol2;
    int k;
};

template <int MODE, typename XR>
__device__ __forceinline__ void grad_at(const XR& X, int g, uint8_t fl, int Hs, float& gx, float& gy, float& xc) {
    xc = X(g);
    gx = 0.f; gy = 0.f;
    if (fl & F_FX) gx = X(g + Hs) - xc; else if (fl & F_BX) gx = xc - X(g - Hs);
    if (fl & F_FY) gy = X(g + 1) - xc; else if (fl & F_BY) gy = xc - X(g - 1);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_apply_simple(ApplyArgs a) {
    __shared__ float sm[16];
    __shared__ double smd[4];
    XRead<MODE> X;
    X.x = a.xin; X.p = a.p_in; X.r = a.r; X.beta = 0.f; X.first = 1;
    if (MODE == 2) {
        const float r1 = (float)sum_partials(a.rr_part, a.n_rr, smd);
        if (!(r1 > a.tol2)) return;                         // converged: dc.cu:252
        X.first = (a.k == 1);
        if (!X.first) X.beta = r1 / a.scal->r0;             // dc.cu:262
    }
    const int Hs = a.Hs;
    const size_t pl = a.plane;
    const int nti = (a.Hg + 63) >> 6, ntj = (a.Wg + 3) >> 2;
    float acc_red = 0.f;
    for (int tile = blockIdx.x; tile < nti * ntj; tile += gridDim.x) {
        const int tj = tile / nti, ti = tile - tj * nti;
        const int i = ti * 64 + threadIdx.x, j = tj * 4 + threadIdx.y;
        if (i >= a.Hg || j >= a.Wg) continue;
        const int g = (j + PAD) * Hs + i + PAD;
        const uint8_t fl = a.flags[g];
        if (!(fl & F_MASK)) continue;
        float gx, gy, xc;
        grad_at<MODE>(X, g, fl, Hs, gx, gy, xc);
        const float m0 = a.M[g], m1 = a.M[pl + g], m2 = a.M[2 * pl + g], m3 = a.M[3 * pl + g], m4 = a.M[4 * pl + g], m5 = a.M[5 * pl + g];
        const float u = m0 * gx + m1 * gy + m2 * xc;
        const float v = m1 * gx + m3 * gy + m4 * xc;
        float acc = m2 * gx + m4 * gy + m5 * xc;                     // w
        if (fl & F_FX) acc -= u; else if (fl & F_BX) acc += u;       // own row of Dx'
        if (fl & F_FY) acc -= v; else if (fl & F_BY) acc += v;       // own row of Dy'
        {   // left neighbour's forward row references this pixel with +1
            const int gl = g - Hs; const uint8_t f2 = a.flags[gl];
            if (f2 & F_FX) { float ax, ay, xx; grad_at<MODE>(X, gl, f2, Hs, ax, ay, xx); acc += a.M[gl] * ax + a.M[pl + gl] * ay + a.M[2 * pl + gl] * xx; }
        }
        {   // right neighbour's backward row references this pixel with -1
            const int gr = g + Hs; const uint8_t f2 = a.flags[gr];
            if (f2 & F_BX) { float ax, ay, xx; grad_at<MODE>(X, gr, f2, Hs, ax, ay, xx); acc -= a.M[gr] * ax + a.M[pl + gr] * ay + a.M[2 * pl + gr] * xx; }
        }
        {   // upper neighbour (i-1), forward in y
            const int gu = g - 1; const uint8_t f2 = a.flags[gu];
            if (f2 & F_FY) { float ax, ay, xx; grad_at<MODE>(X, gu, f2, Hs, ax, ay, xx); acc += a.M[pl + gu] * ax + a.M[3 * pl + gu] * ay + a.M[4 * pl + gu] * xx; }
        }
        {   // lower neighbour (i+1), backward in y
            const int gd = g + 1; const uint8_t f2 = a.flags[gd];
            if (f2 & F_BY) { float ax, ay, xx; grad_at<MODE>(X, gd, f2, Hs, ax, ay, xx); acc -= a.M[pl + gd] * ax + a.M[3 * pl + gd] * ay + a.M[4 * pl + gd] * xx; }
        }
        acc *= a.lambda;
        if (fl & F_KB) {                                              // KT'KT: block sum / sf^4
            const int bi = (i / a.sf) * a.sf, bj = (j / a.sf) * a.sf;
            float S = 0.f;
            for (int dj = 0; dj < a.sf; ++dj)
                for (int di = 0; di < a.sf; ++di) S += X((bj + dj + PAD) * Hs + bi + di + PAD);
            acc += S * a.inv_sf4;
        }
        if (MODE == 0) {
            a.out[g] = acc;
        } else if (MODE == 1) {
            const float rv = a.r[g] - acc;
            a.r[g] = rv;
            acc_red = fmaf(rv, rv, acc_red);
        } else {
            a.p_out[g] = xc;
            a.out[g] = acc;
            acc_red = fmaf(xc, acc, acc_red);
        }
    }
    if (MODE != 0) {
        const float t = block_sum(acc_red, sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) a.part_out[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// CG vector update:  alpha = r1 / (p.w);  x += alpha p;  r -= alpha w;  partial r.r
// (devicecalls.cu:269-274 fused into one pass)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cg_update(int k, float* __restrict__ x, float* __restrict__ r,
                                                   const float* __restrict__ p, const float* __restrict__ w, size_t n4,
                                                   const float* __restrict__ rr_old, float* __restrict__ rr_new, int n_rr,
                                                   const float* __restrict__ pw_part, int n_pw, CgScalars* __restrict__ scal,
                                                   float tol2) {
    __shared__ float sm[16];
    __shared__ double smd[4];
    const float r1 = (float)sum_partials(rr_old, n_rr, smd);
    if (!(r1 > tol2)) {
        if (threadIdx.x == 0) {
            rr_new[blockIdx.x] = rr_old[blockIdx.x];
            if (blockIdx.x == 0) scal->active = 0;
        }
        return;
    }
    const float dot = (float)sum_partials(pw_part, n_pw, smd);
    const float alpha = r1 / dot;                                   // dc.cu:269
    float acc = 0.f;
    float4* x4 = reinterpret_cast<float4*>(x);
    float4* r4 = reinterpret_cast<float4*>(r);
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < n4; t += (size_t)gridDim.x * 256) {
        const float4 pv = p4[t], wv = w4[t];
        float4 xv = x4[t], rv = r4[t];
        xv.x = fmaf(alpha, pv.x, xv.x); xv.y = fmaf(alpha, pv.y, xv.y); xv.z = fmaf(alpha, pv.z, xv.z); xv.w = fmaf(alpha, pv.w, xv.w);
        rv.x = fmaf(-alpha, wv.x, rv.x); rv.y = fmaf(-alpha, wv.y, rv.y); rv.z = fmaf(-alpha, wv.z, rv.z); rv.w = fmaf(-alpha, wv.w, rv.w);
        x4[t] = xv; r4[t] = rv;
        acc = fmaf(rv.x, rv.x, acc); acc = fmaf(rv.y, rv.y, acc); acc = fmaf(rv.z, rv.z, acc); acc = fmaf(rv.w, rv.w, acc);
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        rr_new[blockIdx.x] = t;
        if (blockIdx.x == 0) { scal->r0 = r1; scal->r1_last = r1; scal->iters = k; scal->active = 1; }
    }
}

// Update kernel of the fused protocol (marching operator): alpha = r1 / (p.w); r -= alpha w; partial r.r.
// x += alpha p is left to the next operator launch (or to k_cg_flush_x after the last step).
__global__ __launch_bounds__(256) void k_cg_update_r(int k, float* __restrict__ r, const float* __restrict__ w, size_t n4,
                                                     const float* __restrict__ rr_old, float* __restrict__ rr_new, int n_rr,
                                                     const float* __restrict__ pw_part, int n_pw, CgScalars* __restrict__ scal,
                                                     float tol2) {
    __shared__ float sm[16];
    __shared__ double smd[4];
    const float r1 = (float)sum_partials(rr_old, n_rr, smd);
    if (!(r1 > tol2)) {
        if (threadIdx.x == 0) {
            rr_new[blockIdx.x] = rr_old[blockIdx.x];
            if (blockIdx.x == 0) scal->active = 0;
        }
        return;
    }
    const float dot = (float)sum_partials(pw_part, n_pw, smd);
    const float alpha = r1 / dot;                                   // dc.cu:269
    float acc = 0.f;
    float4* r4 = reinterpret_cast<float4*>(r);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < n4; t += (size_t)gridDim.x * 256) {
        const float4 wv = w4[t];
        float4 rv = r4[t];
        rv.x = fmaf(-alpha, wv.x, rv.x); rv.y = fmaf(-alpha, wv.y, rv.y); rv.z = fmaf(-alpha, wv.z, rv.z); rv.w = fmaf(-alpha, wv.w, rv.w);   // dc.cu:272
        r4[t] = rv;
        acc = fmaf(rv.x, rv.x, acc); acc = fmaf(rv.y, rv.y, acc); acc = fmaf(rv.z, rv.z, acc); acc = fmaf(rv.w, rv.w, acc);
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        rr_new[blockIdx.x] = t;
        if (blockIdx.x == 0) { scal->r0 = r1; scal->r1_last = r1; scal->iters = k; scal->active = 1; scal->alpha = alpha; }
    }
}

// the one x update still pending after the last executed step: x += alpha_K p_K
__global__ __launch_bounds__(256) void k_cg_flush_x(float* __restrict__ x, const float* __restrict__ p0, const float* __restrict__ p1,
                                                    size_t n4, const CgScalars* __restrict__ scal) {
    const int it = scal->iters;
    if (it < 1) return;
    const float alpha = scal->alpha;
    const float4* p4 = reinterpret_cast<const float4*>((it & 1) ? p1 : p0);
    float4* x4 = reinterpret_cast<float4*>(x);
    for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < n4; t += (size_t)gridDim.x * 256) {
        const float4 pv = p4[t];
        float4 xv = x4[t];
        xv.x = fmaf(alpha, pv.x, xv.x); xv.y = fmaf(alpha, pv.y, xv.y); xv.z = fmaf(alpha, pv.z, xv.z); xv.w = fmaf(alpha, pv.w, xv.w);
        x4[t] = xv;
    }
}

// the x update still pending after the last executed step K of the one-launch protocol: alpha_K = r_{K-1}.r_{K-1} / p_K.omega_K from
// the sums launch K left (dc.cu:269-270)
// x2: the steps ran with the x update every second launch (option "march_x2"): after an EVEN last step K two updates are pending --
// alpha_{K-1} p_{K-1} (alpha_{K-1}: left by launch K in alpha_hist[(K - 1) & 1]) and alpha_K p_K, applied in that order
__global__ __launch_bounds__(256) void k_cg_flush_x2(float* __restrict__ x, const float* __restrict__ p0, const float* __restrict__ p1,
                                                     size_t n4, const float* __restrict__ part4, int n_part, int n_live, CgScalars* __restrict__ scal, const double* __restrict__ totals4, int x2) {
    __shared__ double smd4[4][4];
    const int it = scal->iters;
    if (it < 1) return;
    double s4[4];
    if (totals4) { s4[0] = totals4[4 * (it & 1)]; s4[3] = totals4[4 * (it & 1) + 3]; }      // strip-partitioned CG: the sums of launch `it` over all ranks
    else sum_partials4(part4 + (size_t)(it & 1) * 4 * n_part, n_live, n_part, s4, smd4);
    const float alpha = (float)s4[3] / (float)s4[0];
    const float4* p4 = reinterpret_cast<const float4*>((it & 1) ? p1 : p0);
    const bool two = x2 && it >= 2 && (it & 1) == 0;
    const float alpha_before = two ? scal->alpha_hist[(it - 1) & 1] : 0.f;
    const float4* q4 = reinterpret_cast<const float4*>((it & 1) ? p0 : p1);      // p_{K-1}
    float4* x4 = reinterpret_cast<float4*>(x);
    for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < n4; t += (size_t)gridDim.x * 256) {
        const float4 pv = p4[t];
        float4 xv = x4[t];
        if (two) {
            const float4 qv = q4[t];
            xv.x = fmaf(alpha_before, qv.x, xv.x); xv.y = fmaf(alpha_before, qv.y, xv.y); xv.z = fmaf(alpha_before, qv.z, xv.z); xv.w = fmaf(alpha_before, qv.w, xv.w);
        }
        xv.x = fmaf(alpha, pv.x, xv.x); xv.y = fmaf(alpha, pv.y, xv.y); xv.z = fmaf(alpha, pv.z, xv.z); xv.w = fmaf(alpha, pv.w, xv.w);
        x4[t] = xv;
    }
}
__global__ void k_cg_reset2(CgScalars* scal) {
    if (threadIdx.x == 0) { scal->r0 = 0.f; scal->r1_last = 0.f; scal->iters = 0; scal->active = 1; scal->alpha = 0.f; scal->alpha_hist[0] = 0.f; scal->alpha_hist[1] = 0.f; }
}

__global__ void k_cg_reset(CgScalars* scal, float* rr0, int n_rr, const float* first) {
    // rr_part[0][0] = r.r of the initial residual, the rest zero
    for (int t = threadIdx.x; t < n_rr; t += blockDim.x) rr0[t] = (t == 0) ? first[0] : 0.f;
    if (threadIdx.x == 0) { scal->r0 = 0.f; scal->r1_last = first[0]; scal->iters = 0; scal->active = 1; scal->alpha = 0.f; }
}

// single-block sum (same routine as k_final_sum in kernels_pixel.hip)
__global__ void k_sum_to(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ double smd[4];
    const double t = sum_partials(part, n, smd);
    if (threadIdx.x == 0) out[0] = (float)t;
}

static ApplyArgs base_args(srps_ctx* ctx) {
    Grid& G = ctx->grid;
    ApplyArgs a;
    memset(&a, 0, sizeof(a));
    a.M = G.d_M; a.flags = G.d_flags; a.Hg = G.Hg; a.Wg = G.Wg; a.Hs = G.Hs; a.sf = G.sf; a.plane = G.plane;
    a.lambda = ctx->lambda;
    a.inv_sf4 = 1.0f / ((float)(G.sf * G.sf) * (float)(G.sf * G.sf));
    a.scal = G.d_scal;
    a.part_out = G.d_pw_part;
    return a;
}

// A kernel with grid-wide sums makes no progress unless all its blocks are resident together.  The occupancy query gives the
// blocks one CU takes; blocks <= CUs x that number is checked here.  Two persistent kernels launched at the same time on
// one device could still interleave their blocks and wait for each other: hipLaunchCooperativeKernel sends such kernels
// through one queue per device and rules that out, at the price of 13 us of queue time before and after the kernel.
// Default (coop_launch = 1): the cooperative launch; option "exclusive_device" = 1 (coop_launch = 0) selects the plain launch
// (same residency) for a device that nothing else uses.  Neither launch can see other processes, CU masks or a hardware that
// admits one block fewer than the API reports -- which is why every wait inside the kernels is bounded (device_utils.h
// SpinState): such a launch ends with an abort flag within the spin budget and the host repeats the phase with the streaming
// kernels.  SRPS_ERR_UNSUPPORTED: does not fit, the caller streams instead.
int launch_persistent(srps_ctx* ctx, const void* fn, int blocks, int threads, void** args, size_t lds_bytes) {
    ctx->persistent_inflight = 1;
    int per_cu = 0;
    const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds_bytes);
    if (oe != hipSuccess || per_cu < 1 || (long)blocks > (long)per_cu * ctx->num_cus) {
        (void)hipGetLastError();
        return SRPS_ERR_UNSUPPORTED;
    }
    if (ctx->coop_launch == 1 || (ctx->coop_launch == 2 && contexts_on_device(ctx->device) > 1)) {
        const hipError_t le = hipLaunchCooperativeKernel(fn, dim3(blocks), dim3(threads), args, lds_bytes, ctx->stream);
        if (le == hipErrorCooperativeLaunchTooLarge || le == hipErrorLaunchOutOfResources || le == hipErrorNotSupported) {
            (void)hipGetLastError();
            return SRPS_ERR_UNSUPPORTED;
        }
        SRPS_HIP(le);
        return SRPS_OK;
    }
    SRPS_HIP(hipLaunchKernel(fn, dim3(blocks), dim3(threads), args, lds_bytes, ctx->stream));
    return SRPS_OK;
}

bool cg_fused_step(const srps_ctx* ctx) { return ctx->cg_fused_step && use_march(ctx); }

bool use_march(const srps_ctx* ctx) {
    if (ctx->apply_mode == SRPS_APPLY_SIMPLE) return false;
    return march_supported(ctx);
}
int apply_blocks(const srps_ctx* ctx) { return use_march(ctx) ? march_blocks(ctx->grid) : ctx->grid.nb_apply; }

// the kernels that stream the stored 6-plane tensor need it to exist
static int need_stored_tensor(srps_ctx* ctx) {
    if ((!use_march(ctx) || march_recompute_channels(ctx) == 0) && !ctx->grid.M_valid) {
        set_error("the stored tensor was not assembled (tensor_recompute is active): set option keep_stored_tensor=1 or tensor_recompute=0 before the depth assembly");
        return SRPS_ERR_STATE;
    }
    return SRPS_OK;
}

int grid_apply_plain(srps_ctx* ctx, const float* d_in_plane, float* d_out_plane) {
    Grid& G = ctx->grid;
    SRPS_TRY(need_stored_tensor(ctx));
    if (use_march(ctx)) return march_apply_plain(ctx, d_in_plane, d_out_plane);
    ApplyArgs a = base_args(ctx);
    a.xin = d_in_plane; a.out = d_out_plane;
    hipLaunchKernelGGL((k_apply_simple<0>), dim3(G.nb_apply), dim3(64, 4), 0, ctx->stream, a);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

int grid_residual(srps_ctx* ctx) {
    Grid& G = ctx->grid;
    SRPS_TRY(need_stored_tensor(ctx));
    if (use_march(ctx)) {
        SRPS_TRY(march_residual(ctx));
    } else {
        ApplyArgs a = base_args(ctx);
        a.xin = G.d_x; a.r = G.d_r;
        hipLaunchKernelGGL((k_apply_simple<1>), dim3(G.nb_apply), dim3(64, 4), 0, ctx->stream, a);
        SRPS_LAUNCH_CHECK();
    }
    if (cg_fused_step(ctx)) {                      // the first step sums the residual's partials itself
        hipLaunchKernelGGL(k_cg_reset2, dim3(1), dim3(64), 0, ctx->stream, G.d_scal);
        SRPS_LAUNCH_CHECK();
        return SRPS_OK;
    }
    float* first = G.d_misc_part + 4000;
    hipLaunchKernelGGL(k_sum_to, dim3(1), dim3(256), 0, ctx->stream, G.d_pw_part, apply_blocks(ctx), first);
    hipLaunchKernelGGL(k_cg_reset, dim3(1), dim3(256), 0, ctx->stream, G.d_scal, G.d_rr_part, G.nb_update, first);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

int cg_launch_apply(srps_ctx* ctx, int k) {
    Grid& G = ctx->grid;
    if (cg_fused_step(ctx)) return march_cg_step(ctx, k);
    if (use_march(ctx)) return march_cg_apply(ctx, k);
    ApplyArgs a = base_args(ctx);
    float* pbuf[2] = {G.d_p, G.d_p + G.plane};
    a.p_in = pbuf[(k + 1) & 1]; a.p_out = pbuf[k & 1]; a.r = G.d_r; a.out = G.d_w;
    a.rr_part = G.d_rr_part + (size_t)((k - 1) & 1) * G.nb_update; a.n_rr = G.nb_update;
    a.k = k;
    a.tol2 = ctx->cg_fixed ? -1.f : ctx->cg_tol * ctx->cg_tol;
    hipLaunchKernelGGL((k_apply_simple<2>), dim3(G.nb_apply), dim3(64, 4), 0, ctx->stream, a);
    return SRPS_OK;
}

int cg_launch_update(srps_ctx* ctx, int k) {
    Grid& G = ctx->grid;
    if (cg_fused_step(ctx)) return SRPS_OK;       // nothing left to do: the next launch applies this step's updates
    float* pbuf[2] = {G.d_p, G.d_p + G.plane};
    const float tol2 = ctx->cg_fixed ? -1.f : ctx->cg_tol * ctx->cg_tol;
    if (use_march(ctx)) {      // fused protocol: the operator kernel applies x += alpha p
        hipLaunchKernelGGL(k_cg_update_r, dim3(G.nb_update), dim3(256), 0, ctx->stream, k, G.d_r, G.d_w, G.used / 4,
                           G.d_rr_part + (size_t)((k - 1) & 1) * G.nb_update, G.d_rr_part + (size_t)(k & 1) * G.nb_update, G.nb_update,
                           G.d_pw_part, apply_blocks(ctx), G.d_scal, tol2);
        return SRPS_OK;
    }
    hipLaunchKernelGGL(k_cg_update, dim3(G.nb_update), dim3(256), 0, ctx->stream, k, G.d_x, G.d_r, pbuf[k & 1], G.d_w, G.used / 4,
                       G.d_rr_part + (size_t)((k - 1) & 1) * G.nb_update, G.d_rr_part + (size_t)(k & 1) * G.nb_update, G.nb_update,
                       G.d_pw_part, apply_blocks(ctx), G.d_scal, tol2);
    return SRPS_OK;
}

int cg_flush_x(srps_ctx* ctx) {
    Grid& G = ctx->grid;
    if (!use_march(ctx)) return SRPS_OK;       // classic protocol: k_cg_update already moved x
    if (cg_fused_step(ctx)) {
        // a rank of the strip-partitioned CG moves its own columns only (the others arrive with the all-gather of x)
        const size_t off = G.view_w > 0 ? (size_t)(G.view_c0 + PAD) * G.Hs : 0, n4 = G.view_w > 0 ? (size_t)G.view_w * G.Hs / 4 : G.used / 4;
        hipLaunchKernelGGL(k_cg_flush_x2, dim3(G.nb_update), dim3(256), 0, ctx->stream, G.d_x + off, G.d_p + off, G.d_p + G.plane + off, n4, G.d_part4, G.n_part4,
                           march_blocks(G), G.d_scal, (const double*)G.d_totals4, march_x2_on(ctx) ? 1 : 0);
        SRPS_LAUNCH_CHECK();
        return SRPS_OK;
    }
    hipLaunchKernelGGL(k_cg_flush_x, dim3(G.nb_update), dim3(256), 0, ctx->stream, G.d_x, G.d_p, G.d_p + G.plane, G.used / 4, G.d_scal);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// the residual of devicecalls.cu:758 (G.d_r: b -> b - A_ x) and the loop of devicecalls.cu:252-275: "while (r1 > tol^2 && k <= max_iter)" => up to max_iter+1 steps.
// Convergence is tested on the device by every kernel; the host just enqueues the steps.
// A persistent kernel of this pass (the albedo CG) has not been looked at yet: should it turn out to have given up -- on this rank
// or on another one, reported through the energy all-reduce --, the solve ran on an albedo that was never finished and the pass's
// tail is repeated -- from the iterate this solve starts from, which the streaming kernels (and the strips) update in place.
// Keep a copy (one plane: 0.3 % of the solve's traffic); the abort check makes it the current plane again (persistent_aborts).
static int keep_start_plane(srps_ctx* ctx, bool fixed_steps) {
    if (ctx->persistent_inflight && !ctx->x_swapped && !fixed_steps) {
        Grid& G = ctx->grid;
        SRPS_HIP(hipMemcpyAsync(G.d_x2, G.d_x, G.plane * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        ctx->x_swapped = true;
    }
    return SRPS_OK;
}

int grid_cg(srps_ctx* ctx, int max_steps, bool fixed_steps) {
    // the strips update x in place like the streaming kernels below: the copy of the start plane comes first (round-3 advisor
    // finding: with cg_partition = 1 an aborted albedo launch of the same pass found no plane to go back to and the repeat
    // rebuilt normals and dz around the discarded solve's depth)
    // cg_partition = 2: the resident kernel on this rank's strip of tile columns, side by side with the other ranks' (kernels_resident.hip:
    // resident_cg_rank); where that does not fit or could not be set up: the streaming strips (1) or the replicated CG
    if (ctx->cg_strips == 2 && comm_bound(ctx) && ctx->comm_world > 1) {
        const int rc = resident_cg_rank(ctx, max_steps, fixed_steps);
        if (rc != SRPS_ERR_UNSUPPORTED) return rc;
    }
    if (strips_active(ctx)) {
        SRPS_TRY(keep_start_plane(ctx, fixed_steps));
        return strips_cg(ctx, max_steps, fixed_steps);      // this rank's columns only, RCCL between the steps
    }
    if (resident_supported(ctx)) {                 // residual and CG in one persistent launch
        const int rc = resident_cg(ctx, max_steps, fixed_steps);
        if (rc != SRPS_ERR_UNSUPPORTED) return rc;
        ctx->cg_resident = 0;                      // the device refused the launch: stream from now on
    }
    SRPS_TRY(keep_start_plane(ctx, fixed_steps));
    SRPS_TRY(grid_residual(ctx));                  // dc.cu:758
    ctx->cg_fixed = fixed_steps;                   // bench: never stop early (tol^2 := -1)
    for (int k = 1; k <= max_steps; ++k) {
        cg_launch_apply(ctx, k);
        cg_launch_update(ctx, k);
    }
    ctx->cg_fixed = false;
    SRPS_TRY(cg_flush_x(ctx));
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// ---------------------------------------------------------------------------------------------
// t1 = || KT z - z0s ||^2  (devicecalls.cu:762-766): one thread per LR block of the bounding box
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_energy_t1(const float* __restrict__ zg, const int* __restrict__ lr_index,
                                                   const float* __restrict__ z0s, int Hl, int Wl, int Hs, int sf,
                                                   float* __restrict__ part) {
    __shared__ float sm[16];
    float acc = 0.f;
    const int n = Hl * Wl;
    const float inv = 1.0f / (float)(sf * sf);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        const int lr = lr_index[t];
        if (lr < 0) continue;
        const int bj = t / Hl, bi = t - bj * Hl;
        float S = 0.f;
        for (int dj = 0; dj < sf; ++dj)
            for (int di = 0; di < sf; ++di) S += inv * zg[(bj * sf + dj + PAD) * Hs + bi * sf + di + PAD];
        const float d = S - z0s[lr];
        acc = fmaf(d, d, acc);
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
int grid_energy_t1(srps_ctx* ctx, const float* d_z0s, float* d_out, const float** part_out, int* n_part_out) {
    Grid& G = ctx->grid;
    const int nb = std::max(1, std::min(cdiv(G.Hl * G.Wl, 256), 1024));
    float* part = G.d_misc_part + 2048;
    hipLaunchKernelGGL(k_energy_t1, dim3(nb), dim3(256), 0, ctx->stream, G.d_x, G.d_lr_index, d_z0s, G.Hl, G.Wl, G.Hs, G.sf, part);
    if (part_out) { *part_out = part; *n_part_out = nb; }      // the energy sweep's last block adds them (ReportFinish)
    else hipLaunchKernelGGL(k_sum_to, dim3(1), dim3(256), 0, ctx->stream, part, nb, d_out);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

}  // namespace srps
