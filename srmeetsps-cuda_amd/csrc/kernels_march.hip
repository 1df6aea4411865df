// kernels_march.hip -- the depth operator as a register-marching stencil for gfx950.
//
//   A_ x = KT'(KT x) + lambda ( Dx'u + Dy'v + w ),  (u,v,w) = M (Dx x, Dy x, x)      
//
// One wave (64 lanes) owns a strip of TJ grid columns times a segment of up to 248 rows:
//   * lane l holds 4 consecutive rows (one float4 per plane and column, 16 B/lane coalesced
//     loads of 1 KiB per wave-instruction); lanes 0 and n+1 are halo lanes (computed, not stored),
//     so a wave needs nothing from another wave: no LDS, no barrier in the main loop;
//   * the wave marches over the columns of its strip keeping a rolling window of x (5 columns),
//     u (3 columns), v, w and the structure bytes in registers: the x-direction stencil is
//     register-to-register, the y-direction neighbours (rows i-1, i+1) come from the adjacent
//     lane through DPP wave shifts (v_mov_b32 dpp wave_shr:1 / wave_shl:1);
//   * the sf x sf block sum of KT'KT is formed from the lane's own 4 rows and sf columns of the
//     window (sf in {1,2,4});
//   * in CG mode the search-direction update p = beta p + r is applied while loading (also in the
//     halo), p and omega = A_ p are stored for the owned pixels and p.omega is reduced
//     wave -> block (one partial per block, summed deterministically by the next kernel).
// Every plane element of M, p, r and the structure bytes is read once per strip plus the halo
// ((TJ+2)/TJ columns for M, (TJ+5)/TJ for p and r, 66/62 rows).
#include "srps_internal.h"
#include <type_traits>
#include "device_utils.h"

namespace srps {

struct MarchArgs {
    const float* M;
    const uint8_t* flags;
    const float* xin;      // MODE 0/1
    const float* p_in;     // MODE 2
    float* p_out;          // MODE 2
    float* r;              // MODE 1 (in/out), MODE 2 (in), MODE 3 (in: the residual before the pending update)
    float* r_out;          // MODE 3: the updated residual goes to the OTHER plane (neighbouring strips still read the old one as halo)
    float* out;            // MODE 0/2
    float* x;              // MODE 2: the iterate; x += alpha_prev * p_in is applied here (deferred from the last update)
    const float* rr_part;
    int n_rr;
    float* part_out;
    CgScalars* scal;
    int Hg, Wg, Hs, Ws;
    size_t plane;
    float lambda, inv_sf4, tol2;
    int k;
    int own, n_seg, n_strip, n_items, tj, snake;
    // tensor-recompute mode (NC > 0): M is rebuilt per pixel from g_c = (rho_c/dz)^2 (NC planes) and
    // 8 constants per channel derived from the lighting (see k_tensor_consts)
    const float* G;
    const float* consts;   // [NC][8]: Sbb, x*, y*, R00, R01, R11, 0, 0
    float cx, cy;
    int i_lo, j_lo;
    // MODE 3 (the whole CG step in one launch): omega of the previous step, and four partial sums per block --
    // p.omega, r.omega, omega.omega, r.r -- of the previous launch (in) and of this one (out), [4][n_part]
    const float* w_prev;
    const float* part4_in;
    float* part4_out;
    int n_part;
    // strip-partitioned CG (several ranks, each a range of columns): the four sums of the previous step over ALL ranks, all-reduced
    // between the launches (double, the same bits on every rank); null: one rank, the sums are this grid's block partials
    const double* totals4;
};

__device__ __forceinline__ float dpp_from_prev_lane(float v) {      // lane i <- lane i-1 ; lane 0 <- 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_next_lane(float v) {      // lane i <- lane i+1 ; lane 63 <- 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ unsigned dpp_from_prev_lane(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned dpp_from_next_lane(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}

__device__ __forceinline__ float mul_then_add(float beta, float p, float r) {
#pragma clang fp contract(off)
    const float t = beta * p;      // Sscal dc.cu:263
    return t + r;                  // Saxpy dc.cu:264
}

// value if bit `bit` of the structure word is set, else +0.0 -- v_bfe_i32 + v_and_b32, no select, no branch
template <int BIT>
__device__ __forceinline__ float if_bit(float v, unsigned flword) {
    const int m = __builtin_amdgcn_sbfe((int)flword, BIT, 1);      // 0 or 0xffffffff
    return __int_as_float(__float_as_int(v) & m);
}
// bit positions inside one structure byte (F_* = 1 << position)
constexpr int B_FX = 1, B_BX = 2, B_FY = 3, B_BY = 4, B_KB = 5;

// a structure word whose four bytes are each "interior" (F_MASK|F_FX|F_FY|F_KB = 0x2B) or empty (0)
__device__ __forceinline__ bool is_plain(unsigned w) { return w == (w & 0x01010101u) * 0x2Bu; }
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

struct F4 {
    float e[4];
};
// (Non-temporal forms of these loads and stores were measured and removed: the one-launch step re-reads what it wrote a step earlier, and
// its strips re-read their halo columns -- 15 - 35 % slower at 2304^2 ... 4096^2 with every stream non-temporal, 4 - 15 % with the stores only,
// run-to-run noise with the g planes only.  What does pay is order: march_snake = 2 below.  docs/HISTORY.md, profiles/r04_march_nt_sweep.txt)
__device__ __forceinline__ F4 ld4(const float* __restrict__ p) {
    F4 r;
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.e[0] = t.x; r.e[1] = t.y; r.e[2] = t.z; r.e[3] = t.w;
    return r;
}
__device__ __forceinline__ void st4(float* __restrict__ p, const F4& a) { *reinterpret_cast<float4*>(p) = make_float4(a.e[0], a.e[1], a.e[2], a.e[3]); }
__device__ __forceinline__ F4 zero4() { F4 r; r.e[0] = r.e[1] = r.e[2] = r.e[3] = 0.f; return r; }

// MODE 0: out = A_ xin.  MODE 1: r -= A_ xin, partial r.r (residual, dc.cu:758).  MODE 2: p = beta p + r, out = A_ p, partial
// p.out, x += alpha_prev p_prev (the update kernel k_cg_update_r follows).  MODE 3: the whole CG step k in ONE launch --
// the updates of step k-1 that wait for its alpha are applied while the vectors stream through anyway:
//     alpha_{k-1} = r_{k-2}.r_{k-2} / p_{k-1}.omega_{k-1}                    (sums of the previous launch, dc.cu:269)
//     r_{k-1} = r_{k-2} - alpha_{k-1} omega_{k-1},  x_{k-1} = x_{k-2} + alpha_{k-1} p_{k-1}      (dc.cu:270-272)
//     beta_k = r_{k-1}.r_{k-1} / r_{k-2}.r_{k-2},  p_k = beta_k p_{k-1} + r_{k-1},  omega_k = A_ p_k     (dc.cu:262-268)
// r_{k-1}.r_{k-1} is needed for beta_k before r_{k-1} has been summed: it is taken as r.r - 2 alpha r.omega + alpha^2 omega.omega
// from the previous launch's sums (the same identity as in the resident kernel).  The DIRECT sum of r_{k-1}.r_{k-1} is formed by
// this launch and is what the next launch uses for alpha_k and as the denominator of beta_{k+1}: every predicted value is
// anchored on a direct sum one step old, the prediction error does not accumulate.  45 B per unknown instead of 37 + 12, one
// launch per step instead of two.
// XU (MODE 3): what this launch does with x.  1: x += alpha_{k-1} p_{k-1}, every launch (k > 1).  The two-step form (option "march_x2"):
// 0: nothing -- an even launch leaves its update pending --, 2: an odd launch k >= 3 applies BOTH pending updates, x += alpha_{k-2} p_{k-2}
// then += alpha_{k-1} p_{k-1}: the same two fused multiply-adds a launch apart would have done, so the same bits; p_{k-2} is what the
// plane p_out still holds when this launch reaches the column (the lane that overwrites it with p_k reads it first).  x is then
// read and written every second step and p_{k-2} read once: 43 B per unknown and step instead of 45.
template <int SF, int MODE, int NC, int XU = 1>
__global__ __launch_bounds__(256, 2) void k_apply_march(MarchArgs a) {
    const int TJ = a.tj;                      // strip width: even, multiple of SF (runtime, the loop is unrolled by two)
    constexpr int NT = (NC > 0) ? NC : 6;     // planes streamed for the tensor: NC (recompute) or 6 (stored M)
    constexpr int L = (SF == 4) ? 3 : 2;      // look-ahead columns of the x window
    constexpr int NX = L + 2;                 // window holds x[c-1 .. c+L]
    __shared__ float sm[16];
    __shared__ double smd[4];
    __shared__ double smd4[4][4];
    __shared__ float sm4[4 + 4 * 4];
    float beta = 0.f;
    float alpha_prev3 = 0.f;
    float alpha_pp = 0.f;                     // XU == 2: alpha_{k-2}
    if (MODE == 3) {
        if (!a.scal->active) return;                                   // an earlier launch found r.r <= tol^2 (dc.cu:252)
        double s4[4];
        if (a.totals4) { s4[0] = a.totals4[0]; s4[1] = a.totals4[1]; s4[2] = a.totals4[2]; s4[3] = a.totals4[3]; }
        else sum_partials4(a.part4_in, (int)gridDim.x, a.n_part, s4, smd4);      // the previous launch had this launch's blocks
        float r1;
        const float r1_direct = (float)s4[3];                          // r_{k-2}.r_{k-2} (k == 1: of the initial residual)
        if (a.k == 1) r1 = r1_direct;
        else {
            alpha_prev3 = r1_direct / (float)s4[0];                    // dc.cu:269
            const double t1 = 2.0 * (double)alpha_prev3 * s4[1], t2 = (double)alpha_prev3 * (double)alpha_prev3 * s4[2];
            r1 = (float)((double)r1_direct - t1 + t2);
            beta = r1 / r1_direct;                                     // dc.cu:262
            // The three terms come from fp32 per-block partial sums: when they cancel to less than two digits (one step cut the
            // residual by ~1e5: systems that converge within the 101 steps) the predicted value is noise, possibly <= 0.  Then
            // (the resident kernel's guard, kernels_resident.hip) nothing is decided on it: the direction restarts (beta = 0, a
            // steepest-descent step: alpha = r.r / p.omega stays exact because r is orthogonal to the previous p whatever beta
            // was) and the stop test waits for the DIRECT sum, which the next launch holds as its r1_direct.
            if (!((double)r1 > 1e-2 * ((double)r1_direct + fabs(t1) + t2))) { beta = 0.f; r1 = r1_direct; }
        }
        if (!(r1 > a.tol2)) {                                          // converged: dc.cu:252.  The pending x update is k_cg_flush_x2's
            if (blockIdx.x == 0 && threadIdx.x == 0) a.scal->active = 0;
            return;
        }
        if (XU == 2) alpha_pp = a.scal->alpha_hist[a.k & 1];           // left by launch k - 1 in slot (k - 2) & 1; this launch writes the OTHER slot
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.scal->iters = a.k; a.scal->r0 = r1_direct; a.scal->r1_last = r1; a.scal->alpha = alpha_prev3; a.scal->alpha_hist[(a.k - 1) & 1] = alpha_prev3; }
    }
    if (MODE == 2) {
        const float r1 = (float)sum_partials(a.rr_part, a.n_rr, smd);
        if (!(r1 > a.tol2)) return;                                   // converged: dc.cu:252
        // step 1: p = r (dc.cu:258).  The host passes p_in = r and beta stays 0, so that 0*r + r = r
        // goes through the same branch-free load path as every other step.
        if (a.k != 1) beta = r1 / a.scal->r0;                         // dc.cu:262
    }
    // x_k = x_{k-1} + alpha_k p_k (dc.cu:270) of the PREVIOUS step is applied by this launch, which streams
    // p_k anyway; the update kernel then only touches r and omega (12 B/unknown instead of 24)
    const float alpha_prev = (MODE == 3) ? alpha_prev3 : (MODE == 2 && a.k != 1) ? a.scal->alpha : 0.f;
    // XCD-aware block order: blocks are dealt round-robin over the 8 XCDs; give every XCD a
    // contiguous range of work items (neighbouring strips share halo columns through its L2)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, kk = bid >> 3;
        bid = xcd * q + min(xcd, rem) + kk;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = __builtin_amdgcn_readfirstlane(bid * 4 + wave);      // wave-uniform: keeps the column loop scalar
    float red = 0.f, red_rw = 0.f, red_ww = 0.f, red_rr = 0.f;
    const int strip = (item < a.n_items) ? item / a.n_seg : 0;
    // D = +1: the wave marches left to right over its strip, D = -1: right to left.  Odd strips march
    // backwards ("snake"): two neighbouring strips then reach their shared halo columns at the same time
    // (both at the start or both at the end of their march), so the second reader hits the XCD's L2
    // instead of fetching the columns again from the fabric a whole kernel later.
    auto body = [&](auto dtag) {
        constexpr int D = decltype(dtag)::value;
        const int seg = item - strip * a.n_seg;
        const int Hs = a.Hs;
        const size_t pl = a.plane;
        const int row0 = seg * a.own + 4 * lane;                       // storage row of element 0 (PAD == halo == 4)
        const int nl = a.own >> 2;                                     // owned lanes are 1..nl
        const bool act = (lane <= nl + 1) && (row0 < Hs);              // lanes that take part at all
        const bool owned = act && lane >= 1 && lane <= nl && (row0 - PAD < a.Hg);
        const int c0 = strip * TJ + PAD;                               // first storage column of the strip
        // Inactive lanes read rows 0..3 of the column instead: that is the zero halo of every plane
        // (never written), so no load in the loop needs a predicate or a branch.
        const int rowL = act ? row0 : 0;

        const float* tens = (NC > 0) ? a.G : a.M;
        // Loads of one step, kept RAW (unconverted) so that they can stay in flight for a whole step:
        // tensor planes and structure word of the next column (c+D), x of column c+D*L.
        struct Raw {
            F4 T[NT];
            unsigned fl;
            F4 r, p;
            F4 xo, po;         // MODE 2, 3: x and the previous p of the column this step outputs
            F4 po2;            // XU == 2: p of two steps ago at that column (still in the plane this launch overwrites)
            F4 wp;             // MODE 3: omega of the previous step (column c + D*L)
        };
        auto issue = [&](Raw& w, int c) {
#pragma unroll
            for (int t = 0; t < NT; ++t) w.T[t] = ld4(tens + (size_t)t * pl + (size_t)(c + D) * Hs + rowL);
            w.fl = *reinterpret_cast<const unsigned*>(a.flags + (size_t)(c + D) * Hs + rowL);
            const size_t off = (size_t)(c + D * L) * Hs + rowL;
            if (MODE < 2) { w.r = ld4(a.xin + off); }
            else {
                w.r = ld4(a.r + off); w.p = ld4(a.p_in + off);
                const size_t oo = (size_t)c * Hs + rowL;               // the column step c outputs
                if (MODE != 3 || XU != 0) { w.xo = ld4(a.x + oo); w.po = ld4(a.p_in + oo); }
                if (MODE == 3 && XU == 2) w.po2 = ld4(a.p_out + oo);
                if (MODE == 3) w.wp = (a.k != 1) ? ld4(a.w_prev + off) : zero4();      // step 1 has no update pending (uniform)
            }
        };
        auto convert = [&](const Raw& w, F4& rnew) -> F4 {     // x of the loaded column (CG: p_new = beta p + r)
            if (MODE < 2) return w.r;
            F4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rnew.e[e] = (MODE == 3) ? fmaf(-alpha_prev, w.wp.e[e], w.r.e[e]) : w.r.e[e];      // Saxpy dc.cu:272, deferred
                o.e[e] = mul_then_add(beta, w.p.e[e], rnew.e[e]);
            }
            return o;
        };
        auto load_x = [&](int col, F4& rnew) -> F4 {           // prologue only (waited immediately)
            const size_t off = (size_t)col * Hs + rowL;
            if (MODE < 2) return ld4(a.xin + off);
            const F4 rv = ld4(a.r + off);
            const F4 pv = ld4(a.p_in + off);
            const F4 wv = (MODE == 3 && a.k != 1) ? ld4(a.w_prev + off) : zero4();
            F4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rnew.e[e] = (MODE == 3) ? fmaf(-alpha_prev, wv.e[e], rv.e[e]) : rv.e[e];
                o.e[e] = mul_then_add(beta, pv.e[e], rnew.e[e]);
            }
            return o;
        };

        F4 X[NX];                      // X[k] = x at column c + D*(k-1): [previous, current, next, next+1, ...] in march order
        F4 R[MODE == 3 ? NX - 1 : 1];  // MODE 3: the updated residual of the columns X[1..NX-1] (stored and summed when a column is output)
        F4 Uprev = zero4(), U0 = zero4(), V0 = zero4(), W0 = zero4();
        unsigned FLm1 = 0u, FL0 = 0u;
        // tensor-recompute mode: M = sum_c g_c Q_c with
        //   Q_c = [[S dx^2 + R00, S dx dy + R01, S dx], [., S dy^2 + R11, S dy], [., ., S]],  dx = xx - x*, dy = yy - y*
        // dy depends on the row only (fixed for the lane), dx on the column only (uniform per step)
        float kS[NC > 0 ? NC : 1], kX[NC > 0 ? NC : 1], kR00[NC > 0 ? NC : 1], kR01[NC > 0 ? NC : 1];
        float cSdy[NC > 0 ? NC : 1][4], cQ11[NC > 0 ? NC : 1][4];          // S dy  and  S dy^2 + R11
        if (NC > 0) {
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) {
                const float* k8 = a.consts + ch * 8;           // uniform -> scalar loads
                kS[ch] = k8[0]; kX[ch] = k8[1]; kR00[ch] = k8[3]; kR01[ch] = k8[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dy = ((float)(a.i_lo + row0 - PAD + e) - a.cy) - k8[2];   // yy - y*
                    cSdy[ch][e] = k8[0] * dy;
                    cQ11[ch][e] = (k8[0] * dy) * dy + k8[5];
                }
            }
        }
        F4 S = zero4();                // block sums of the current column group
        X[0] = zero4();
        const int c_first = (D > 0) ? c0 - 2 : c0 + TJ + 1;      // two priming steps before the strip
        const int c_final = (D > 0) ? c0 + TJ - 1 : c0;          // last output column
#pragma unroll
        for (int k = 1; k + 1 < NX; ++k) X[k] = load_x(c_first + D * (k - 1), R[MODE == 3 ? k - 1 : 0]);
        X[NX - 1] = zero4();
        // Software pipeline: the loads of step c+1 are issued before the arithmetic of step c, into two
        // alternating buffer sets (the loop is unrolled by two, so no in-flight register is ever copied).
        // The last step re-reads its own columns (cache hits) instead of branching, which keeps the vmcnt
        // bookkeeping exact.  (A distance-3 pipeline with four buffer sets measured 8 % slower.)
        Raw bufA, bufB;
        issue(bufA, c_first);

        auto step = [&](const int m, const Raw& cur, Raw& nxt) {      // m = march index: -2, -1 prime, 0..TJ-1 output
            const int c = c_first + D * (m + 2);
            issue(nxt, (D > 0) ? min(c + 1, c_final) : max(c - 1, c_final));
            X[NX - 1] = convert(cur, R[MODE == 3 ? NX - 2 : 0]);             // x of column c+D*L
            const unsigned FL1 = cur.fl;          // structure bytes of the next column (c+D)
            const F4* Mc = cur.T;                 // tensor data of the next column
            // ---- (u,v,w) of column c+1 ----------------------------------------------------------
            // Wave-uniform fast path: when every structure byte the wave touches is either "interior"
            // (masked, forward in x and y, inside a KT block) or empty, all selects collapse to plain
            // arithmetic.  Empty pixels have M = 0, so their u, v, w vanish by themselves.  The results are
            // bit-identical to the general path (the same operations minus additions of +0).
            const bool plain1 = wave_all(is_plain(FL1));
            F4 U1, V1, W1;
            {
                const F4& xc = X[2];                       // next column n = c+D
                const F4& xl = (D > 0) ? X[1] : X[3];     // column n-1
                const F4& xr = (D > 0) ? X[3] : X[1];     // column n+1
                const float x_up = dpp_from_prev_lane(xc.e[3]);
                const float x_dn = dpp_from_next_lane(xc.e[0]);
                float q00[NC > 0 ? NC : 1], q02[NC > 0 ? NC : 1], dxc[NC > 0 ? NC : 1];
                if (NC > 0) {
                    const float xxv = (float)(a.j_lo + (c + D) - PAD) - a.cx;      // xx of the next column (meshgrid: j - K[6])
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) {
                        dxc[ch] = xxv - kX[ch];
                        q02[ch] = kS[ch] * dxc[ch];                                 // S dx
                        q00[ch] = q02[ch] * dxc[ch] + kR00[ch];                     // S dx^2 + R00
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float m0, m1, m2, m3, m4, m5;
                    if (NC > 0) {
                        m0 = m1 = m2 = m3 = m4 = m5 = 0.f;
#pragma unroll
                        for (int ch = 0; ch < NC; ++ch) {
                            const float g = Mc[ch].e[e];
                            m0 = fmaf(g, q00[ch], m0);
                            m1 = fmaf(g, fmaf(dxc[ch], cSdy[ch][e], kR01[ch]), m1);     // S dx dy + R01
                            m2 = fmaf(g, q02[ch], m2);
                            m3 = fmaf(g, cQ11[ch][e], m3);
                            m4 = fmaf(g, cSdy[ch][e], m4);
                            m5 = fmaf(g, kS[ch], m5);
                        }
                    } else {
                        m0 = Mc[0].e[e]; m1 = Mc[1 % NT].e[e]; m2 = Mc[2 % NT].e[e]; m3 = Mc[3 % NT].e[e]; m4 = Mc[4 % NT].e[e]; m5 = Mc[5 % NT].e[e];
                    }
                    const float xv = xc.e[e];
                    const float up = (e == 0) ? x_up : xc.e[e > 0 ? e - 1 : 0];
                    const float dn = (e == 3) ? x_dn : xc.e[e < 3 ? e + 1 : 3];
                    float gx, gy;
                    if (plain1) {
                        gx = xr.e[e] - xv;
                        gy = dn - xv;
                    } else {
                        // forward / backward are exclusive (SRPS.cu:39-46, 31-38): at most one term survives
                        gx = (e == 0) ? if_bit<B_FX>(xr.e[e] - xv, FL1) + if_bit<B_BX>(xv - xl.e[e], FL1)
                           : (e == 1) ? if_bit<B_FX + 8>(xr.e[e] - xv, FL1) + if_bit<B_BX + 8>(xv - xl.e[e], FL1)
                           : (e == 2) ? if_bit<B_FX + 16>(xr.e[e] - xv, FL1) + if_bit<B_BX + 16>(xv - xl.e[e], FL1)
                                      : if_bit<B_FX + 24>(xr.e[e] - xv, FL1) + if_bit<B_BX + 24>(xv - xl.e[e], FL1);
                        gy = (e == 0) ? if_bit<B_FY>(dn - xv, FL1) + if_bit<B_BY>(xv - up, FL1)
                           : (e == 1) ? if_bit<B_FY + 8>(dn - xv, FL1) + if_bit<B_BY + 8>(xv - up, FL1)
                           : (e == 2) ? if_bit<B_FY + 16>(dn - xv, FL1) + if_bit<B_BY + 16>(xv - up, FL1)
                                      : if_bit<B_FY + 24>(dn - xv, FL1) + if_bit<B_BY + 24>(xv - up, FL1);
                    }
                    U1.e[e] = m0 * gx + m1 * gy + m2 * xv;
                    V1.e[e] = m1 * gx + m3 * gy + m4 * xv;
                    W1.e[e] = m2 * gx + m4 * gy + m5 * xv;
                }
            }
            // ---- output column c ----------------------------------------------------------------
            if (m >= 0) {
                // left / right neighbours of the output column in march terms
                const F4& Uleft = (D > 0) ? Uprev : U1;
                const F4& Uright = (D > 0) ? U1 : Uprev;
                const unsigned FLleft = (D > 0) ? FLm1 : FL1;
                const unsigned FLright = (D > 0) ? FL1 : FLm1;
                if ((m & (SF - 1)) == 0) {                   // first visited column of an sf-group: block sums
                    F4 cs = zero4();
#pragma unroll
                    for (int d = 0; d < SF; ++d)
#pragma unroll
                        for (int e = 0; e < 4; ++e) cs.e[e] += X[1 + d].e[e];
                    if (SF == 1) S = cs;
                    else if (SF == 2) { S.e[0] = S.e[1] = cs.e[0] + cs.e[1]; S.e[2] = S.e[3] = cs.e[2] + cs.e[3]; }
                    else { const float t = (cs.e[0] + cs.e[1]) + (cs.e[2] + cs.e[3]); S.e[0] = S.e[1] = S.e[2] = S.e[3] = t; }
                }
                const float v_up = dpp_from_prev_lane(V0.e[3]);
                const float v_dn = dpp_from_next_lane(V0.e[0]);
                const unsigned fl_up = dpp_from_prev_lane(FL0);     // its byte 3 = row above this lane's first row
                const unsigned fl_dn = dpp_from_next_lane(FL0);     // its byte 0 = row below this lane's last row
                F4 acc;
                if (wave_all(is_plain(FLleft) && is_plain(FL0) && is_plain(FLright) && is_plain(fl_up) && is_plain(fl_dn))) {
                    // interior: own rows are forward (-u, -v), left / upper neighbours are forward or empty
                    // (+u_left, +v_up; empty ones are zero), right / lower neighbours are never backward
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float vu = (e == 0) ? v_up : V0.e[e > 0 ? e - 1 : 0];
                        float t = W0.e[e];
                        t += 0.f - U0.e[e];
                        t += Uleft.e[e];
                        t += 0.f - V0.e[e];
                        t += vu;
                        t *= a.lambda;
                        t += S.e[e] * a.inv_sf4;
                        acc.e[e] = t;
                    }
                    // empty pixels of the wave (outside the mask) must stay zero
                    acc.e[0] = if_bit<0>(acc.e[0], FL0); acc.e[1] = if_bit<8>(acc.e[1], FL0);
                    acc.e[2] = if_bit<16>(acc.e[2], FL0); acc.e[3] = if_bit<24>(acc.e[3], FL0);
                } else {
#define SRPS_ROW(EE, FUW, FUB, FDW, FDB, VU, VD)                                                   \
    {                                                                                              \
        float t = W0.e[EE];                                                                        \
        t += if_bit<B_BX + 8 * EE>(U0.e[EE], FL0) - if_bit<B_FX + 8 * EE>(U0.e[EE], FL0);          \
        t += if_bit<B_FX + 8 * EE>(Uleft.e[EE], FLleft);                                           \
        t -= if_bit<B_BX + 8 * EE>(Uright.e[EE], FLright);                                         \
        t += if_bit<B_BY + 8 * EE>(V0.e[EE], FL0) - if_bit<B_FY + 8 * EE>(V0.e[EE], FL0);          \
        t += if_bit<B_FY + FUB>(VU, FUW);                                                          \
        t -= if_bit<B_BY + FDB>(VD, FDW);                                                          \
        t *= a.lambda;                                                                             \
        t += if_bit<B_KB + 8 * EE>(S.e[EE] * a.inv_sf4, FL0);                                      \
        acc.e[EE] = t;                                                                             \
    }
                    SRPS_ROW(0, fl_up, 24, FL0, 8, v_up, V0.e[1])
                    SRPS_ROW(1, FL0, 0, FL0, 16, V0.e[0], V0.e[2])
                    SRPS_ROW(2, FL0, 8, FL0, 24, V0.e[1], V0.e[3])
                    SRPS_ROW(3, FL0, 16, fl_dn, 0, V0.e[2], v_dn)
#undef SRPS_ROW
                }
                // a strip wider than what is left of the (view of the) grid computes columns beyond it: zeros on a whole grid, a
                // neighbouring rank's columns on a strip view -- neither stored nor summed
                if (owned && (unsigned)(c - PAD) < (unsigned)a.Wg) {
                    const size_t off = (size_t)c * Hs + row0;
                    if (MODE == 0) {
                        st4(a.out + off, acc);
                    } else if (MODE == 1) {
                        F4 rv = ld4(a.r + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { rv.e[e] -= acc.e[e]; red = fmaf(rv.e[e], rv.e[e], red); }
                        st4(a.r + off, rv);
                    } else {
                        st4(a.p_out + off, X[1]);
                        st4(a.out + off, acc);
                        if (MODE == 3) {
                            st4(a.r_out + off, R[0]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float rv = R[0].e[e];
                                red_rw = fmaf(rv, acc.e[e], red_rw); red_ww = fmaf(acc.e[e], acc.e[e], red_ww); red_rr = fmaf(rv, rv, red_rr);
                            }
                        }
                        if (MODE == 3 && XU == 2) {
                            F4 xn;
#pragma unroll
                            for (int e = 0; e < 4; ++e) xn.e[e] = fmaf(alpha_prev, cur.po.e[e], fmaf(alpha_pp, cur.po2.e[e], cur.xo.e[e]));   // two Saxpy of dc.cu:270, in their order
                            st4(a.x + off, xn);
                        } else if ((MODE != 3 || XU == 1) && a.k != 1) {      // uniform
                            F4 xn;
#pragma unroll
                            for (int e = 0; e < 4; ++e) xn.e[e] = fmaf(alpha_prev, cur.po.e[e], cur.xo.e[e]);   // Saxpy dc.cu:270
                            st4(a.x + off, xn);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) red = fmaf(X[1].e[e], acc.e[e], red);
                    }
                }
            }
            // ---- shift the windows --------------------------------------------------------------
            Uprev = U0; U0 = U1; V0 = V1; W0 = W1;
            FLm1 = FL0; FL0 = FL1;
#pragma unroll
            for (int k = 0; k + 1 < NX; ++k) X[k] = X[k + 1];
            if (MODE == 3) {
#pragma unroll
                for (int k = 0; k + 2 < NX; ++k) R[MODE == 3 ? k : 0] = R[MODE == 3 ? k + 1 : 0];
            }
        };
        for (int m = -2; m < TJ; m += 2) {
            step(m, bufA, bufB);
            step(m + 1, bufB, bufA);
        }
    };
    if (item < a.n_items) {
        // snake = 2: the direction also alternates from STEP to step -- what a wave touched last in step k (the far end of its strip, still in
        // the Infinity Cache: a step's planes are three times its size) is what it touches first in step k + 1
        const int flip = (a.snake == 2 && MODE == 3) ? (a.k & 1) : 0;
        if (a.snake && ((strip ^ flip) & 1)) body(std::integral_constant<int, -1>{});
        else body(std::integral_constant<int, 1>{});
    }
    if (MODE == 3) {
        float v4[4] = {red, red_rw, red_ww, red_rr};
        block_sum4(v4, sm4);
        if (threadIdx.x < 4) a.part4_out[(size_t)threadIdx.x * a.n_part + blockIdx.x] = sm4[threadIdx.x];
    } else if (MODE != 0) {
        const float t = block_sum(red, sm);
        if (threadIdx.x == 0) a.part_out[blockIdx.x] = t;
    }
}

// ---- host side ---------------------------------------------------------------------------------

// channels of the tensor-recompute form in use (0 = stream the stored 6-plane tensor)
int march_recompute_channels(const srps_ctx* ctx) {
    if (!ctx->tensor_recompute) return 0;
    const int C = ctx->grid.tensor_channels;
    return (C == 1 || C == 3) ? C : 0;      // 0 also when the last assembly did not produce the g_c planes
}

bool march_supported(const srps_ctx* ctx) {
    const int sf = ctx->grid.sf;
    return sf == 1 || sf == 2 || sf == 4;
}

// Decomposition: segments of `own` rows (multiple of 4, at most 248 = 62 owned lanes x 4) and strips of
// `tj` columns (multiple of 4).  A wave runs tj+2 column steps and the kernel lasts as long as the most
// loaded SIMD: with W waves on 1024 SIMDs that is ceil(W/1024) rounds of tj+2 steps.  tj <= 0 picks the
// width that minimises rounds * (tj + 2) (measured: 2048^2 -> 1152 waves at tj=16 take 42 us, 927 waves
// at tj=20 take 32 us).
void march_plan(Grid& G, int tj, int num_cus) {
    const int nseg0 = std::max(1, cdiv(G.Hg, 248));
    int own = cdiv(G.Hg, nseg0);
    own = ((own + 3) / 4) * 4;
    G.seg_rows = own;
    G.n_seg = cdiv(G.Hg, own);
    const int Wv = G.view_w > 0 ? G.view_w : G.Wg;      // a rank of the strip-partitioned CG plans its own columns only
    if (tj <= 0) {
        const int simds = 4 * std::max(1, num_cus);  // 256 CUs x 4 SIMDs on a whole MI355X; fewer under a CU mask or a partition
        long best = -1;
        for (int cand = 4; cand <= 128; cand += 4) {
            const long waves = (long)G.n_seg * cdiv(Wv, cand);
            long cost = ((waves + simds - 1) / simds) * (cand + 2) * 64 + cand;      // ties: narrower strips
            // measured at 4096^2, sf 2 (same box, per CG step): 72 columns (969 waves) 161 us, 76 columns (918 waves) 152 us,
            // 80 / 84: 181 / 176 on a slower box -- widths that are a multiple of 8 columns lose 3 - 6 % against their neighbours;
            // 36 columns (1938 waves: two resident waves per SIMD, 38 steps each) 182 us where 76 columns took 173
            if (cand % 8 == 0) cost += cost / 16;
            if (waves > simds) cost += cost / 16;
            if (best < 0 || cost < best) { best = cost; tj = cand; }
        }
    }
    G.strip_cols = tj;
    G.n_strip = cdiv(Wv, tj);
}

template <int MODE>
static int launch_march(srps_ctx* ctx, MarchArgs& a) {
    Grid& G = ctx->grid;
    a.own = G.seg_rows; a.n_seg = G.n_seg; a.n_strip = G.n_strip; a.n_items = G.n_seg * G.n_strip;
    if (G.view_w > 0) {
        // this rank's columns [view_c0, view_c0 + view_w) of the grid as a grid of their own: every plane has the column stride
        // Hs, so the view is a pointer offset; the columns left and right of it are the halo the neighbours keep current
        const size_t off = (size_t)G.view_c0 * G.Hs;
        auto sh = [&](auto*& ptr) { if (ptr) ptr += off; };
        sh(a.M); sh(a.flags); sh(a.xin); sh(a.p_in); sh(a.p_out); sh(a.r); sh(a.r_out); sh(a.out); sh(a.x); sh(a.G); sh(a.w_prev);
        a.Wg = G.view_w; a.j_lo += G.view_c0;
    }
    const int nb = cdiv(a.n_items, 4);
#define SRPS_MARCH_NC(SF, NCV)                                                                                \
    do {                                                                                                      \
        if constexpr (MODE == 3) {                                                                            \
            if (xu == 0) { hipLaunchKernelGGL((k_apply_march<SF, MODE, NCV, 0>), dim3(nb), dim3(256), 0, ctx->stream, a); break; } \
            if (xu == 2) { hipLaunchKernelGGL((k_apply_march<SF, MODE, NCV, 2>), dim3(nb), dim3(256), 0, ctx->stream, a); break; } \
        }                                                                                                     \
        hipLaunchKernelGGL((k_apply_march<SF, MODE, NCV, 1>), dim3(nb), dim3(256), 0, ctx->stream, a);       \
    } while (0)
#define SRPS_MARCH_TJ(SF)                                                                                     \
    switch (nc) {                                                                                             \
        case 0: SRPS_MARCH_NC(SF, 0); break;                                                                  \
        case 1: SRPS_MARCH_NC(SF, 1); break;                                                                  \
        case 3: SRPS_MARCH_NC(SF, 3); break;                                                                  \
        default: set_error("march kernel: unsupported channel count %d for tensor recompute", nc); return SRPS_ERR_UNSUPPORTED; \
    }
    const int nc = march_recompute_channels(ctx);
    // the two-step x update of the one-launch step: launch 1 has nothing pending (the one-step form's k == 1 case), even launches leave
    // theirs pending, odd launches from 3 on apply two
    const int xu = (MODE == 3 && a.k > 1 && march_x2_on(ctx)) ? ((a.k & 1) ? 2 : 0) : 1;
    (void)xu;
    a.tj = G.strip_cols;
    a.snake = ctx->march_snake;
    switch (G.sf) {
        case 1: SRPS_MARCH_TJ(1) break;
        case 2: SRPS_MARCH_TJ(2) break;
        case 4: SRPS_MARCH_TJ(4) break;
        default: set_error("march kernel: unsupported sf %d", G.sf); return SRPS_ERR_UNSUPPORTED;
    }
#undef SRPS_MARCH_TJ
#undef SRPS_MARCH_NC
    return SRPS_OK;
}

static MarchArgs march_base(srps_ctx* ctx) {
    Grid& G = ctx->grid;
    MarchArgs a;
    memset(&a, 0, sizeof(a));
    a.M = G.d_M; a.flags = G.d_flags; a.Hg = G.Hg; a.Wg = G.Wg; a.Hs = G.Hs; a.Ws = G.Ws; a.plane = G.plane;
    a.lambda = ctx->lambda;
    a.inv_sf4 = 1.0f / ((float)(G.sf * G.sf) * (float)(G.sf * G.sf));
    a.scal = G.d_scal;
    a.part_out = G.d_pw_part;
    a.G = G.d_G; a.consts = G.d_tconsts; a.cx = G.cx; a.cy = G.cy; a.i_lo = G.i_lo; a.j_lo = G.j_lo;
    return a;
}

int march_blocks(const Grid& G) { return cdiv(G.n_seg * G.n_strip, 4); }

// The two-step x update pays where the step is HBM-bound: 4096^2 158.3 -> 152.5 us same box; at 2048^2, where the step's 201 MB sit in the
// Infinity Cache, the alternation of two kernel bodies costs more than the 4 % of bytes it saves (34.6 -> 38.3 us).  Automatic (2): on when
// the step's planes -- g [nc or 6], x, r [2], p [2], omega [2] -- exceed 240 MB.
bool march_x2_on(const srps_ctx* ctx) {
    if (!cg_fused_step(ctx) || ctx->march_x2 == 0) return false;
    if (ctx->march_x2 == 1) return true;
    const int nc = march_recompute_channels(ctx);
    return (size_t)((nc > 0 ? nc : 6) + 7) * ctx->grid.used * sizeof(float) > ((size_t)240 << 20);
}

int march_apply_plain(srps_ctx* ctx, const float* d_in_plane, float* d_out_plane) {
    MarchArgs a = march_base(ctx);
    a.xin = d_in_plane; a.out = d_out_plane;
    SRPS_TRY(launch_march<0>(ctx, a));
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

int march_residual(srps_ctx* ctx) {
    Grid& G = ctx->grid;
    MarchArgs a = march_base(ctx);
    a.xin = G.d_x; a.r = G.d_r;
    if (cg_fused_step(ctx)) a.part_out = G.d_part4 + 3 * (size_t)G.n_part4;      // r.r of the initial residual: slot 3 of the sums [0]
    SRPS_TRY(launch_march<1>(ctx, a));
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// one launch per CG step (MODE 3); the residual launch has left r.r of the initial residual in slot 3 of the partial sums [0]
int march_cg_step(srps_ctx* ctx, int k) {
    Grid& G = ctx->grid;
    MarchArgs a = march_base(ctx);
    float* pbuf[2] = {G.d_p, G.d_p + G.plane};
    float* wbuf[2] = {G.d_w, G.d_w2};
    float* rbuf[2] = {G.d_r, G.d_r2};                     // like p: updated while neighbouring strips read the old values as halo
    a.p_in = (k == 1) ? rbuf[0] : pbuf[(k + 1) & 1];
    a.p_out = pbuf[k & 1]; a.r = rbuf[(k + 1) & 1]; a.r_out = rbuf[k & 1]; a.x = G.d_x;
    a.out = wbuf[k & 1]; a.w_prev = wbuf[(k + 1) & 1];
    a.n_part = G.n_part4;
    a.part4_in = G.d_part4 + (size_t)((k + 1) & 1) * 4 * G.n_part4;
    a.part4_out = G.d_part4 + (size_t)(k & 1) * 4 * G.n_part4;
    a.k = k;
    a.tol2 = ctx->cg_fixed ? -1.f : ctx->cg_tol * ctx->cg_tol;
    a.totals4 = G.d_totals4 ? G.d_totals4 + 4 * ((k + 1) & 1) : nullptr;      // the previous launch's sums over all ranks; null unless the strip-partitioned CG is driving (srps_strips.hip)
    return launch_march<3>(ctx, a);
}

// the block partials a launch of this grid left (p.omega, r.omega, omega.omega, r.r; `which` = the launch's parity, 0 for the
// residual launch) summed in the fixed order into four doubles: what a rank contributes to the all-reduce between two steps
__global__ __launch_bounds__(256) void k_part4_totals(const float* __restrict__ part4, int n_live, int n_part, double* __restrict__ out) {
    __shared__ double smd4[4][4];
    double s4[4];
    sum_partials4(part4, n_live, n_part, s4, smd4);
    if (threadIdx.x < 4) out[threadIdx.x] = s4[threadIdx.x];
}
int march_part4_totals(srps_ctx* ctx, int which, double* d_out) {
    Grid& G = ctx->grid;
    hipLaunchKernelGGL(k_part4_totals, dim3(1), dim3(256), 0, ctx->stream, (const float*)(G.d_part4 + (size_t)(which & 1) * 4 * G.n_part4), march_blocks(G), G.n_part4, d_out);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

int march_cg_apply(srps_ctx* ctx, int k) {
    Grid& G = ctx->grid;
    MarchArgs a = march_base(ctx);
    float* pbuf[2] = {G.d_p, G.d_p + G.plane};
    a.p_in = (k == 1) ? G.d_r : pbuf[(k + 1) & 1];       // step 1: p = r
    a.p_out = pbuf[k & 1]; a.r = G.d_r; a.out = G.d_w; a.x = G.d_x;
    a.rr_part = G.d_rr_part + (size_t)((k - 1) & 1) * G.nb_update; a.n_rr = G.nb_update;
    a.k = k;
    a.tol2 = ctx->cg_fixed ? -1.f : ctx->cg_tol * ctx->cg_tol;
    return launch_march<2>(ctx, a);
}

}  // namespace srps
