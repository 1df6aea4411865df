// kernels_resident.hip -- the truncated depth CG (devicecalls.cu:229-279 on A_ = KT'KT + lambda A'A) as ONE
// persistent launch whose state lives on the chip.
//
// One block of 512 threads per CU owns a tile of 256 rows x 64 columns of the grid (a second shape, 256 threads and
// 256 x 32, is built from this source for smaller grids: see SRPS_RES_NT below); thread (wave w, lane l)
// owns rows 4l..4l+3 of columns 8w..8w+7 and keeps p, r, x and omega of its 32 pixels in registers (two waves
// per SIMD, 256 registers each) for all 101 steps; two of the three g planes sit in LDS (128 KiB), the third is
// re-read every step through the XCD's L2 (2 MB per XCD, it stays there).  2048 x 2048 is exactly 256 tiles.
// Per step nothing else moves:
//   * the dot products are grid-wide sums (device_utils.h grid_sum / grid_sum3: generation-tagged granules of 8 or 16
//     bytes, no read-modify-write atomics); the default form (ONE_SYNC) needs one such sum per step;
//   * a tile needs r on the one-pixel ring around it: every block publishes the values of its four edges as
//     generation-tagged granules (r after the update, or omega before alpha is known in the one-wait form) and
//     keeps its own copy of p on the ring (same recurrence, same bits as the owner);
//   * inside a block, columns cross waves through two 8 KiB LDS buffers and rows cross lanes through DPP.
// The operator is the one of kernels_march.hip; the tensor is applied in factored form (M = E P E', see below) and the
// contributions of the neighbours are added in a different order, so results agree to rounding.
// Grids that need more than one tile per CU fall back to the streaming kernels (kernels_march.hip).
//
// Three things keep the compiler from spilling (it would otherwise need > 500 registers per thread):
//   * an opaque zero (asm) added to every coordinate: the step-invariant tensor terms, masks and LDS reads are
//     not hoisted out of the CG loop;
//   * the column body has no branches and its results are pinned by empty asm statements: otherwise the
//     arithmetic is sunk to the first use of omega, after the loop, together with everything it reads;
//   * structure masks are formed by v_bfe_i32 in assembly (the portable forms become and + compare + select).
#include <unistd.h>
#include <ctime>
#include <random>

#include "srps_internal.h"
#include "device_utils.h"

namespace srps {

namespace {

// Three tile shapes are built from this source: 256 x 64 with 512 threads (two waves per SIMD, 8 columns per thread) for grids
// that need up to one tile per CU at that size; 256 x 32 with 256 threads (kernels_resident_n256.hip: one wave per SIMD, half the
// arithmetic per CU and step) and 256 x 16 with 256 threads and 4 columns per thread (kernels_resident_n256c4.hip: a quarter)
// for smaller grids, which would otherwise leave most CUs idle while a few do all the arithmetic of a step.
#ifndef SRPS_RES_NT
#define SRPS_RES_NT 512
#endif
#ifndef SRPS_RES_CPT
#define SRPS_RES_CPT 8
#endif
#ifndef SRPS_RES_TAG
#define SRPS_RES_TAG SRPS_RES_NT
#endif
#define SRPS_RES_CAT2(a, b) a##b
#define SRPS_RES_CAT(a, b) SRPS_RES_CAT2(a, b)
#define SRPS_RES_NAME(base) SRPS_RES_CAT(base##_n, SRPS_RES_TAG)
constexpr int NT = SRPS_RES_NT, NWV = NT / 64;  // threads, waves per block
constexpr int CPT = SRPS_RES_CPT;              // columns per thread (a power of two, a multiple of sf: 8 or 4)
constexpr int TR = 256, TC = CPT * NWV;        // tile rows, columns (64, 32 or 16)
constexpr int RING_COL = TR + 8;               // ring column: rows -4..259 (row r at index r + 4, float4-aligned)
constexpr int RING_ROW = TC + 8;               // ring row: columns -4..TC+3
constexpr int RING = 2 * RING_COL + 2 * RING_ROW;
constexpr int HALO_N = 2 * TR + 2 * TC;        // granules a block publishes per step: first/last column, first/last row
constexpr int B_FX = 1, B_BX = 2, B_FY = 3, B_BY = 4, B_KB = 5;

__device__ __forceinline__ int ring_colL(int row) { return 4 + row; }
__device__ __forceinline__ int ring_colR(int row) { return RING_COL + 4 + row; }
__device__ __forceinline__ int ring_rowT(int col) { return 2 * RING_COL + 4 + col; }
__device__ __forceinline__ int ring_rowB(int col) { return 2 * RING_COL + RING_ROW + 4 + col; }

struct F4 {
    float e[4];
};
__device__ __forceinline__ F4 ld4(const float* __restrict__ p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    F4 r; r.e[0] = t.x; r.e[1] = t.y; r.e[2] = t.z; r.e[3] = t.w;
    return r;
}
__device__ __forceinline__ void st4(float* __restrict__ p, const F4& a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.e[0], a.e[1], a.e[2], a.e[3]);
}
__device__ __forceinline__ F4 zero4() { F4 r; r.e[0] = r.e[1] = r.e[2] = r.e[3] = 0.f; return r; }

__device__ __forceinline__ float dpp_from_prev_lane(float v) {      // lane i <- lane i-1 ; lane 0 <- 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_next_lane(float v) {      // lane i <- lane i+1 ; lane 63 <- 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
// bit BIT of the structure word as an all-ones / all-zeros mask. Written in assembly: the compiler turns the
// portable forms into and + compare + select (three instructions and a scalar register pair per use).
// a value every lane holds alike (read from LDS after a barrier), told to the compiler: in a scalar register the CG loop's exit test is
// uniform -- as a vector value the loop counts as divergent, and everything live behind it (x, 32 registers) is copied aside and back every step
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// lane l of `old` replaced by the (wave-uniform) value s: v_writelane_b32 -- no lane mask, no select
template <int L>
__device__ __forceinline__ float writelane_c(float old, float s) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(s), "n"(L));      // the lane is an inline constant (one scalar operand per instruction)
    return old;
}
template <int BIT>
__device__ __forceinline__ int msk(unsigned flword) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(flword), "n"(BIT));
    return m;
}
__device__ __forceinline__ float andm(float v, int m) { return __int_as_float(__float_as_int(v) & m); }
// a where the mask is all ones, b where it is zero: v_bfi_b32 on three vector registers
__device__ __forceinline__ float selm(int m, float a, float b) { return __int_as_float((m & __float_as_int(a)) | (~m & __float_as_int(b))); }
#define SRPS_MSK(B, e, FL) ((e) == 0 ? msk<(B)>(FL) : (e) == 1 ? msk<(B) + 8>(FL) : (e) == 2 ? msk<(B) + 16>(FL) : msk<(B) + 24>(FL))
__device__ __forceinline__ float if_bit_rt(float v, unsigned flword, int bit) {
    return ((flword >> bit) & 1u) ? v : 0.f;
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }      // v_pk_fma_f32
__device__ __forceinline__ v2f mul2(v2f a, v2f b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ v2f andm2(v2f v, v2i m) { return __builtin_bit_cast(v2f, __builtin_bit_cast(v2i, v) & m); }
// The column body's pair arithmetic: packed (two rows per v_pk_* instruction, the default) or row by row (-DSRPS_RES_UNPACKED=1, an
// experiment that stays for the record).  Two waves share a SIMD here; the issue microbenchmark (tools/valu_issue_bench.hip) has
// v_fma_f32 at 4.56 clocks per instruction with one wave per SIMD and 2.29 with two, v_pk_fma_f32 at 5.5 and 4.67 -- two plain
// instructions of two waves pair up, two packed ones do not -- and the per-wave stamps (tools/resident_wave_stamps.py) show waves
// 0 - 3 through their columns 1.6 us before waves 4 - 7 of the same SIMDs: the older wave runs at its solo speed, the younger one in
// its gaps.  If plain instructions paired up in this kernel as they do in the microbenchmark, a body of plain instructions (except
// where a scalar register is read: the tensor rebuild) would let both waves advance together.  Measured on one box: 8.97 against
// 8.58 us per step (general body 10.35 against 9.90) -- they do not; the packed body stays.
#ifndef SRPS_RES_UNPACKED
#define SRPS_RES_UNPACKED 0
#endif
#ifndef SRPS_RES_EARLY_EDGES
#define SRPS_RES_EARLY_EDGES 0        // measured in round 5 (same box, three rounds): 7.94 - 7.96 us per step against 7.77 - 7.80 with the burst behind the loop (general body 9.87 - 9.93 / 9.73 - 9.78)
#endif
#ifndef SRPS_RES_ROWEDGE_LDS
#define SRPS_RES_ROWEDGE_LDS 0       // measured in round 5 (same box, three rounds): 8.08 - 8.14 us per step against 7.81 - 7.83 -- the one store behind the sums' barrier delays the polling wave's loads
#endif
struct r2f { float x, y; };
struct r2i { int x, y; };
__device__ __forceinline__ r2f operator+(r2f a, r2f b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ r2f operator-(r2f a, r2f b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ r2f operator-(r2f a) { return {-a.x, -a.y}; }
__device__ __forceinline__ r2f fma2(r2f a, r2f b, r2f c) { return {__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ r2f mul2(r2f a, r2f b) {
#pragma clang fp contract(off)
    return {a.x * b.x, a.y * b.y};
}
__device__ __forceinline__ r2f andm2(r2f v, r2i m) { return {andm(v.x, m.x), andm(v.y, m.y)}; }
__device__ __forceinline__ void pin2(v2f& a, v2f& b) { asm("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void pin2(r2f& a, r2f& b) { asm("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y)); }
#if SRPS_RES_UNPACKED
typedef r2f B2;
typedef r2i B2i;
#else
typedef v2f B2;
typedef v2i B2i;
#endif
__device__ __forceinline__ F4 as_f4(v4i v) {
    F4 r; r.e[0] = __int_as_float(v.x); r.e[1] = __int_as_float(v.y); r.e[2] = __int_as_float(v.z); r.e[3] = __int_as_float(v.w);
    return r;
}
__device__ __forceinline__ v4i as_v4i(const F4& a) {
    v4i v; v.x = __float_as_int(a.e[0]); v.y = __float_as_int(a.e[1]); v.z = __float_as_int(a.e[2]); v.w = __float_as_int(a.e[3]);
    return v;
}

// The timing-experiment switch (option "cg_resident_debug": no grid-wide sums, no ring polls, wrong results) is compiled in only
// with -DSRPS_RES_DEBUG: in the shipped kernel it cost scalar registers -- which the CG loop is short of -- and a test per use.
#ifdef SRPS_RES_DEBUG
#define SRPS_RES_DEBUG_ON(A) ((A).debug & 1)
#else
#define SRPS_RES_DEBUG_ON(A) false
#endif
struct ResidentArgs {
    const float* G;            // [NC][plane]
    const uint8_t* flags;      // [plane]
    const float* consts;       // [NC][8]: S, x*, y*, R00, R01, R11
    const float* x;            // [plane] the iterate the solve starts from: read, never written
    float* x_out;              // [plane] the result (another plane: an aborted launch leaves x intact)
    const float* r;            // [plane] right-hand side b (the kernel forms the residual b - A_ x0 itself)
    unsigned long long* ent;   // [2][tiles]          reduction granules, zeroed before the launch
    unsigned long long* ent3;  // [2][256] 16-byte granules: the three-value reduction of the one-sync form
    unsigned long long* halo;  // [tiles][2][HALO_N]  edge granules, zeroed before the launch
    CgScalars* scal;
    int Hs, Ws;
    size_t plane;
    int nbr, nbc;              // tiles per column / per row of tiles
    float lambda, inv_sf4, tol2;
    int max_steps;
    float cx, cy;
    int i_lo, j_lo;
    int debug;                 // timing experiments only: 1 = no grid-wide sums, no ring polls (wrong results)
    unsigned gen_base;         // every generation tag of this launch lies above it (a multiple of 1024: slot parities are unchanged).  The single launch
                               // numbers its launches, so that granules left by EARLIER launches never carry a tag this one waits for and the arrays need
                               // no zeroing per solve (5.4 us + a launch per pass); 0 with freshly zeroed arrays (the group / rank launches)
    unsigned long long spin_ticks;   // budget of the whole launch in s_memrealtime ticks (10 ns): waits give up after it
    const uint8_t* tile_cls;   // [tiles] TILE_* bits per tile (kernels_structure.hip), nullptr: every tile takes the general body
    const uint8_t* tile_occ;   // [tiles] the same array, always: TILE_OCCUPIED says whether a neighbouring tile has a block at all
    const int* tile_list;      // [blocks] the occupied tiles, ascending: block b works on tile tile_list[b]
    // ---- several launches as one grid (GROUP kernels: the ranks of a strip partition, srps_strip_group_solve_resident) ----
    GridPeers grp;             // slots of the whole group's blocks, the other ranks' granule arrays (grp.slot is set per block)
    int grp_list_base;         // index of this rank's first tile in the GROUP's ascending tile list (tile_list holds the rank's own tiles)
    int grp_bc_first, grp_bc_last;       // the rank's first / last column of tiles
    unsigned long long* grp_halo_left;   // the edge-granule arrays of the ranks to the left / right (null: none): same layout, global tile numbers
    unsigned long long* grp_halo_right;
};

// per-channel constants of the tensor-recompute form (uniform)
template <int NC>
struct TensorConsts {
    float kS[NC], kX[NC], kY[NC], kR00[NC], kR01[NC], kR11[NC];
};

#ifdef SRPS_STAMPS
// Development aid (make EXTRA=-DSRPS_STAMPS, then tools/resident_stamps.py): wall-clock stamps (s_memrealtime, 100 MHz)
// taken by thread 0 of every block at the phase boundaries of every CG step, [step][block][16]; stamps 8..10 come from
// inside the three-value reduction.  Costs ~0.7 us per step; never part of the shipped build.
__device__ unsigned long long g_stamps[128 * 256 * 16];
// per wave as well (lane 0 of every wave, the first 64 steps of the first 64 blocks): where the waves of a block are at the barriers
__device__ unsigned long long g_wstamps[64 * 64 * 8 * 16];
#define SRPS_STAMP(ID) do { \
    if (tid == 0 && k < 128) g_stamps[((size_t)k * 256 + blockIdx.x) * 16 + (ID)] = __builtin_amdgcn_s_memrealtime(); \
    if ((tid & 63) == 0 && k < 64 && blockIdx.x < 64) g_wstamps[(((size_t)k * 64 + blockIdx.x) * 8 + (tid >> 6)) * 16 + (ID)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SRPS_STAMP(ID) do { } while (0)
#endif
#ifdef SRPS_STAMPS
#define SRPS_STAMP_PTR ((k < 128) ? &g_stamps[((size_t)k * 256 + blockIdx.x) * 16 + 8] : nullptr)
#else
#define SRPS_STAMP_PTR nullptr
#endif

// RECT: the body for a tile that lies wholly inside the mask and whose ring sides are each wholly masked or wholly empty
// (TILE_RECT; every tile of a full-frame mask, every tile in the interior of an object).  All its pixels are forward
// differences inside complete KT blocks, except -- when the ring below / to the right is empty (TILE_BOTTOM_EMPTY /
// TILE_RIGHT_EMPTY: the tile touches the mask's last row / column) -- the backward differences of row 255 / of the last
// column (SRPS.cu:35-38, 43-46).  No structure bits are decoded: the two edge cases are selects on loop-invariant lane
// masks.  The arithmetic per pixel is that of the general body (the masked-out terms there are additions of 0), so the two
// bodies agree bit for bit up to the sign of zeros.
template <int SF, int NC, bool ONE_SYNC, bool RECT, bool GROUP = false>
__device__ __forceinline__ void resident_body(const ResidentArgs& a, const int tile, const unsigned cls, const GridPeers* gp = nullptr) {
    extern __shared__ float4 lds4[];
    // LDS map: [NC==3: g0, g1 as float4 [CPT][NT]] | ex, ex2 (float4 [NT]) | ring (floats): hp, hg[NC], hfl | sm, sflag | ucol
    constexpr int GL = (NC == 3) ? 2 : 0;                 // g planes kept in LDS
    float4* lg = lds4;                                    // [GL][CPT][NT]
    float4* ex = lds4 + GL * CPT * NT;                    // [NT]
    float4* ex2 = ex + NT;                                // [NT]
    float* hp = reinterpret_cast<float*>(ex2 + NT);       // [RING]
    float* hg = hp + RING;                                // [NC][RING]
    unsigned* hfl = reinterpret_cast<unsigned*>(hg + NC * RING);   // [RING]
    float* sm = reinterpret_cast<float*>(hfl + RING);     // [40]
    int* sflag = reinterpret_cast<int*>(sm + 40);         // [4] block-wide structure summary
    float* ucol = reinterpret_cast<float*>(sflag + 4);    // [2][TR] what the ring columns add to the tile's first / last column
    // [2][TC] the tile's first / last row of the vector whose edges travel (ROWEDGE_LDS, RECT body): in the half of ucol that only a tile with
    // backward differences in x writes and reads -- a RECT tile has none (the LDS is full: 512 bytes more than this do not fit beside the
    // static arrays of the grid-wide sums)
    float* erow = ucol + TR;
    constexpr bool ROWEDGE_LDS = SRPS_RES_ROWEDGE_LDS && RECT && 2 * TC <= TR;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ntile = a.nbr * a.nbc;
    const int bc = tile / a.nbr, br = tile - bc * a.nbr;          // consecutive tiles are vertical neighbours
    const int Hs = a.Hs;
    const size_t pl = a.plane;

    TensorConsts<NC> K;
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
        const float* k8 = a.consts + ch * 8;
        K.kS[ch] = k8[0]; K.kX[ch] = k8[1]; K.kY[ch] = k8[2]; K.kR00[ch] = k8[3]; K.kR01[ch] = k8[4]; K.kR11[ch] = k8[5];
    }

    // Factored form used by the column body: M = E P E', E = [[1,0,-x],[0,1,-y],[0,0,-1]] with (x, y) measured from the
    // mean (xm, ym) of the channels' vertices, P = sum_c g_c T_c, T_c the constant 3 x 3 moment matrix of channel c about
    // that point: T00 = R00 + S ex^2, T01 = R01 + S ex ey, T11 = R11 + S ey^2, T02 = S ex, T12 = S ey, T22 = S
    // (ex, ey = vertex of the channel minus the mean). No per-column / per-row terms are needed.
    float xm = 0.f, ym = 0.f;
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) { xm += K.kX[ch]; ym += K.kY[ch]; }
    xm *= 1.f / NC; ym *= 1.f / NC;
    float T00[NC], T01[NC], T02[NC], T11[NC], T12[NC], T22[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
        const float ex = K.kX[ch] - xm, ey = K.kY[ch] - ym;
        T22[ch] = K.kS[ch]; T02[ch] = K.kS[ch] * ex; T12[ch] = K.kS[ch] * ey;
        T00[ch] = fmaf(T02[ch], ex, K.kR00[ch]); T01[ch] = fmaf(T02[ch], ey, K.kR01[ch]); T11[ch] = fmaf(T12[ch], ey, K.kR11[ch]);
        // uniform, but computed by the vector unit: moved to scalar registers (as VGPR pairs they cost 30 registers)
        T00[ch] = readlane_f(T00[ch], 0); T01[ch] = readlane_f(T01[ch], 0); T02[ch] = readlane_f(T02[ch], 0);
        T11[ch] = readlane_f(T11[ch], 0); T12[ch] = readlane_f(T12[ch], 0);
    }
    // the constants of a channel as three pairs {T00, T01}, {T02, T11}, {T12, T22} for the packed column body
    v2f TP[NC][3];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
        TP[ch][0] = (v2f){T00[ch], T01[ch]}; TP[ch][1] = (v2f){T02[ch], readlane_f(T11[ch], 0)}; TP[ch][2] = (v2f){readlane_f(T12[ch], 0), readlane_f(T22[ch], 0)};
    }
    const float xoff = readlane_f(a.cx + xm, 0), yoff = readlane_f(a.cy + ym, 0);
    // (u, v) = first two components of M (gx, gy, xv) at one pixel with coordinates (xs, ys) from (xoff, yoff)
    auto uv_pixel = [&](auto need_u, const float (&g)[NC], float xs, float ys, float gx, float gy, float xv) -> float {
        constexpr bool NU = decltype(need_u)::value;
        float Pa = 0.f, P01 = 0.f, P02 = 0.f, P12 = 0.f, P22 = 0.f;      // Pa = P00 (u) or P11 (v)
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            Pa = fmaf(g[ch], NU ? T00[ch] : T11[ch], Pa);
            P01 = fmaf(g[ch], T01[ch], P01);
            P02 = fmaf(g[ch], T02[ch], P02);
            P12 = fmaf(g[ch], T12[ch], P12);
            P22 = fmaf(g[ch], T22[ch], P22);
        }
        // every multiply-add spelled out: which products the compiler fuses must not depend on the surrounding code (the two
        // bodies, RECT and general, have to produce the same bits)
        const float t2 = -fmaf(gx, xs, fmaf(gy, ys, xv));
        const float Y2 = fmaf(P02, gx, fmaf(P12, gy, P22 * t2));
        if (NU) return fmaf(-Y2, xs, fmaf(Pa, gx, fmaf(P01, gy, P02 * t2)));
        return fmaf(-Y2, ys, fmaf(P01, gx, fmaf(Pa, gy, P12 * t2)));
    };
    auto xs_of = [&](int gcol) { return (float)(a.j_lo + gcol) - xoff; };
    auto ys_of = [&](int grow) { return (float)(a.i_lo + grow) - yoff; };

    // ---- own pixels -------------------------------------------------------------------------------------
    const int grow0 = br * TR + 4 * lane;                  // grid row of element 0
    // y of the thread's four rows from (xoff, yoff): step-invariant, kept in two register pairs (saves 10 instructions per column)
    const v2f ys01 = {(float)(a.i_lo + grow0) - yoff, (float)(a.i_lo + grow0 + 1) - yoff};
    const v2f ys23 = {(float)(a.i_lo + grow0 + 2) - yoff, (float)(a.i_lo + grow0 + 3) - yoff};
    const int gcol0 = bc * TC + CPT * wave;                  // grid column of column 0
    const int srow0 = grow0 + PAD;
    const bool act = srow0 < Hs;                           // Hs is a multiple of 32: the float4 is inside or outside as a whole
    const int rowL = act ? srow0 : 0;                      // rows 0..3 are the zero halo of every plane
    F4 p[CPT], r[CPT], w[CPT], x[CPT];
    unsigned fl[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const size_t off = (size_t)(gcol0 + c + PAD) * Hs + rowL;
        r[c] = ld4(a.r + off);
        x[c] = ld4(a.x + off);
        fl[c] = RECT ? 0u : *reinterpret_cast<const unsigned*>(a.flags + off);
#pragma unroll
        for (int t = 0; t < GL; ++t) lg[(t * CPT + c) * NT + tid] = *reinterpret_cast<const float4*>(a.G + (size_t)t * pl + off);
        p[c] = zero4(); w[c] = zero4();
    }

    // ---- ring ---------------------------------------------------------------------------------------------
    // ring pixels: 0..257 left column (rows -1..256), 258..515 right column, then the top row (columns -1..TC) and the
    // bottom row; thread t looks after ring pixels t, t + NT, ...
    constexpr int RC_N = TR + 2, RR_N = TC + 2, NRING = 2 * RC_N + 2 * RR_N, RPT = (NRING + NT - 1) / NT;
    // Ring pixel number i sits at index i + 3 / + 9 / + 15 / + 21 of the ring arrays (left column, right column, top row, bottom
    // row: ring_colL .. ring_rowB): the index is recomputed where it is needed -- kept per thread it was spilled to scratch and
    // reloaded (a memory round trip) at the top of every CG step.
    static_assert(RING_COL - RC_N == 6 && RING_ROW - RR_N == 6, "ring layout");
    auto ring_index = [](int i) -> int { return i + (i < RC_N ? 3 : i < 2 * RC_N ? 9 : i < 2 * RC_N + RR_N ? 15 : 21); };
    int hoff[RPT];                                         // granule of the ring pixel in its owner's edge arrays (slot 0), as an
                                                           // offset from a.halo (one register instead of a pointer's two); -1: none
    float rh[RPT];                                         // r on the ring pixel
    unsigned rflags = 0u;                                  // union of the ring pixels' structure bytes
    for (int t = tid; t < RING; t += NT) { hp[t] = 0.f; hfl[t] = 0u; }
    for (int t = tid; t < NC * RING; t += NT) hg[t] = 0.f;
    if (tid < 4) sflag[tid] = 0;
    // every wait below is bounded by one deadline (device_utils.h): a block that never becomes resident ends the launch
    // with an abort flag instead of hanging it; the first grid-wide sum (pass 0) doubles as the census of resident blocks
    spin_guard_init(a.spin_ticks, &a.scal->abort_flags, ABORT_DEPTH);      // here, not at the top: 44 bytes of scratch less
    grid_sum3_prepare();
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int i = tid + q * NT;
        hoff[q] = -1; rh[q] = 0.f;
        if (i >= NRING) continue;
        int kind, u;
        if (i < RC_N) { kind = 0; u = i - 1; }
        else if (i < 2 * RC_N) { kind = 1; u = i - RC_N - 1; }
        else if (i < 2 * RC_N + RR_N) { kind = 2; u = i - 2 * RC_N - 1; }
        else { kind = 3; u = i - 2 * RC_N - RR_N - 1; }
        const int pr = (kind == 0 || kind == 1) ? u : (kind == 2 ? -1 : TR);
        const int pc = (kind == 0) ? -1 : (kind == 1) ? TC : u;
        const int ridx = (kind == 0) ? ring_colL(u) : (kind == 1) ? ring_colR(u) : (kind == 2) ? ring_rowT(u) : ring_rowB(u);      // == ring_index(i)
        const int dbr = pr < 0 ? -1 : (pr >= TR ? 1 : 0), dbc = pc < 0 ? -1 : (pc >= TC ? 1 : 0);
        const int nbr_ = br + dbr, nbc_ = bc + dbc;
        const int lr = pr - TR * dbr, lc = pc - TC * dbc;      // coordinates inside the owning tile
        // a neighbouring tile without a masked pixel has no block: nothing is published for it, its side of the ring is empty
        if (nbr_ >= 0 && nbr_ < a.nbr && nbc_ >= 0 && nbc_ < a.nbc && (a.tile_occ[nbc_ * a.nbr + nbr_] & TILE_OCCUPIED)) {
            const int nt = nbc_ * a.nbr + nbr_;
            int gi;
            if (lc == 0) gi = lr;                               // its first column
            else if (lc == TC - 1) gi = TR + lr;                // last column
            else if (lr == 0) gi = 2 * TR + lc;                 // first row
            else gi = 2 * TR + TC + lc;                         // last row
            hoff[q] = nt * 2 * HALO_N + gi;
        }
        const int srow = br * TR + pr + PAD, scol = bc * TC + pc + PAD;
        if (srow >= 0 && srow < Hs && scol >= 0 && scol < a.Ws) {
            const size_t rstor = (size_t)scol * Hs + srow;
            if (ONE_SYNC) { rh[q] = a.r[rstor]; hp[ridx] = a.x[rstor]; }      // r (= b) and p (= x) on the ring for pass 0
            else rh[q] = a.x[rstor];                        // the residual pass applies the operator to x
            const unsigned f = a.flags[rstor];
            hfl[ridx] = f;
            rflags |= f;
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) hg[ch * RING + ridx] = a.G[(size_t)ch * pl + rstor];
        }
    }
    // structure summary of the block: is there any backward difference in x (own pixels or ring) / in y
    {
        unsigned any = 0u;
#pragma unroll
        for (int c = 0; c < CPT; ++c) any |= fl[c];
        const unsigned rf = rflags;
        const bool bx = ((any | rf * 0x01010101u) & (0x01010101u << B_BX)) != 0u;
        const bool by = ((any | rf * 0x01010101u) & (0x01010101u << B_BY)) != 0u;
        if (bx) sflag[0] = 1;
        if (by) sflag[1] = 1;
    }
    __syncthreads();
    const bool any_bx = RECT ? false : sflag[0] != 0;      // a RECT tile has no backward difference on its ring
    // RECT: the lanes / the wave that hold the mask's last row / last column (loop-invariant lane masks in scalar registers)
    const bool bot63 = RECT && (cls & TILE_BOTTOM_EMPTY) != 0u && lane == 63;
    const bool rightw = RECT && (cls & TILE_RIGHT_EMPTY) != 0u && wave == NWV - 1;
    // The lane masks of the selects inside the CG loop live in VECTOR registers (all ones / zero), and so do the scalars of the
    // element-wise updates (alpha, beta, lambda): an instruction that reads a scalar register -- v_cndmask_b32 with its mask in a
    // register pair, v_fmac_f32 with a scalar factor -- takes 4.5 - 5 clocks of the SIMD, the same instruction on vector
    // registers 2.3 (tools/valu_issue_bench.hip), and scalar registers are what this kernel has fewest of.
    int m_l0 = (lane == 0) ? -1 : 0, m_l63 = (lane == 63) ? -1 : 0, m_bot = bot63 ? -1 : 0, m_right = rightw ? -1 : 0;
    asm volatile("" : "+v"(m_l0), "+v"(m_l63), "+v"(m_bot), "+v"(m_right));
    float lambda_v = a.lambda, inv_sf4_v = a.inv_sf4;
    asm volatile("" : "+v"(lambda_v), "+v"(inv_sf4_v));

    // num_records in bytes as an int: the host refuses planes of 2 GiB and more before the launch (resident_cg)
    const auto g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.G + (size_t)(NC - 1) * pl), 0, (int)(pl * sizeof(float)), 0x00020000);
    unsigned gen = a.gen_base;                             // reduction generation (first is base + 1; no granule of the arrays carries a tag above the base)
    float r1 = 0.f, r0 = 0.f, alpha = 0.f, r1_anchor = 0.f;
    int k = 0;
    // Pass 0 forms the residual r = b - A_ x0 (devicecalls.cu:758) with the same operator code: p := x, then r -= omega.
    // The CG steps k = 1.. follow (dc.cu:252: while r1 > tol^2 and k <= max_iter, max_steps = max_iter + 1).
#pragma unroll
    for (int c = 0; c < CPT; ++c) p[c] = x[c];
    // The pass is written once, as a function of "pass 0 or a CG step" -- a compile-time constant or a run-time flag (below).
    auto cg_pass = [&](auto pass0_c) __attribute__((always_inline)) {
        const bool pass0 = pass0_c.value;
        if (!pass0) ++k;
        SRPS_STAMP(0);
        const bool first = pass0 || k == 1;               // p is taken as it is (x, or r), not updated
        float beta = (ONE_SYNC || first) ? 0.f : r1 / r0;      // dc.cu:262
        asm volatile("" : "+v"(beta));                     // a vector register: see the lane masks above
        // An opaque zero added to every coordinate: without it the compiler hoists the (step-invariant) tensor terms
        // and the LDS reads of g out of the CG loop and keeps ~100 more values per thread alive than there are registers.
        int oz = 0;
        asm volatile("" : "+s"(oz));
        // ---- p = beta p + r, own pixels and ring (pass 0: p = x, set before the loop) ----------------------------------
        // (one-wait form: the previous pass has formed this step's p at its end, where r.r -- and with it beta -- is first known)
        if (!ONE_SYNC) {
            if (!pass0) {
#pragma unroll
                for (int c = 0; c < CPT; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[c].e[e] = scal_then_axpy(beta, p[c].e[e], r[c].e[e]);      // k == 1: beta = 0, p = 0 p + r = r (dc.cu:258)
            }
#pragma unroll
            for (int q = 0; q < RPT; ++q)
                if (tid + q * NT < NRING) {
                    const int ri = ring_index(tid + q * NT);
                    hp[ri] = first ? rh[q] : scal_then_axpy(beta, hp[ri], rh[q]);
                }
        }
        ex[tid] = make_float4(p[0].e[0], p[0].e[1], p[0].e[2], p[0].e[3]);
        // (a RECT tile reads no column to the left across a wave boundary -- backward differences occur in its last column only,
        // whose left neighbour is the thread's own -- : its ex2 carries u below, without a barrier in between)
        if (!RECT) ex2[tid] = make_float4(p[CPT - 1].e[0], p[CPT - 1].e[1], p[CPT - 1].e[2], p[CPT - 1].e[3]);
        __syncthreads();

        SRPS_STAMP(1);
        // ---- omega = A_ p ------------------------------------------------------------------------------------
        w[0] = zero4();                                    // the other columns start from the u of the column to their left
        F4 u3 = zero4();                                   // forward-x u of the last column: goes to the next wave
        F4 u0 = zero4();                                   // backward-x u of the first column: goes to the previous wave
        float ksum[CPT >= 4 ? CPT / 4 : 1] = {};           // SF == 4: sums of the thread's 4 x 4 blocks (CPT = 2 is not built for sf 4)
        F4 S[CPT];
        // the last g plane is streamed one column ahead (read-only, 2 MB per XCD: it stays in the L2)
        // the streamed plane goes through a buffer descriptor: one per-lane row offset (voffset) serves every column,
        // the column offset is a scalar (soffset) -- no per-column address registers
        const unsigned rowLb = (unsigned)rowL * 4u;
        const unsigned colb = (unsigned)__builtin_amdgcn_readfirstlane((gcol0 + PAD + oz) * Hs * 4);
        const unsigned hsb = (unsigned)Hs * 4u;
        F4 gnext = as_f4(__builtin_amdgcn_raw_buffer_load_b128(g_rsrc, rowLb, colb, 0));
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const F4 grc = gnext;
            if (c + 1 < CPT) gnext = as_f4(__builtin_amdgcn_raw_buffer_load_b128(g_rsrc, rowLb, colb + (c + 1) * hsb, 0));
            int ozc = 0;
            asm volatile("" : "+s"(ozc));                  // per column: keeps the column's terms from being formed early
            const float xs = (float)(a.j_lo + gcol0 + c + ozc) - xoff;
            F4 g0v, g1v;
            if (GL == 2) {
                // The thread index goes through an opaque copy made inside the column: otherwise the sixteen addresses (invariant:
                // tid * 16 + a constant) are formed before the CG loop, cannot all stay in registers, and every column starts
                // with a scratch reload -- a memory round trip -- of its address.
                int tidc = tid;
                asm volatile("" : "+v"(tidc));
                const float4 t0 = lg[(0 * CPT + c) * NT + tidc], t1 = lg[(1 * CPT + c) * NT + tidc];
                g0v.e[0] = t0.x; g0v.e[1] = t0.y; g0v.e[2] = t0.z; g0v.e[3] = t0.w;
                g1v.e[0] = t1.x; g1v.e[1] = t1.y; g1v.e[2] = t1.z; g1v.e[3] = t1.w;
            }
            const F4& xc = p[c];
            // the neighbour columns in the next / previous wave (or on the ring) are fetched where they are used
            F4 pedge = zero4();
            if (c == CPT - 1) {                                // compile-time
                const float* src = (wave < NWV - 1) ? reinterpret_cast<const float*>(ex + tid + 64) : hp + ring_colR(4 * lane);
                pedge = ld4(src);
            }
            if (c == 0 && !RECT) {                             // only backward differences read the column to the left
                const float* src = (wave > 0) ? reinterpret_cast<const float*>(ex2 + tid - 64) : hp + ring_colL(4 * lane);
                pedge = ld4(src);
            }
            const F4& xr = (c < CPT - 1) ? p[c < CPT - 1 ? c + 1 : CPT - 1] : pedge;
            const F4& xl = (c > 0) ? p[c > 0 ? c - 1 : 0] : pedge;
            float x_up = dpp_from_prev_lane(xc.e[3]);
            float x_dn = dpp_from_next_lane(xc.e[0]);
            {   // no branches inside the column body: a branch splits the block and the compiler then sinks half of the
                // column's arithmetic (and everything it reads) below the loop
                const float rt = hp[ring_rowT(CPT * wave + c)], rb = hp[ring_rowB(CPT * wave + c)];
                x_up = selm(m_l0, rt, x_up);
                x_dn = selm(m_l63, rb, x_dn);
            }
            unsigned FL = fl[c];
            if (!RECT) asm volatile("" : "+v"(FL));  // opaque: the 16 masks per column derived from it are not worth 16 registers
            float send_dn = 0.f, send_up = 0.f;
            // two rows per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): rows (0,1), then rows (2,3)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e0 = 2 * h, e1 = 2 * h + 1;
                const v2f ys = (h == 0) ? ys01 : ys23;
                // P = sum_c g_c T_c.  The scalar operand of a packed instruction is a register PAIR; written in C the compiler keeps
                // every constant twice (36 scalar registers for 18 constants -- and then spills them to vector-register lanes,
                // reloading them with v_readlane for every column).  In assembly one pair carries two constants and op_sel picks
                // the half both rows multiply with.  Same instruction, same rounding (the first channel as a product: g T + 0).
                v2f P00, P01, P02, P11, P12, P22;
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) {
                    v2f g;
                    if (NC == 3) g = (ch == 0) ? (v2f){g0v.e[e0], g0v.e[e1]} : (ch == 1) ? (v2f){g1v.e[e0], g1v.e[e1]} : (v2f){grc.e[e0], grc.e[e1]};
                    else g = (v2f){grc.e[e0], grc.e[e1]};
                    if (ch == 0) {
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(P00) : "v"(g), "s"(TP[ch][0]));
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(P01) : "v"(g), "s"(TP[ch][0]));
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(P02) : "v"(g), "s"(TP[ch][1]));
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(P11) : "v"(g), "s"(TP[ch][1]));
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(P12) : "v"(g), "s"(TP[ch][2]));
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(P22) : "v"(g), "s"(TP[ch][2]));
                    } else {
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(P00) : "v"(g), "s"(TP[ch][0]));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(P01) : "v"(g), "s"(TP[ch][0]));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(P02) : "v"(g), "s"(TP[ch][1]));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(P11) : "v"(g), "s"(TP[ch][1]));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(P12) : "v"(g), "s"(TP[ch][2]));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(P22) : "v"(g), "s"(TP[ch][2]));
                    }
                }
                const B2 xv = {xc.e[e0], xc.e[e1]};
                const B2 up = {(h == 0) ? x_up : xc.e[1], (h == 0) ? xc.e[0] : xc.e[2]};
                const B2 dn = {(h == 0) ? xc.e[1] : xc.e[3], (h == 0) ? xc.e[2] : x_dn};
                const B2 xrv = {xr.e[e0], xr.e[e1]}, xlv = {xl.e[e0], xl.e[e1]};
                // forward / backward are exclusive (SRPS.cu:39-46, 31-38)
                B2i mfx, mbx, mfy, mby;
                B2 gx, gy;
                if constexpr (RECT) {
                    gx = xrv - xv;
                    if (c == CPT - 1) {                        // the mask's last column: backward (the wave is uniform)
                        const B2 gb = xv - xlv;
                        gx.x = selm(m_right, gb.x, gx.x); gx.y = selm(m_right, gb.y, gx.y);
                    }
                    // row by row: the pair below (rows 1, 2 / row 3 and the next lane's row 0) is not a register pair, and a packed
                    // subtraction costs what two plain ones do (tools/valu_issue_bench.hip) -- without the moves that align its operand
                    gy.x = dn.x - xv.x; gy.y = dn.y - xv.y;
                    asm("" : "+v"(gy.x), "+v"(gy.y));
                    if (h == 1) gy.y = selm(m_bot, gy.x, gy.y);      // the mask's last row: backward = the forward difference of the row above
                } else {
                    mfx = B2i{SRPS_MSK(B_FX, e0, FL), SRPS_MSK(B_FX, e1, FL)}; mbx = B2i{SRPS_MSK(B_BX, e0, FL), SRPS_MSK(B_BX, e1, FL)};
                    mfy = B2i{SRPS_MSK(B_FY, e0, FL), SRPS_MSK(B_FY, e1, FL)}; mby = B2i{SRPS_MSK(B_BY, e0, FL), SRPS_MSK(B_BY, e1, FL)};
                    gx = andm2(xrv - xv, mfx) + andm2(xv - xlv, mbx);
                    gy = andm2(dn - xv, mfy) + andm2(xv - up, mby);
                }
                // every multiply-add spelled out (see uv_pixel)
                const B2 xs2 = {xs, xs}, ysb = {ys.x, ys.y};
                const B2 Q00 = {P00.x, P00.y}, Q01 = {P01.x, P01.y}, Q02 = {P02.x, P02.y}, Q11 = {P11.x, P11.y}, Q12 = {P12.x, P12.y}, Q22 = {P22.x, P22.y};
                const B2 t2 = -fma2(gx, xs2, fma2(gy, ysb, xv));         // E'(gx, gy, x)
                const B2 Y0 = fma2(Q02, t2, fma2(Q00, gx, mul2(Q01, gy)));      // the t2 term last: shortest dependent chain
                const B2 Y1 = fma2(Q12, t2, fma2(Q01, gx, mul2(Q11, gy)));
                const B2 Y2 = fma2(Q22, t2, fma2(Q02, gx, mul2(Q12, gy)));
                const B2 U = fma2(-Y2, xs2, Y0);
                const B2 V = fma2(-Y2, ysb, Y1);
                const B2 W = -Y2;
                B2 fxU, bxU, fyV, byV;
                if constexpr (RECT) {
                    const B2 zero = {0.f, 0.f};
                    // u and v as values of their own, as in the general body (whose bit masks see them): without this the compiler
                    // fuses the multiply-adds that make them into the sums below (aggressive FMA fusion) and rounds differently
                    B2 Uo = U, Vo = V;
                    pin2(Uo, Vo);
                    fxU = Uo; bxU = zero; fyV = Vo; byV = zero;
                    if (c == CPT - 1) { fxU.x = andm(Uo.x, ~m_right); fxU.y = andm(Uo.y, ~m_right); bxU.x = andm(Uo.x, m_right); bxU.y = andm(Uo.y, m_right); }
                    if (h == 1) { fyV.y = andm(Vo.y, ~m_bot); byV.y = andm(Vo.y, m_bot); }
                } else {
                    fxU = andm2(U, mfx); bxU = andm2(U, mbx);
                    fyV = andm2(V, mfy); byV = andm2(V, mby);
                }
                B2 own;                                              // A'(u, v, w) at the pixel itself
                if constexpr (RECT) {
                    // where no backward difference can occur (compile-time: every column but the last, every row but the thread's
                    // last) bxU = byV = 0 and 0 - u = -u: two additions instead of four (the sign of a zero result aside, as
                    // between the two bodies anyway)
                    if (c < CPT - 1) own = W - fxU; else own = W + (bxU - fxU);
                    if (h == 0) own = own - fyV;
                    else { own.x = own.x - fyV.x; own.y = own.y + (byV.y - fyV.y); }
                } else own = W + (bxU - fxU) + (byV - fyV);
                w[c].e[e0] += own.x; w[c].e[e1] += own.y;
                // Dx': +u right of a forward pixel -- the first term of the next column's omega: set, not added to a zero
                if (c < CPT - 1) { w[c < CPT - 1 ? c + 1 : CPT - 1].e[e0] = fxU.x; w[c < CPT - 1 ? c + 1 : CPT - 1].e[e1] = fxU.y; }
                else { u3.e[e0] = fxU.x; u3.e[e1] = fxU.y; }
                if (!RECT || c == CPT - 1) {
                    if (c > 0) { w[c > 0 ? c - 1 : 0].e[e0] -= bxU.x; w[c > 0 ? c - 1 : 0].e[e1] -= bxU.y; }                              //      -u left of a backward pixel
                    else { u0.e[e0] = bxU.x; u0.e[e1] = bxU.y; }
                }
                // Dy': +v below a forward pixel, -v above a backward pixel
                w[c].e[e1] += fyV.x;
                if (h == 0) w[c].e[2] += fyV.y; else send_dn = fyV.y;
                if (!RECT || h == 1) w[c].e[e0] -= byV.y;
                if (!RECT) { if (h == 1) w[c].e[1] -= byV.x; else send_up = byV.x; }
            }
            w[c].e[0] += dpp_from_prev_lane(send_dn);
            if (!RECT) w[c].e[3] -= dpp_from_next_lane(send_up);
            // block sums of KT'KT
            if (SF == 1) S[c] = xc;
            else if (SF == 2) {
                S[c].e[0] = S[c].e[1] = xc.e[0] + xc.e[1];
                S[c].e[2] = S[c].e[3] = xc.e[2] + xc.e[3];
            } else ksum[c / 4] += (xc.e[0] + xc.e[1]) + (xc.e[2] + xc.e[3]);
            // pin what this column produced: otherwise the compiler sinks the arithmetic (with all its inputs) to where omega
            // is first read, after the loop
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                asm volatile("" : "+v"(w[c].e[e]));
                if (c + 1 < CPT) asm volatile("" : "+v"(w[c + 1 < CPT ? c + 1 : c].e[e]));
                if (c > 0) asm volatile("" : "+v"(w[c > 0 ? c - 1 : 0].e[e]));
            }
            asm volatile("" : "+v"(u3.e[0]), "+v"(u3.e[1]), "+v"(u3.e[2]), "+v"(u3.e[3]));
            __builtin_amdgcn_sched_barrier(0);             // one column at a time: interleaving them costs registers
        }
        if (SF == 2) {
#pragma unroll
            for (int c = 0; c < CPT; c += 2)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = S[c].e[e] + S[c + 1].e[e]; S[c].e[e] = S[c + 1].e[e] = t; }
        }
        SRPS_STAMP(2);
        // ring rows: the pixels above row 0 (they act on row 0 when they are forward in y) and below row 255 (on row 255
        // when backward in y). Only lane 0 / lane 63 own the rows they act on, so instead of every lane repeating the
        // work per column, lanes 0..7 take the top ring pixel of columns 0..7 and lanes 8..15 the bottom one; the row-0 /
        // row-255 values travel through v_readlane.
        {
            const int j = lane & (CPT - 1);
            const bool bot = (lane & CPT) != 0;
            float pown = 0.f;                                  // p of the own pixel next to the ring pixel
            // lane c takes row 0 of column c (from lane 0), lane CPT + c row 255 (from lane 63): written straight into the lanes
            // (as selects, the eight lane masks (j == c) lived in scalar register pairs that were spilled and reloaded every step)
#define SRPS_ROWPICK(C)                                                                                        \
    if (C < CPT) {                                                                                             \
        const float t0 = readlane_f(p[C < CPT ? C : 0].e[0], 0), t3 = readlane_f(p[C < CPT ? C : 0].e[3], 63); \
        pown = writelane_c<C>(pown, t0);                                                                       \
        pown = writelane_c<(C < CPT ? C + CPT : C)>(pown, t3);                                                 \
    }
            SRPS_ROWPICK(0) SRPS_ROWPICK(1) SRPS_ROWPICK(2) SRPS_ROWPICK(3) SRPS_ROWPICK(4) SRPS_ROWPICK(5) SRPS_ROWPICK(6) SRPS_ROWPICK(7)
#undef SRPS_ROWPICK
            const int cc = CPT * wave + j;
            const int ir = bot ? ring_rowB(cc) : ring_rowT(cc);
            const unsigned f = hfl[ir];
            float g[NC];
            const float xs = xs_of(gcol0 + j + oz), ys = ys_of((bot ? br * TR + TR : br * TR - 1) + oz);
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) g[ch] = hg[ch * RING + ir];
            const float xv = hp[ir];
            const float gx = if_bit_rt(hp[ir + 1] - xv, f, B_FX) + if_bit_rt(xv - hp[ir - 1], f, B_BX);
            const float gy = bot ? (xv - pown) : (pown - xv);     // bottom: backward in y; top: forward in y
            const float V = uv_pixel(std::false_type{}, g, xs, ys, gx, gy, xv);
            const float cv = if_bit_rt(V, f, bot ? B_BY : B_FY);
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const float tt = readlane_f(cv, c), tb = readlane_f(cv, CPT + c);
                w[c].e[0] += andm(tt, m_l0);
                w[c].e[3] -= andm(tb, m_l63);
            }
        }
        // ring columns: the pixels left of column 0 (they act on column 0 when forward in x) and right of the last column (on it,
        // when backward in x).  One pixel per lane, spread over the waves -- left ring rows 0..255 on threads 0..255, the right
        // ring on the next 256 (a second round where the block has only 256 threads) -- instead of four pixels per lane on the
        // wave that owns the column: that wave's SIMD was the last to reach the barrier below by three evaluations.  The values
        // reach the owner through LDS (ucol), behind that barrier; the owner's p comes from the columns in ex / ex2.
        {
            const float* pL = reinterpret_cast<const float*>(ex);                        // p of column 0, rows 0..255 (wave 0's entries)
            const float* pR = reinterpret_cast<const float*>(ex2 + 64 * (NWV - 1));      // p of the last column (last wave's entries)
#pragma unroll
            for (int rr0 = 0; rr0 < 2 * TR; rr0 += NT) {
                const int side = __builtin_amdgcn_readfirstlane((rr0 + 64 * wave) >= TR ? 1 : 0);      // uniform per wave: 0 left, 1 right
                if (side == 1 && !any_bx) continue;
                const int row = rr0 + tid - side * TR;
                const int ir = side ? ring_colR(row) : ring_colL(row);
                const unsigned f = hfl[ir];
                float g[NC];
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) g[ch] = hg[ch * RING + ir];
                const float xs = xs_of((side ? bc * TC + TC : bc * TC - 1) + oz), ys = ys_of(br * TR + row + oz);
                const float xv = hp[ir];
                const float gx = side ? xv - pR[row] : pL[row] - xv;      // left: used only if the ring pixel is forward in x; right: backward
                const float gy = if_bit_rt(hp[ir + 1] - xv, f, B_FY) + if_bit_rt(xv - hp[ir - 1], f, B_BY);
                const float U = uv_pixel(std::true_type{}, g, xs, ys, gx, gy, xv);
                ucol[side * TR + row] = side ? if_bit_rt(U, f, B_BX) : if_bit_rt(U, f, B_FX);
            }
        }
        SRPS_STAMP(3);
        // u across the wave boundaries
        float4* uex = RECT ? ex2 : ex;                     // RECT: ex2 is free (above) -- one barrier instead of two
        if (!RECT) __syncthreads();                        // every wave has read the p columns in ex / ex2
        SRPS_STAMP(14);
        uex[tid] = make_float4(u3.e[0], u3.e[1], u3.e[2], u3.e[3]);
        if (any_bx) ex2[tid] = make_float4(u0.e[0], u0.e[1], u0.e[2], u0.e[3]);
        __syncthreads();
        SRPS_STAMP(15);
        if (wave > 0) {
            const float4 t = uex[tid - 64];
            w[0].e[0] += t.x; w[0].e[1] += t.y; w[0].e[2] += t.z; w[0].e[3] += t.w;
        } else {                                               // the left ring column acts on column 0
            const float4 t = reinterpret_cast<const float4*>(ucol)[lane];
            w[0].e[0] += t.x; w[0].e[1] += t.y; w[0].e[2] += t.z; w[0].e[3] += t.w;
        }
        if (any_bx) {
            if (wave < NWV - 1) {
                const float4 t = ex2[tid + 64];
                w[CPT - 1].e[0] -= t.x; w[CPT - 1].e[1] -= t.y; w[CPT - 1].e[2] -= t.z; w[CPT - 1].e[3] -= t.w;
            } else {                                           // the right ring column on the last column
                const float4 t = reinterpret_cast<const float4*>(ucol + TR)[lane];
                w[CPT - 1].e[0] -= t.x; w[CPT - 1].e[1] -= t.y; w[CPT - 1].e[2] -= t.z; w[CPT - 1].e[3] -= t.w;
            }
        }
        // omega = lambda * (...) + KT'KT p ; partial p.omega (and, one-sync form, r.omega and omega.omega)
        float red = 0.f, red_rw = 0.f, red_ww = 0.f;
        unsigned flk[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) { flk[c] = fl[c]; if (!RECT) asm volatile("" : "+v"(flk[c])); }
        const unsigned hgen = a.gen_base + (unsigned)k + 1u;      // edge generation: base + k + 1 (never a tag an earlier launch left), slot (k + 1) & 1
        // EARLY_EDGES (one-wait form, single launch): the edge granules of omega leave column by column, as soon as a column's final values
        // exist, between the multiply-adds of the columns that follow -- not as a burst of six store instructions per wave behind the
        // loop (0.45 us of the chain between "finalize done" and the block's sums, profiles/r05_resident_stamps.txt)
        constexpr bool EARLY_EDGES = SRPS_RES_EARLY_EDGES && ONE_SYNC && !GROUP;
        unsigned long long* const hb_e = a.halo + ((size_t)tile * 2 + (hgen & 1u)) * HALO_N;
        auto store2_e = [&](unsigned long long* d, float v0, float v1) {
            const srps_v4u g = {__float_as_uint(v0), hgen, __float_as_uint(v1), hgen};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(d), "v"(g) : "memory");
        };
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = ((SF == 4) ? ksum[c / 4] : S[c].e[e]) * inv_sf4_v;
                if (RECT) asm("" : "+v"(s));               // a value of its own, as behind the general body's mask (see u, v above)
                w[c].e[e] = w[c].e[e] * lambda_v + (RECT ? s : andm(s, SRPS_MSK(B_KB, e, flk[c])));
                red = fmaf(p[c].e[e], w[c].e[e], red);
                if (ONE_SYNC) { red_rw = fmaf(r[c].e[e], w[c].e[e], red_rw); red_ww = fmaf(w[c].e[e], w[c].e[e], red_ww); }
            }
            if constexpr (EARLY_EDGES) {
                if (c == 0 && wave == 0) {                           // the tile's first column
                    unsigned long long* d = hb_e + 4 * lane;
                    store2_e(d, w[0].e[0], w[0].e[1]); store2_e(d + 2, w[0].e[2], w[0].e[3]);
                }
                if (c == CPT - 1 && wave == NWV - 1) {               // its last column
                    unsigned long long* d = hb_e + TR + 4 * lane;
                    store2_e(d, w[CPT - 1].e[0], w[CPT - 1].e[1]); store2_e(d + 2, w[CPT - 1].e[2], w[CPT - 1].e[3]);
                }
                if ((c & 1) && (lane == 0 || lane == 63)) {          // first / last row, two columns per store
                    unsigned long long* d = hb_e + 2 * TR + (lane == 0 ? 0 : TC) + CPT * wave + (c - 1);
                    store2_e(d, lane == 0 ? w[c - 1 >= 0 ? c - 1 : 0].e[0] : w[c - 1 >= 0 ? c - 1 : 0].e[3], lane == 0 ? w[c].e[0] : w[c].e[3]);
                }
            }
        }
        // the tile's edges of `src` as generation-tagged granules
        auto publish_edges = [&](const F4 (&src)[CPT]) {
            // two granules {value, generation} per 16-byte store (each granule is read on its own, as 8 bytes): half the store
            // instructions of one per granule, and every store with data registers of its own
            unsigned long long* hb = a.halo + ((size_t)tile * 2 + (hgen & 1u)) * HALO_N;
            auto store2 = [&](unsigned long long* d, float v0, float v1) {
                const srps_v4u g = {__float_as_uint(v0), hgen, __float_as_uint(v1), hgen};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(d), "v"(g) : "memory");
            };
            if (wave == 0 || wave == NWV - 1) {
                const F4& rc = (wave == 0) ? src[0] : src[CPT - 1];
                unsigned long long* d = hb + (wave == 0 ? 0 : TR) + 4 * lane;
                if constexpr (GROUP) {
                    // a tile in the strip's first / last column of tiles: its first / last column is the ring column of a tile on the
                    // neighbouring rank -- the same granules, also into that rank's array (system scope: it may be another device's memory)
                    auto store2s = [&](unsigned long long* dd, float v0, float v1) {
                        const srps_v4u g = {__float_as_uint(v0), hgen, __float_as_uint(v1), hgen};
                        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(dd), "v"(g) : "memory");
                    };
                    store2s(d, rc.e[0], rc.e[1]);
                    store2s(d + 2, rc.e[2], rc.e[3]);
                    unsigned long long* peer = (wave == 0) ? ((bc == a.grp_bc_first) ? a.grp_halo_left : nullptr) : ((bc == a.grp_bc_last) ? a.grp_halo_right : nullptr);
                    if (peer) {                                            // wave-uniform
                        unsigned long long* d2 = peer + (d - a.halo);
                        store2s(d2, rc.e[0], rc.e[1]);
                        store2s(d2 + 2, rc.e[2], rc.e[3]);
                    }
                } else {
                store2(d, rc.e[0], rc.e[1]);
                store2(d + 2, rc.e[2], rc.e[3]);
                }
            }
            if constexpr (ROWEDGE_LDS) {
                // The row edges -- lanes 0 and 63 of every wave, CPT values each -- used to leave as CPT / 2 store instructions per wave with
                // one or two active lanes each: 32 sparse write-through stores per block in front of the block's sums.  They go to LDS here
                // (two ds_write_b128 per wave) and out as ONE store instruction of one wave behind the sums' barrier (publish_row_edges).
                if (lane == 0 || lane == 63) {
                    float* d = erow + (lane == 0 ? 0 : TC) + CPT * wave;
#pragma unroll
                    for (int c = 0; c < CPT; ++c) d[c] = lane == 0 ? src[c].e[0] : src[c].e[3];
                }
            } else {
                if (lane == 0 || lane == 63) {
                    unsigned long long* d = hb + 2 * TR + (lane == 0 ? 0 : TC) + CPT * wave;
#pragma unroll
                    for (int c = 0; c < CPT; c += 2)
                        store2(d + c, lane == 0 ? src[c].e[0] : src[c].e[3], lane == 0 ? src[c + 1].e[0] : src[c + 1].e[3]);
                }
            }
        };
        // behind a barrier that follows publish_edges (the one inside the sums' publish): wave 1 -- not the polling wave -- stores the 2 TC
        // row-edge granules, two per lane, TC lanes: the first-row region and the last-row region of the tile's edge array are adjacent
        auto publish_row_edges = [&]() {
            if constexpr (ROWEDGE_LDS) {
            if (wave == 1 && lane < TC) {
                unsigned long long* hb = a.halo + ((size_t)tile * 2 + (hgen & 1u)) * HALO_N;
                const float2 v = reinterpret_cast<const float2*>(erow)[lane];       // values 2 lane, 2 lane + 1 of [first row | last row]
                const srps_v4u g = {__float_as_uint(v.x), hgen, __float_as_uint(v.y), hgen};
                unsigned long long* d = hb + 2 * TR + 2 * lane;
                if constexpr (GROUP) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(d), "v"(g) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(d), "v"(g) : "memory");
            }
            }
        };
        // the ring values of this generation: requested early (request), waited for late (await)
        unsigned long long hv[RPT];
        auto request_ring = [&]() {
#pragma unroll
            for (int q = 0; q < RPT; ++q)
                hv[q] = (hoff[q] >= 0 && !(SRPS_RES_DEBUG_ON(a)))
                            ? (GROUP ? __hip_atomic_load(a.halo + hoff[q] + (hgen & 1u) * HALO_N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                     : __hip_atomic_load(a.halo + hoff[q] + (hgen & 1u) * HALO_N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0ull;
        };
        auto await_ring = [&](float (&val)[RPT]) {
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                val[q] = 0.f;
                if (hoff[q] >= 0 && !(SRPS_RES_DEBUG_ON(a))) {
                    const unsigned long long* s = a.halo + hoff[q] + (hgen & 1u) * HALO_N;
                    while ((unsigned)(hv[q] >> 32) != hgen) {
                        // the clock is read by the scalar unit (no counter register); uniform for the lanes still waiting
                        if (spin_deadline_passed()) { spin_give_up(-1, hgen); break; }
                        __builtin_amdgcn_s_sleep(1);
                        hv[q] = GROUP ? __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    val[q] = __uint_as_float((unsigned)hv[q]);
                }
            }
        };
        if constexpr (ONE_SYNC) {
            // One grid-wide wait per step.  The edges of omega travel BEFORE alpha is known: a neighbour applies
            // r_ring -= alpha omega_ring itself, with the owner's instruction (same bits).  r.r of the updated residual is
            // not summed again: |r - alpha omega|^2 = r.r - 2 alpha r.omega + alpha^2 omega.omega, and the three products on
            // the right are reduced together with p.omega (float per thread and wave, double from there on).  Guard: when
            // that difference cancels more than two digits the direct sum is taken (one more wait, rare).
            SRPS_STAMP(4);
            if constexpr (!EARLY_EDGES) publish_edges(w);
            SRPS_STAMP(11);
            float wr[RPT];
            if (pass0) {
                red = 0.f;
#pragma unroll
                for (int c = 0; c < CPT; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        r[c].e[e] -= w[c].e[e];                                  // r = b - A_ x   (dc.cu:758)
                        red = fmaf(r[c].e[e], r[c].e[e], red);
                    }
                ++gen;
                grid_sum_publish(red, a.ent, gen, sm, gp);
                publish_row_edges();
                request_ring();
                r1 = uniform_f((SRPS_RES_DEBUG_ON(a)) ? 1.f : grid_sum_collect(a.ent, gen, sm, gp));
                r1_anchor = r1;
                await_ring(wr);
#pragma unroll
                for (int q = 0; q < RPT; ++q) rh[q] -= wr[q];
            } else {
                ++gen;
                grid_sum3_publish<NWV, true>(red, red_rw, red_ww, a.ent3, gen, SRPS_STAMP_PTR, gp);
                publish_row_edges();
                request_ring();
                SRPS_STAMP(5);
                double pw, rw, ww;
                if (SRPS_RES_DEBUG_ON(a)) { pw = 1e30; rw = 0.0; ww = 0.0; }
                else grid_sum3_collect<true>(a.ent3, gen, pw, rw, ww, SRPS_STAMP_PTR, gp);
                SRPS_STAMP(6);
                alpha = r1 / (float)pw;                    // dc.cu:269
                asm volatile("" : "+v"(alpha));
                red = 0.f;
#pragma unroll
                for (int c = 0; c < CPT; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[c].e[e] = fmaf(alpha, p[c].e[e], x[c].e[e]);      // dc.cu:270
                        r[c].e[e] = fmaf(-alpha, w[c].e[e], r[c].e[e]);     // dc.cu:272
                        red = fmaf(r[c].e[e], r[c].e[e], red);      // (forming this only in the 1-in-16 steps that use it measured the same: 7.93 - 7.99 us)
                    }
                SRPS_STAMP(12);
                await_ring(wr);
                SRPS_STAMP(13);
#pragma unroll
                for (int q = 0; q < RPT; ++q) rh[q] = fmaf(-alpha, wr[q], rh[q]);
                r0 = r1;
                const double t1 = 2.0 * (double)alpha * rw, t2 = (double)alpha * (double)alpha * ww;
                const double pred = (double)r1 - t1 + t2;
                // The predicted value is taken
                //  * while its three terms cancel less than two digits (float partial sums: relative error below 1e-5; a
                //    regular step has pred / (r.r + |t1| + t2) ~ 0.3), and
                //  * while it cannot have drifted: its error accumulates from step to step (it is never re-measured), so the
                //    direct sum is taken again -- and becomes the new anchor -- once r.r has fallen to a quarter of the anchor
                //    and at every 16th step (relative drift below 1e-5; about one step in 16 at the metric's size).
                // Otherwise the direct sum, one more wait.  Every block holds the same numbers: the decision is uniform.
                // (A step counter kept in a register instead of k & 15 cost 1.4 us per step in code generation.)
                if ((SRPS_RES_DEBUG_ON(a)) || (pred > 1e-2 * ((double)r1 + fabs(t1) + t2) && pred > 0.25 * (double)r1_anchor && (k & 15) != 0))
                    r1 = uniform_f((SRPS_RES_DEBUG_ON(a)) ? 1.f : (float)pred);
                else { r1 = uniform_f(grid_sum(red, a.ent, ++gen, sm, gp)); r1_anchor = r1; }
            }
            // ---- the next step's p = beta p + r, own pixels and ring (dc.cu:256-264), here, where r.r has just become known: at the
            // top of the next pass it cost a phase of its own behind the loop's turn.  After pass 0: beta = 0, p = 0 p + r = r.
            {
                float beta_n = pass0 ? 0.f : r1 / r0;      // dc.cu:262
                asm volatile("" : "+v"(beta_n));
#pragma unroll
                for (int c = 0; c < CPT; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[c].e[e] = scal_then_axpy(beta_n, p[c].e[e], r[c].e[e]);
#pragma unroll
                for (int q = 0; q < RPT; ++q)
                    if (tid + q * NT < NRING) {
                        const int ri = ring_index(tid + q * NT);
                        hp[ri] = pass0 ? rh[q] : scal_then_axpy(beta_n, hp[ri], rh[q]);
                    }
            }
        } else {
            if (pass0) {
                // ---- r = b - A_ x ; r.r -----------------------------------------------------------------------------
                red = 0.f;
#pragma unroll
                for (int c = 0; c < CPT; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        r[c].e[e] -= w[c].e[e];
                        red = fmaf(r[c].e[e], r[c].e[e], red);
                    }
            } else {
                const float dot = (SRPS_RES_DEBUG_ON(a)) ? fmaxf(block_sum(red, sm), 1e30f) : grid_sum(red, a.ent, ++gen, sm, gp);
                alpha = r1 / dot;                              // dc.cu:269
                asm volatile("" : "+v"(alpha));
                // ---- x += alpha p ; r -= alpha omega ; r.r ------------------------------------------------------------
                red = 0.f;
#pragma unroll
                for (int c = 0; c < CPT; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[c].e[e] = fmaf(alpha, p[c].e[e], x[c].e[e]);      // dc.cu:270
                        r[c].e[e] = fmaf(-alpha, w[c].e[e], r[c].e[e]);     // dc.cu:272
                        red = fmaf(r[c].e[e], r[c].e[e], red);
                    }
            }
            publish_edges(r);
            if (!pass0) r0 = r1;
            // r.r, and r on the ring (the neighbours' edges of this generation): the ring granules are requested before
            // the wait for the partial sums, so that both arrive within one round trip
            if (SRPS_RES_DEBUG_ON(a)) r1 = uniform_f(fminf(fmaxf(block_sum(red, sm), 1.f), 2.f));
            else {
                ++gen;
                grid_sum_publish(red, a.ent, gen, sm, gp);
                publish_row_edges();
                request_ring();
                r1 = uniform_f(grid_sum_collect(a.ent, gen, sm, gp));
                float rv[RPT];
                await_ring(rv);
#pragma unroll
                for (int q = 0; q < RPT; ++q)
                    if (hoff[q] >= 0) rh[q] = rv[q];
            }
        }
        SRPS_STAMP(7);
    };
    // The exit test goes through a scalar register: every lane holds the same r1 and k, but the compiler cannot know that of a
    // value read from LDS, and counts the loop as divergent.
    // General body: pass 0 instantiated ahead of the loop, the CG step inside it.  With pass 0 as a branch inside the loop its two
    // routes deliver r and x in different registers -- 48 64-bit moves per step to bring them together, the loop carries the
    // pass-0 code along, and the body spills (64 bytes of scratch per lane, 12 this way): 10.5 -> 9.9 us per step on one box.
    // RECT body: the one loop with the run-time flag; it does not spill either way, and instantiated twice it measured 8.67
    // against 8.52 us per step (same box, tools/ab_variants.sh), although its loop is 12 % shorter in issue clocks.
    if constexpr (RECT) {
        struct RtBool { bool value; };
        bool pass0 = true;
        while (__builtin_amdgcn_readfirstlane((int)(pass0 || (r1 > a.tol2 && k < a.max_steps)))) { cg_pass(RtBool{pass0}); pass0 = false; }
    } else {
        cg_pass(std::true_type{});
        while (__builtin_amdgcn_readfirstlane((int)(r1 > a.tol2 && k < a.max_steps))) cg_pass(std::false_type{});
    }
    // ---- results ---------------------------------------------------------------------------------------------
    // a block one of whose waits gave up stores nothing: x keeps the iterate the launch started from
    const bool dead = spin_block_dead();
    if (act && !dead) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) st4(a.x_out + (size_t)(gcol0 + c + PAD) * Hs + srow0, x[c]);
    }
    if (blockIdx.x == 0 && tid == 0) {
        a.scal->r0 = r0; a.scal->r1_last = r1; a.scal->iters = k; a.scal->active = (r1 > a.tol2) ? 1 : 0;
        a.scal->alpha = 0.f;                               // nothing pending: x is final
    }
    (void)ntile;
}

// Two kernels, one per body.  A CG step lasts as long as its slowest tile (every step ends in a grid-wide wait), so the body
// without structure bits only pays when EVERY tile of the grid qualifies for it -- a full-frame mask, or an object that covers
// its whole bounding box -- and that is when the host launches the RECT kernel; any other grid runs the general kernel on all
// its tiles.  (Both bodies in one kernel, chosen per block, made the general body 0.6 us per step slower: the two share one
// register allocation.)
template <int SF, int NC, bool ONE_SYNC, bool RECT>
__global__ __launch_bounds__(NT) void k_cg_resident(ResidentArgs a) {
    // XCD-aware tile order: the blocks of one XCD (b, b+8, ...) get a contiguous range of tiles
    int tile = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = tile & 7, kk = tile >> 3;
#ifdef SRPS_RES_REVERSE_KK      // experiment: is a slow block slow because of its CU or because of its tile?
        tile = xcd * q + min(xcd, rem) + ((q + (xcd < rem ? 1 : 0)) - 1 - kk);
#else
        tile = xcd * q + min(xcd, rem) + kk;
#endif
    }
    tile = a.tile_list[tile];                              // one block per occupied tile
    const unsigned cls = RECT ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.tile_cls[tile]) : 0u;
    resident_body<SF, NC, ONE_SYNC, RECT>(a, tile, cls);
}

// The same body as one RANK of a group of launches that together cover the grid (srps_strip_group_solve_resident): the rank's blocks
// work on its own tiles (a.tile_list, a column range of tiles), publish their sums into the slot a single launch over the whole
// grid would give them -- in every rank's array -- and the tiles on the strip's border also leave their edge columns with the
// neighbouring rank.  Same arithmetic per block, same order of the grid-wide sums: the same bits as the single launch.
template <int SF, int NC, bool ONE_SYNC, bool RECT>
__global__ __launch_bounds__(NT) void k_cg_resident_group(ResidentArgs a) {
    __shared__ GridPeers gps;
    int li = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = li & 7, kk = li >> 3;
        li = xcd * q + min(xcd, rem) + kk;                 // XCD-aware order inside the rank's own launch
    }
    const int tile = a.tile_list[li];
    if (threadIdx.x == 0) {
        // the block index that the single launch's XCD-aware order (k_cg_resident) maps onto entry L of the group's tile list
        const int L = a.grp_list_base + li, NB = a.grp.nb, q = NB >> 3, rem = NB & 7;
        int xcd, kk;
        if (L < rem * (q + 1)) { xcd = L / (q + 1); kk = L - xcd * (q + 1); }
        else { const int L2 = L - rem * (q + 1); xcd = rem + L2 / q; kk = L2 - (L2 / q) * q; }
        gps = a.grp;
        gps.slot = kk * 8 + xcd;
    }
    __syncthreads();
    const unsigned cls = RECT ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.tile_cls[tile]) : 0u;
    resident_body<SF, NC, ONE_SYNC, RECT, true>(a, tile, cls, &gps);
}

size_t resident_lds_bytes(int NC) {
    const size_t gl = (NC == 3) ? 2 : 0;
    return (gl * CPT * NT + 2 * NT) * sizeof(float4) + (size_t)RING * sizeof(float) * (1 + NC + 1) + 40 * sizeof(float) + 16 + 2 * TR * sizeof(float);
}

}  // namespace

#if defined(SRPS_STAMPS) && SRPS_RES_NT == 512 && SRPS_RES_CPT == 8
extern "C" int srps_debug_read_wave_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), sizeof(g_wstamps));
}
extern "C" int srps_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#endif

bool SRPS_RES_NAME(resident_supported)(const srps_ctx* ctx) {
    const Grid& G = ctx->grid;
    if (!ctx->cg_resident || !use_march(ctx)) return false;
    const int nc = march_recompute_channels(ctx);
    if (nc != 1 && nc != 3) return false;
    if (G.sf != 1 && G.sf != 2 && G.sf != 4) return false;
    if (G.sf > CPT) return false;                          // a thread's columns hold whole sf x sf blocks of KT
    const int shape = TC == 64 ? 1 : (TC == 32 ? 0 : 2);      // index of the tiling in Grid::d_tile_cls
    return G.n_occ[shape] > 0 && G.n_occ[shape] <= ctx->num_cus;      // a CU for every tile that holds unknowns; empty tiles of the bounding box take none
}

// the residual b - A_ x0 (devicecalls.cu:758) and the whole CG of devicecalls.cu:252-275: G.d_r holds b, G.d_x holds x0
int SRPS_RES_NAME(resident_cg)(srps_ctx* ctx, int max_steps, bool fixed_steps) {
    Grid& G = ctx->grid;
    const int nc = march_recompute_channels(ctx);
    const int nbr = cdiv(G.Hg, TR), nbc = cdiv(G.Wg, TC), tiles = nbr * nbc;
    // ent [2][tiles] | ent3 [2][tiles rounded up to 256] 16-byte granules | edge granules [tiles][2][HALO_N]
    const size_t ent_n = ((size_t)tiles * 2 + 1) & ~(size_t)1, ent3_n = (size_t)((tiles + 255) & ~255) * 2 * SRPS_G3_REPLICAS * (SRPS_G3_STRIDE / 8);
    const size_t need = (ent_n + ent3_n + (size_t)tiles * 2 * HALO_N) * sizeof(unsigned long long);
    SRPS_TRY(ensure(ctx->ws_resident, need));
    // The granule arrays are zeroed when they are new, when another driver has used them (the strip groups number from zero) and when the
    // launch numbers run out; otherwise this launch's tags lie above every tag in them (gen_base) and nothing is zeroed: a tag is
    // compared for EQUALITY with the generation a block waits for, and a launch uses fewer than 1024 generations (pass 0 + 101 steps +
    // the direct sums of the one-wait form; two waits per step: 2 x 102).
    // ... and when the LAYOUT of the arrays changes (another grid or tile shape in the same buffer: ent | ent3 | edge granules move with
    // `tiles`, TC and HALO_N, and a word that held a value -- float bits -- could then lie where this layout reads a tag; round-5 advisor finding)
    const unsigned long long layout = ((unsigned long long)tiles << 32) | ((unsigned long long)TC << 16) | (unsigned long long)NT;
    if (ctx->res_tags_ptr != ctx->ws_resident.p || ctx->res_tags_bytes < need || ctx->res_tags_layout != layout || ctx->res_launch_seq + (unsigned)((2 * ((long long)max_steps + 2) + 16 + 1023) >> 10) >= (1u << 21)) {
        SRPS_HIP(hipMemsetAsync(ctx->ws_resident.p, 0, ctx->ws_resident.bytes, ctx->stream));
        ctx->res_tags_ptr = ctx->ws_resident.p; ctx->res_tags_bytes = ctx->ws_resident.bytes; ctx->res_launch_seq = 0; ctx->res_tags_layout = layout;
    }
    const unsigned gen_base = ctx->res_launch_seq << 10;
    ctx->res_launch_seq += (unsigned)((2 * ((long long)max_steps + 2) + 16 + 1023) >> 10);      // the tags this launch may use, in units of 1024 (option "cg_max_iter" can ask for many steps)
    // the kernel addresses the streamed g plane through a buffer descriptor whose num_records is an int of BYTES
    SRPS_REQUIRE((unsigned long long)G.plane * sizeof(float) < (1ull << 31), SRPS_ERR_UNSUPPORTED,
                 "resident CG: a plane of %zu floats does not fit the kernel's buffer descriptor (2 GiB)", (size_t)G.plane);
    ResidentArgs a;
    memset(&a, 0, sizeof(a));
    a.G = G.d_G; a.flags = G.d_flags; a.consts = G.d_tconsts; a.x = G.d_x; a.x_out = G.d_x2; a.r = G.d_r;
    a.ent = (unsigned long long*)ctx->ws_resident.p;
    a.ent3 = a.ent + ent_n;
    a.halo = a.ent3 + ent3_n;
    a.scal = G.d_scal;
    a.Hs = G.Hs; a.Ws = G.Ws; a.plane = G.plane; a.nbr = nbr; a.nbc = nbc;
    a.lambda = ctx->lambda;
    a.inv_sf4 = 1.0f / ((float)(G.sf * G.sf) * (float)(G.sf * G.sf));
    a.tol2 = fixed_steps ? -1.f : ctx->cg_tol * ctx->cg_tol;
    a.max_steps = max_steps;
    a.cx = G.cx; a.cy = G.cy; a.i_lo = G.i_lo; a.j_lo = G.j_lo;
    a.debug = ctx->cg_resident_debug;
    a.gen_base = gen_base;
    a.spin_ticks = (unsigned long long)ctx->spin_budget_ms * 100000ull;      // s_memrealtime: 100 MHz
    const int shape = TC == 64 ? 1 : (TC == 32 ? 0 : 2);      // index of the tiling in Grid::d_tile_cls
    const int blocks = G.n_occ[shape];
    const bool rect = ctx->cg_resident_rect && G.n_tiles[shape] == tiles && G.n_rect_tiles[shape] == blocks;      // every occupied tile qualifies
    a.tile_cls = rect ? G.d_tile_cls[shape] : nullptr;
    a.tile_occ = G.d_tile_cls[shape];
    a.tile_list = G.d_tile_list[shape];
    const void* fn = nullptr;
#define SRPS_RES(SFV, NCV) fn = rect ? (ctx->cg_one_sync ? (const void*)k_cg_resident<SFV, NCV, true, true> : (const void*)k_cg_resident<SFV, NCV, false, true>) \
                                    : (ctx->cg_one_sync ? (const void*)k_cg_resident<SFV, NCV, true, false> : (const void*)k_cg_resident<SFV, NCV, false, false>)
    if (G.sf == 4) {
        if constexpr (CPT >= 4) { if (nc == 3) SRPS_RES(4, 3); else SRPS_RES(4, 1); }
        else return SRPS_ERR_UNSUPPORTED;
    } else if (nc == 3) { if (G.sf == 1) SRPS_RES(1, 3); else SRPS_RES(2, 3); }
    else { if (G.sf == 1) SRPS_RES(1, 1); else SRPS_RES(2, 1); }
#undef SRPS_RES
    const size_t lds = resident_lds_bytes(nc);
    SRPS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {&a};
    const int rc = launch_persistent(ctx, fn, blocks, NT, args, lds);      // SRPS_ERR_UNSUPPORTED: the caller falls back to the streaming kernels
    if (rc == SRPS_OK) { std::swap(G.d_x, G.d_x2); ctx->x_swapped = true; }      // the result is in the other plane (see persistent_aborts)
    return rc;
}

// ---- one rank of a GROUP of resident launches in this unit's tile shape (drivers: resident_cg_group / resident_cg_rank below) --------
size_t SRPS_RES_NAME(resident_group_bytes)(int tiles) {
    const size_t ent_n = ((size_t)tiles * 2 + 1) & ~(size_t)1, ent3_n = (size_t)((tiles + 255) & ~255) * 2 * SRPS_G3_REPLICAS * (SRPS_G3_STRIDE / 8);
    return (ent_n + ent3_n + (size_t)tiles * 2 * HALO_N) * sizeof(unsigned long long);
}
int SRPS_RES_NAME(resident_group_launch)(srps_ctx* ctx, const ResidentGroupSpec& sp) {
    Grid& G = ctx->grid;
    const int nc = march_recompute_channels(ctx);
    const int shape = TC == 64 ? 1 : (TC == 32 ? 0 : 2);
    const size_t ent_n = ((size_t)sp.tiles * 2 + 1) & ~(size_t)1, ent3_n = (size_t)((sp.tiles + 255) & ~255) * 2 * SRPS_G3_REPLICAS * (SRPS_G3_STRIDE / 8);
    auto carve = [&](void* base, unsigned long long*& ent, unsigned long long*& ent3, unsigned long long*& halo) {
        ent = (unsigned long long*)base; ent3 = ent + ent_n; halo = ent3 + ent3_n;
    };
    ResidentArgs a;
    memset(&a, 0, sizeof(a));
    a.G = G.d_G; a.flags = G.d_flags; a.consts = G.d_tconsts; a.x = G.d_x; a.x_out = G.d_x2; a.r = G.d_r;
    carve(sp.exch, a.ent, a.ent3, a.halo);
    a.scal = G.d_scal;
    a.Hs = G.Hs; a.Ws = G.Ws; a.plane = G.plane; a.nbr = sp.nbr; a.nbc = sp.nbc;
    a.lambda = ctx->lambda; a.inv_sf4 = 1.0f / ((float)(G.sf * G.sf) * (float)(G.sf * G.sf));
    a.tol2 = sp.fixed_steps ? -1.f : ctx->cg_tol * ctx->cg_tol; a.max_steps = sp.max_steps;
    a.cx = G.cx; a.cy = G.cy; a.i_lo = G.i_lo; a.j_lo = G.j_lo;
    a.spin_ticks = (unsigned long long)ctx->spin_budget_ms * 100000ull;
    a.tile_cls = sp.rect ? G.d_tile_cls[shape] : nullptr; a.tile_occ = G.d_tile_cls[shape];
    a.tile_list = G.d_tile_list[shape] + sp.list_base;
    a.grp.nb = sp.nb_total; a.grp.slot = 0; a.grp.n_peers = sp.n_peers; a.grp_list_base = sp.list_base;
    a.grp_bc_first = sp.bc_first; a.grp_bc_last = sp.bc_last;
    for (int q = 0; q < sp.n_peers; ++q) {
        unsigned long long *e, *e3, *hl;
        carve(sp.peer_exch[q], e, e3, hl);
        a.grp.ent[q] = e; a.grp.ent3[q] = e3;
        if (q == sp.left_peer) a.grp_halo_left = hl;
        if (q == sp.right_peer) a.grp_halo_right = hl;
    }
    if (sp.blocks == 0) {      // a strip without a masked pixel launches nothing: its report record says what the others' kernels say of a full solve
        SRPS_HIP(hipMemsetAsync(G.d_scal, 0, 5 * sizeof(int), ctx->stream));
        const int steps = sp.max_steps;
        SRPS_TRY(host_upload(ctx, &G.d_scal->iters, &steps, sizeof(int), ctx->stream));
        return SRPS_OK;
    }
    const void* fn = nullptr;
#define SRPS_RESG(SFV, NCV) fn = sp.rect ? (const void*)k_cg_resident_group<SFV, NCV, true, true> : (const void*)k_cg_resident_group<SFV, NCV, true, false>
    if (G.sf == 4) {
        if constexpr (CPT >= 4) { if (nc == 3) SRPS_RESG(4, 3); else SRPS_RESG(4, 1); }
        else return SRPS_ERR_UNSUPPORTED;
    } else if (nc == 3) { if (G.sf == 1) SRPS_RESG(1, 3); else SRPS_RESG(2, 3); }
    else { if (G.sf == 1) SRPS_RESG(1, 1); else SRPS_RESG(2, 1); }
#undef SRPS_RESG
    const size_t lds = resident_lds_bytes(nc);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return SRPS_ERR_UNSUPPORTED; }
    ctx->persistent_inflight = 1;
    void* kargs[] = {&a};
    // a PLAIN launch: the ranks' kernels must run side by side, and cooperative launches of one device queue one behind the other
    SRPS_HIP(hipLaunchKernel(fn, dim3(sp.blocks), dim3(NT), kargs, lds, ctx->stream));
    return SRPS_OK;
}

#if SRPS_RES_NT == 512 && SRPS_RES_CPT == 8
// ---- the resident CG on the column strips of a group of contexts ----------------------------------------
// n contexts that hold the SAME assembled depth system (replicated state, as after srps_depth_partial on every one) each run the
// resident kernel on a range of tile columns; together the launches cover the grid.  What crosses the ranks inside a CG step is what
// crosses the blocks of the single launch: the 16-byte granules of the three sums (every block into every rank's array) and the
// 8-byte edge granules of the tiles on a strip's border (into the neighbouring rank's array) -- plain stores through pointers into
// the other rank's memory: the same device (several contexts of one process: the test bed of a one-GPU box) or a peer device over
// xGMI (hipDeviceEnablePeerAccess; not run anywhere yet).  No host step between the 101 CG steps.  All launches must be resident
// TOGETHER: plain launches on the contexts' own streams, the bounded waits of device_utils.h end a group that cannot be
// (SRPS_ERR_UNSUPPORTED after the budget; nothing is stored).
namespace {
// the tile shapes a group can run in (one compile unit each): columns per tile, threads, columns per thread, index of the tiling
struct GroupUnit {
    int tc, nt, cpt, shape, option;      // option: the value of "cg_resident_tile" that forces the unit
    size_t (*bytes)(int);
    int (*launch)(srps_ctx*, const ResidentGroupSpec&);
};
const GroupUnit kGroupUnits[] = {
    {16, 512, 2, 2, 2, resident_group_bytes_n512c2, resident_group_launch_n512c2},
    {16, 256, 4, 2, 16, resident_group_bytes_n256c4, resident_group_launch_n256c4},
    {32, 512, 4, 0, 32, resident_group_bytes_n512c4, resident_group_launch_n512c4},
    {32, 256, 8, 0, 256, resident_group_bytes_n256, resident_group_launch_n256},
    {64, 512, 8, 1, 512, resident_group_bytes_n512, resident_group_launch_n512},
};
struct GroupPlan {
    const GroupUnit* unit = nullptr;
    int nbr = 0, nbc = 0, tiles = 0, NB = 0;
    bool rect = false;
    std::vector<int> tc0, lbase;         // per rank: first column of tiles, first entry of the tile list ([n] = end)
};
// ranges of tile columns for n ranks in unit u's tiling, and whether every rank's tiles get a CU each (all of them together when the
// ranks share a device)
bool group_partition(const srps_ctx* ctx, const GroupUnit& u, int n, bool one_device, GroupPlan& pl) {
    const Grid& G = ctx->grid;
    if (G.sf > u.cpt || (G.sf == 4 && u.cpt < 4)) return false;      // a thread's columns hold whole sf x sf blocks of KT
    pl.nbr = cdiv(G.Hg, 256); pl.nbc = cdiv(G.Wg, u.tc); pl.tiles = pl.nbr * pl.nbc; pl.NB = G.n_occ[u.shape];
    const std::vector<int>& list = G.h_tile_list[u.shape];
    if (pl.nbc < n || pl.NB <= 0 || G.n_tiles[u.shape] != pl.tiles || (int)list.size() != pl.NB) return false;
    pl.tc0.assign(n + 1, 0); pl.lbase.assign(n + 1, 0);
    for (int r = 0; r <= n; ++r) pl.tc0[r] = (int)((long long)pl.nbc * r / n);
    for (int r = 0, i = 0; r <= n; ++r) {
        while (i < pl.NB && list[(size_t)i] / pl.nbr < pl.tc0[r]) ++i;      // tiles are numbered column by column
        pl.lbase[r] = i;
    }
    for (int r = 0; r < n; ++r)
        if (pl.lbase[r + 1] - pl.lbase[r] > ctx->num_cus) return false;
    if (one_device && pl.NB > ctx->num_cus) return false;
    pl.rect = ctx->cg_resident_rect && G.n_rect_tiles[u.shape] == pl.NB;
    pl.unit = &u;
    return true;
}
// the unit of a group: the one "cg_resident_tile" forces, else the smallest tiles that still give every tile of every rank a CU --
// with the single launch's preferences (resident_shape below): 256 x 16 while a rank has few of them, then 256 x 32, then 256 x 64
bool group_plan(const srps_ctx* ctx, int n, bool one_device, GroupPlan& pl) {
    const int nc = march_recompute_channels(ctx);
    const Grid& G = ctx->grid;
    if (!((nc == 1 || nc == 3) && (G.sf == 1 || G.sf == 2 || G.sf == 4) && use_march(ctx))) return false;
    const int want = ctx->cg_resident_tile;
    for (const GroupUnit& u : kGroupUnits) {
        if (want != 0 && u.option != want) continue;
        GroupPlan cand;
        if (!group_partition(ctx, u, n, one_device, cand)) continue;
        if (want == 0) {
            int most = 0;
            for (int r = 0; r < n; ++r) most = std::max(most, cand.lbase[r + 1] - cand.lbase[r]);
            if (u.option == 2 && most > 240) continue;
            if (u.option == 16 && most > 96) continue;
            if (u.option == 256) continue;                 // 256 x 32 with 256 threads: only when asked for
        }
        pl = cand;
        return true;
    }
    return false;
}
}  // namespace

int resident_exchange_buffer(srps_ctx* ctx, size_t need, bool coarse_ok);      // below

int resident_cg_group(srps_ctx* const* ctxs, int n, int max_steps, bool fixed_steps) {
    SRPS_REQUIRE(n >= 1 && n <= 8, SRPS_ERR_INVALID, "resident strip group: 1 .. 8 ranks");
    srps_ctx* c0 = ctxs[0];
    const Grid& G0 = c0->grid;
    bool one_device = true;
    for (int r = 0; r < n; ++r) {
        srps_ctx* c = ctxs[r];
        SRPS_REQUIRE(c->grid.Hg == G0.Hg && c->grid.Wg == G0.Wg && c->grid.sf == G0.sf && c->grid.Hs == G0.Hs && c->grid.P == G0.P, SRPS_ERR_INVALID,
                     "resident strip group: rank %d holds another grid", r);
        SRPS_REQUIRE(march_recompute_channels(c) == march_recompute_channels(c0), SRPS_ERR_INVALID, "resident strip group: rank %d has another operator form", r);
        for (int q = 0; q < r; ++q)
            SRPS_REQUIRE(ctxs[q]->stream != c->stream, SRPS_ERR_INVALID, "resident strip group: ranks %d and %d share a stream (their launches must run side by side)", q, r);
        one_device = one_device && c->device == c0->device;
    }
    GroupPlan pl;
    SRPS_REQUIRE(cdiv(G0.Wg, 16) >= n, SRPS_ERR_INVALID, "resident strip group: %d ranks for %d columns of tiles", n, cdiv(G0.Wg, 16));
    if (one_device) {
        // streams of one process share the runtime's hardware queues (GPU_MAX_HW_QUEUES, 4 by default): a fifth launch would wait in a
        // queue behind one of the first four, which wait for it -- found by a test that dealt a grid to five contexts
        const char* e = getenv("GPU_MAX_HW_QUEUES");
        const int queues = e ? atoi(e) : 4;
        SRPS_REQUIRE(n <= std::max(queues, 1), SRPS_ERR_UNSUPPORTED,
                     "resident strip group: %d launches of one process on one device need %d hardware queues (GPU_MAX_HW_QUEUES is %d): they could not run side by side", n, n, queues);
    }
    SRPS_REQUIRE(group_plan(c0, n, one_device, pl), SRPS_ERR_UNSUPPORTED,
                 "resident strip group: no tile shape gives every tile of every rank a CU (%d ranks%s), or the operator is not the tensor-recompute form", n,
                 one_device ? " on one device" : "");
    const GroupUnit& U = *pl.unit;
    const size_t need = U.bytes(pl.tiles);
    std::vector<hipEvent_t> ready((size_t)n, nullptr);
    std::vector<void*> exch((size_t)n, nullptr);           // every rank's granule arrays
    auto cleanup = [&]() { for (auto e : ready) if (e) (void)hipEventDestroy(e); };
    int rc = SRPS_OK;
    for (int r = 0; r < n && rc == SRPS_OK; ++r) {
        srps_ctx* c = ctxs[r];
        if (hipSetDevice(c->device) != hipSuccess) { rc = SRPS_ERR_HIP; break; }
        if (!one_device) {
            // Peers on other DEVICES store into this rank's arrays and this rank polls them while the kernels run: that needs memory that is
            // coherent across agents at instruction granularity -- fine-grained -- and a device that can reach the peer's memory at all.
            // (hipMalloc memory is coherent across agents at kernel boundaries only: the owner's polls may be served from its own L2 for
            // ever.  Round-4 review: on one device both kinds work, which is why the one-GPU tests could not tell.)
            for (int q = 0; q < n && rc == SRPS_OK; ++q) {
                if (ctxs[q]->device == c->device) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, c->device, ctxs[q]->device) != hipSuccess || !can) {
                    (void)hipGetLastError();
                    set_error("resident strip group: device %d cannot access device %d's memory", c->device, ctxs[q]->device);
                    rc = SRPS_ERR_UNSUPPORTED; break;
                }
                const hipError_t e = hipDeviceEnablePeerAccess(ctxs[q]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error("resident strip group: hipDeviceEnablePeerAccess(%d): %s", ctxs[q]->device, hipGetErrorString(e)); rc = SRPS_ERR_UNSUPPORTED; }
                (void)hipGetLastError();
            }
            if (rc != SRPS_OK) break;
            if ((rc = resident_exchange_buffer(c, need, /*coarse_ok=*/false)) != SRPS_OK) break;
            exch[(size_t)r] = c->xg_buf;
        } else {
            if ((rc = ensure(c->ws_resident, need)) != SRPS_OK) break;
            exch[(size_t)r] = c->ws_resident.p;
            c->res_tags_ptr = nullptr;                       // the group numbers its generations from zero: the next single launch zeroes the arrays
        }
        if (hipMemsetAsync(exch[(size_t)r], 0, need, c->stream) != hipSuccess || hipEventCreateWithFlags(&ready[(size_t)r], hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(ready[(size_t)r], c->stream) != hipSuccess) { rc = SRPS_ERR_HIP; break; }
    }
    // launches: every rank waits until ALL granule arrays are zeroed, then runs on its own stream
    for (int r = 0; r < n && rc == SRPS_OK; ++r) {
        srps_ctx* c = ctxs[r];
        if (hipSetDevice(c->device) != hipSuccess) { rc = SRPS_ERR_HIP; break; }
        for (int q = 0; q < n; ++q)
            if (q != r && hipStreamWaitEvent(c->stream, ready[(size_t)q], 0) != hipSuccess) rc = SRPS_ERR_HIP;
        if (rc != SRPS_OK) break;
        ResidentGroupSpec sp;
        memset(&sp, 0, sizeof(sp));
        sp.exch = exch[(size_t)r]; sp.left_peer = -1; sp.right_peer = -1;
        for (int q = 0; q < n; ++q) {
            if (q == r) continue;
            if (q == r - 1) sp.left_peer = sp.n_peers;
            if (q == r + 1) sp.right_peer = sp.n_peers;
            sp.peer_exch[sp.n_peers++] = exch[(size_t)q];
        }
        sp.tiles = pl.tiles; sp.nbr = pl.nbr; sp.nbc = pl.nbc; sp.nb_total = pl.NB; sp.list_base = pl.lbase[r]; sp.blocks = pl.lbase[r + 1] - pl.lbase[r];
        sp.bc_first = pl.tc0[r]; sp.bc_last = pl.tc0[r + 1] - 1; sp.rect = pl.rect; sp.max_steps = max_steps; sp.fixed_steps = fixed_steps;
        rc = U.launch(c, sp);
    }
    // wait for all of them; did every wait get served?
    bool aborted = false;
    for (int r = 0; r < n; ++r) {
        srps_ctx* c = ctxs[r];
        (void)hipSetDevice(c->device);
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = rc == SRPS_OK ? SRPS_ERR_HIP : rc;
        CgScalars hs;
        if (hipMemcpy(&hs, c->grid.d_scal, sizeof(hs), hipMemcpyDeviceToHost) == hipSuccess && hs.abort_flags) {
            aborted = true;
            (void)hipMemset(&c->grid.d_scal->abort_flags, 0, 3 * sizeof(int));
        }
        c->persistent_inflight = 0;
    }
    cleanup();
    if (rc == SRPS_OK && aborted) {
        set_error("resident strip group: the %d launches did not become resident together within %d ms (shared device?); nothing was stored", n, c0->spin_budget_ms);
        return SRPS_ERR_UNSUPPORTED;
    }
    // the result: every rank's strip is in its second plane; the planes swap roles and the strips travel to the other ranks
    for (int r = 0; r < n && rc == SRPS_OK; ++r) std::swap(ctxs[r]->grid.d_x, ctxs[r]->grid.d_x2);
    for (int r = 0; r < n && rc == SRPS_OK; ++r) {
        srps_ctx* c = ctxs[r];
        if (hipSetDevice(c->device) != hipSuccess) { rc = SRPS_ERR_HIP; break; }
        for (int q = 0; q < n; ++q) {
            if (q == r) continue;
            const int cb = pl.tc0[q] * U.tc, ce = std::min(pl.tc0[q + 1] * U.tc, G0.Wg);
            if (ce <= cb) continue;
            const size_t off = (size_t)(cb + PAD) * G0.Hs, cnt = (size_t)(ce - cb) * G0.Hs;
            if (hipMemcpyAsync(c->grid.d_x + off, ctxs[q]->grid.d_x + off, cnt * sizeof(float), hipMemcpyDefault, c->stream) != hipSuccess) { rc = SRPS_ERR_HIP; break; }
        }
    }
    for (int r = 0; r < n; ++r) { (void)hipSetDevice(ctxs[r]->device); (void)hipStreamSynchronize(ctxs[r]->stream); }
    if (rc != SRPS_OK && rc != SRPS_ERR_UNSUPPORTED) set_error("resident strip group: a HIP call failed (%s)", hipGetErrorString(hipGetLastError()));
    return rc;
}

// ---- the same, as ONE RANK of a communicator (cg_partition = 2): the other ranks are other processes, on other devices or on this one ----
// Every rank keeps its granule arrays (ent | ent3 | halo, for the tiles of the WHOLE grid) in one fine-grained buffer, exports it
// with hipIpcGetMemHandle and opens the others' (the 64-byte handles travel through the context's all-reduce, one float per byte:
// any collective the context has -- RCCL or the caller's -- will do).  A solve: zero the own buffer, all-reduce one float (a barrier:
// nobody publishes into a buffer that is not zeroed yet), launch the group kernel on the own strip of tile columns, then the strips
// of x travel (one broadcast per rank).  The kernels of the ranks run side by side for the whole solve and talk through the mapped
// buffers; nothing else happens between the 101 steps.
void resident_rank_release(srps_ctx* ctx) {
    for (int q = 0; q < 8; ++q) {
        if (ctx->xg_peer[q] && ctx->xg_peer_ipc[q]) (void)hipIpcCloseMemHandle(ctx->xg_peer[q]);      // (a same-process peer's pointer is that rank's own buffer: not ours to close)
        ctx->xg_peer[q] = nullptr; ctx->xg_peer_ipc[q] = 0;
    }
    ctx->xg_world = 0;
    if (ctx->xg_buf) (void)hipFree(ctx->xg_buf);
    ctx->xg_buf = nullptr; ctx->xg_bytes = 0; ctx->xg_fine = 0;
    (void)hipGetLastError();
}
// A rank's exchange buffer: FINE-grained device memory (coherent across agents while kernels run: another device stores into it and this
// device's waves poll it) -- coarse-grained memory (hipMalloc) is coherent across agents at kernel boundaries only, and is accepted only
// when `coarse_ok`: every rank of the group sits on this very device (the one-GPU test beds), where the L2 in front of the memory is one.
int resident_exchange_buffer(srps_ctx* ctx, size_t need, bool coarse_ok) {
    if (ctx->xg_buf && ctx->xg_bytes >= need && (ctx->xg_fine || coarse_ok)) return SRPS_OK;
    if (ctx->xg_buf) { (void)hipFree(ctx->xg_buf); ctx->xg_buf = nullptr; ctx->xg_bytes = 0; ctx->xg_fine = 0; }
    if (hipExtMallocWithFlags(&ctx->xg_buf, need, hipDeviceMallocFinegrained) == hipSuccess) { ctx->xg_bytes = need; ctx->xg_fine = 1; return SRPS_OK; }
    (void)hipGetLastError();
    ctx->xg_buf = nullptr;
    if (!coarse_ok) { set_error("resident strips: no fine-grained device memory for the exchange buffer (%zu bytes), and the ranks are not all on one device", need); return SRPS_ERR_UNSUPPORTED; }
    if (hipMalloc(&ctx->xg_buf, need) != hipSuccess) { (void)hipGetLastError(); ctx->xg_buf = nullptr; set_error("resident strips: no memory for the exchange buffer"); return SRPS_ERR_UNSUPPORTED; }
    ctx->xg_bytes = need;
    return SRPS_OK;
}

// What a rank tells the others in the handshake, one float per byte (any float all-reduce carries it: RCCL or the caller's collectives):
//   [0, 64)   the hipIpc handle of its exchange buffer (zeros: none)
//   [64, 72)  the buffer's address -- what a rank of the SAME PROCESS uses instead of the handle: HIP does not open a handle in the process
//             that exported it (srps_comm_init_all and the C++ host keep all ranks in one process, a thread per device)
//   [72, 76)  the process id
//   [76, 80)  the device: PCI domain (2 bytes), bus, device
//   [80]      the device's ordinal in that process (for hipDeviceEnablePeerAccess between ranks of one process)
//   [81]      1: the buffer is fine-grained memory
//   [82]      1: this rank got as far as having a buffer at all
//   [83, 91)  a random 64-bit number drawn once per PROCESS, [91, 95) a hash of the host's boot id and name: a pid alone says "same
//             process" only inside one PID namespace -- two ranks in separate containers of one node, or on two hosts under a caller's
//             transport, can carry the same pid, and a foreign virtual address taken for a local one is a memory fault or a spin to the
//             deadline (round-5 advisor finding).  Same process = same pid AND same number AND same host.
constexpr int XG_REC = 96;
struct XgRecord {
    unsigned char handle[64];
    unsigned long long addr;
    unsigned pid;
    unsigned char pci[4];
    int ordinal, fine, ok;
    unsigned long long nonce;
    unsigned host;
};
static unsigned long long process_nonce() {
    static const unsigned long long n = [] {
        unsigned long long v = 0;
        if (FILE* f = fopen("/dev/urandom", "rb")) { if (fread(&v, sizeof(v), 1, f) != 1) v = 0; fclose(f); }
        if (!v) { std::random_device rd; v = ((unsigned long long)rd() << 32) ^ rd() ^ ((unsigned long long)getpid() << 17) ^ (unsigned long long)time(nullptr); }
        return v ? v : 1ull;
    }();
    return n;
}
static unsigned host_hash() {
    static const unsigned h = [] {
        char buf[320]; size_t n = 0;
        memset(buf, 0, sizeof(buf));
        if (FILE* f = fopen("/proc/sys/kernel/random/boot_id", "rb")) { n = fread(buf, 1, 64, f); fclose(f); }
        if (gethostname(buf + n, sizeof(buf) - n - 1) != 0) buf[n] = 0;
        unsigned v = 2166136261u;                          // FNV-1a
        for (size_t i = 0; i < sizeof(buf) && (i < n || buf[i]); ++i) { v ^= (unsigned char)buf[i]; v *= 16777619u; }
        return v;
    }();
    return h;
}
static void xg_pack(const XgRecord& r, float* f) {
    for (int b = 0; b < 64; ++b) f[b] = (float)r.handle[b];
    for (int b = 0; b < 8; ++b) f[64 + b] = (float)((r.addr >> (8 * b)) & 0xffull);
    for (int b = 0; b < 4; ++b) f[72 + b] = (float)((r.pid >> (8 * b)) & 0xffu);
    for (int b = 0; b < 4; ++b) f[76 + b] = (float)r.pci[b];
    f[80] = (float)r.ordinal; f[81] = (float)r.fine; f[82] = (float)r.ok;
    for (int b = 0; b < 8; ++b) f[83 + b] = (float)((r.nonce >> (8 * b)) & 0xffull);
    for (int b = 0; b < 4; ++b) f[91 + b] = (float)((r.host >> (8 * b)) & 0xffu);
}
static void xg_unpack(const float* f, XgRecord& r) {
    memset(&r, 0, sizeof(r));
    for (int b = 0; b < 64; ++b) r.handle[b] = (unsigned char)f[b];
    for (int b = 0; b < 8; ++b) r.addr |= (unsigned long long)(unsigned char)f[64 + b] << (8 * b);
    for (int b = 0; b < 4; ++b) r.pid |= (unsigned)(unsigned char)f[72 + b] << (8 * b);
    for (int b = 0; b < 4; ++b) r.pci[b] = (unsigned char)f[76 + b];
    r.ordinal = (int)f[80]; r.fine = (int)f[81]; r.ok = (int)f[82];
    for (int b = 0; b < 8; ++b) r.nonce |= (unsigned long long)(unsigned char)f[83 + b] << (8 * b);
    for (int b = 0; b < 4; ++b) r.host |= (unsigned)(unsigned char)f[91 + b] << (8 * b);
}
// Collective: every rank of the communicator calls it in the same solve (the grid, and with it `need`, is the same on all).  Whatever
// goes wrong locally is RECORDED and the rank still takes part in the one exchange (with a record that says so): no return path lies
// in front of the collective (round-4 advisor finding), and every rank reads the same records, so all ranks reach the same verdict.
// `debug_ipc_same_process` (tests): a same-process peer is mapped through its handle as if it were another process's -- HIP refuses
// that, which is what the direct-pointer route exists for; the failure must be recognised, not hang or crash.
static int resident_rank_open(srps_ctx* ctx, size_t need) {
    const int world = ctx->comm_world, rank = ctx->comm_rank;
    if (ctx->xg_world == world && ctx->xg_bytes >= need) return SRPS_OK;
    bool ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
    for (int q = 0; q < 8; ++q) {
        if (ctx->xg_peer[q] && ctx->xg_peer_ipc[q]) (void)hipIpcCloseMemHandle(ctx->xg_peer[q]);
        ctx->xg_peer[q] = nullptr; ctx->xg_peer_ipc[q] = 0;
    }
    ctx->xg_world = 0;
    // fine-grained if it can be had; whether coarse memory will do is known only after the exchange (are all ranks on this device?)
    ok = ok && resident_exchange_buffer(ctx, need, /*coarse_ok=*/true) == SRPS_OK;
    XgRecord mine;
    memset(&mine, 0, sizeof(mine));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t");
    if (ok) {
        hipIpcMemHandle_t hm;
        if (hipIpcGetMemHandle(&hm, ctx->xg_buf) == hipSuccess) memcpy(mine.handle, &hm, 64);
        else (void)hipGetLastError();                      // ranks of this process do not need it; another process's will say so
        mine.addr = (unsigned long long)(uintptr_t)ctx->xg_buf;
        mine.fine = ctx->xg_fine;
    }
    mine.pid = (unsigned)getpid();
    mine.nonce = ctx->debug_foreign_pid_twin ? process_nonce() ^ (0x9e3779b97f4a7c15ull * (unsigned)(rank + 1)) : process_nonce();
    mine.host = host_hash();
    {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, ctx->device) == hipSuccess) {
            mine.pci[0] = (unsigned char)(pr.pciDomainID & 0xff); mine.pci[1] = (unsigned char)((pr.pciDomainID >> 8) & 0xff);
            mine.pci[2] = (unsigned char)(pr.pciBusID & 0xff); mine.pci[3] = (unsigned char)(pr.pciDeviceID & 0xff);
        } else { (void)hipGetLastError(); ok = false; }
    }
    mine.ordinal = ctx->device;
    mine.ok = ok ? 1 : 0;
    const size_t nf = (size_t)world * XG_REC;
    std::vector<float> h(nf, 0.f);
    xg_pack(mine, h.data() + (size_t)rank * XG_REC);
    // the exchange itself: its failure is every rank's failure (a collective that one rank cannot enter cannot be entered by the others either)
    int xrc = ensure(ctx->ws_misc, nf * sizeof(float));
    float* d = (float*)ctx->ws_misc.p;
    if (xrc == SRPS_OK) xrc = host_upload(ctx, d, h.data(), nf * sizeof(float), ctx->stream);
    if (xrc != SRPS_OK) return xrc;                        // no device buffer to enter the collective with: fatal for the job, not a fall-back
    SRPS_TRY(comm_all_reduce_sum(ctx, d, nf));             // every rank's record, one float per byte
    SRPS_TRY(host_download(ctx, h.data(), d, nf * sizeof(float), ctx->stream));
    std::vector<XgRecord> rec((size_t)world);
    bool all_ok = true, one_device = true, all_fine = true;
    for (int q = 0; q < world; ++q) {
        xg_unpack(h.data() + (size_t)q * XG_REC, rec[(size_t)q]);
        all_ok = all_ok && rec[(size_t)q].ok == 1;
        all_fine = all_fine && rec[(size_t)q].fine == 1;
        one_device = one_device && memcmp(rec[(size_t)q].pci, rec[0].pci, 4) == 0;
    }
    if (!all_ok) { set_error("resident strips: a rank could not allocate or export its exchange buffer"); return SRPS_ERR_UNSUPPORTED; }
    if (!all_fine && !one_device) {                        // the same verdict on every rank: they all read the same records
        set_error("resident strips: the ranks sit on different devices and not every exchange buffer is fine-grained memory");
        return SRPS_ERR_UNSUPPORTED;
    }
    bool mapped = true;
    for (int q = 0; q < world && mapped; ++q) {
        if (q == rank) { ctx->xg_peer[q] = ctx->xg_buf; continue; }
        const XgRecord& r = rec[(size_t)q];
        const bool same_process = r.pid == mine.pid && r.nonce == mine.nonce && r.host == mine.host && !ctx->debug_ipc_same_process;
        if (r.host != mine.host) {                          // neither a pointer nor a hipIpc handle crosses hosts
            set_error("resident strips: rank %d runs on another host (the exchange buffers are reached through device memory mappings)", q);
            mapped = false; break;
        }
        if (same_process) {
            // a rank of this process (another host thread, its own context): its pointer is valid here as it stands; a peer DEVICE must be
            // made accessible from this one
            if (r.ordinal != ctx->device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, ctx->device, r.ordinal) != hipSuccess || !can) {
                    (void)hipGetLastError();
                    set_error("resident strips: device %d cannot access device %d's memory", ctx->device, r.ordinal);
                    mapped = false; break;
                }
                const hipError_t e = hipDeviceEnablePeerAccess(r.ordinal, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error("resident strips: hipDeviceEnablePeerAccess(%d): %s", r.ordinal, hipGetErrorString(e)); mapped = false; }
                (void)hipGetLastError();
            }
            // the address must be what the record says it is -- device memory of that ordinal in THIS process -- before a kernel
            // stores through it
            hipPointerAttribute_t at;
            memset(&at, 0, sizeof(at));
            if (hipPointerGetAttributes(&at, (void*)(uintptr_t)r.addr) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != r.ordinal) {
                (void)hipGetLastError();
                set_error("resident strips: rank %d's buffer address is not device memory of device %d in this process", q, r.ordinal);
                mapped = false; break;
            }
            ctx->xg_peer[q] = (void*)(uintptr_t)r.addr;
            continue;
        }
        hipIpcMemHandle_t hq;
        memcpy(&hq, r.handle, 64);
        bool any = false;
        for (int b = 0; b < 64; ++b) any = any || r.handle[b] != 0;
        void* pq = nullptr;
        const hipError_t e = any ? hipIpcOpenMemHandle(&pq, hq, hipIpcMemLazyEnablePeerAccess) : hipErrorInvalidValue;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("resident strips: rank %d's exchange buffer could not be mapped (%s)", q, any ? hipGetErrorString(e) : "that rank exported no handle");
            mapped = false; break;
        }
        ctx->xg_peer[q] = pq; ctx->xg_peer_ipc[q] = 1;
    }
    if (!mapped) return SRPS_ERR_UNSUPPORTED;              // (the caller's one-float all-reduce tells the other ranks)
    ctx->xg_world = world;
    return SRPS_OK;
}
int resident_cg_rank(srps_ctx* ctx, int max_steps, bool fixed_steps) {
    Grid& G = ctx->grid;
    const int n = ctx->comm_world, rank = ctx->comm_rank;
    if (ctx->xg_failed || n < 2 || n > 8 || !comm_bound(ctx) || !ctx->cg_resident || !ctx->cg_one_sync) return SRPS_ERR_UNSUPPORTED;
    // (every decision up to the handshake depends on replicated state only -- the grid, the options --: all ranks take the same way)
    GroupPlan pl;
    if (!group_plan(ctx, n, /*one_device=*/false, pl)) return SRPS_ERR_UNSUPPORTED;
    const GroupUnit& U = *pl.unit;
    const size_t need = U.bytes(pl.tiles);
    const int orc = resident_rank_open(ctx, need);
    if (orc != SRPS_OK && orc != SRPS_ERR_UNSUPPORTED) return orc;      // the exchange itself failed: no rank can go on
    bool opened = orc == SRPS_OK && !forced_failure("resident_strips");      // (forced: as a rank whose mapping failed -- the flag below tells the others)
    if (orc == SRPS_OK && !opened) set_error("resident strips: refused on purpose (SRPS_FORCE_FAIL=resident_strips)");
    if (opened && hipMemsetAsync(ctx->xg_buf, 0, need, ctx->stream) != hipSuccess) { (void)hipGetLastError(); opened = false; }
    // One float through the all-reduce: the barrier (every rank's buffer is zeroed before any rank's kernel publishes into it) AND the
    // decision -- a rank whose mapping failed says so, and ALL ranks leave this path together (a rank that went its own way would meet
    // the others in different collectives).  Local failures up to here are in `opened`: nothing returns in front of the collective.
    const float mine_failed = opened ? 0.f : 1.f;
    float failed = 0.f;
    SRPS_TRY(ensure(ctx->ws_misc, 64));                    // (holds the handshake's records already: cannot fail here)
    SRPS_TRY(host_upload(ctx, ctx->ws_misc.p, &mine_failed, sizeof(float), ctx->stream));
    SRPS_TRY(comm_all_reduce_sum(ctx, (float*)ctx->ws_misc.p, 1));
    SRPS_TRY(host_download(ctx, &failed, ctx->ws_misc.p, sizeof(float), ctx->stream));
    if (failed != 0.f) { ctx->xg_failed = 1; return SRPS_ERR_UNSUPPORTED; }
    ResidentGroupSpec sp;
    memset(&sp, 0, sizeof(sp));
    sp.exch = ctx->xg_buf; sp.left_peer = -1; sp.right_peer = -1;
    for (int q = 0; q < n; ++q) {
        if (q == rank) continue;
        if (q == rank - 1) sp.left_peer = sp.n_peers;
        if (q == rank + 1) sp.right_peer = sp.n_peers;
        sp.peer_exch[sp.n_peers++] = ctx->xg_peer[q];
    }
    sp.tiles = pl.tiles; sp.nbr = pl.nbr; sp.nbc = pl.nbc; sp.nb_total = pl.NB; sp.list_base = pl.lbase[rank]; sp.blocks = pl.lbase[rank + 1] - pl.lbase[rank];
    sp.bc_first = pl.tc0[rank]; sp.bc_last = pl.tc0[rank + 1] - 1; sp.rect = pl.rect; sp.max_steps = max_steps; sp.fixed_steps = fixed_steps;
    SRPS_TRY(U.launch(ctx, sp));
    std::swap(G.d_x, G.d_x2); ctx->x_swapped = true;      // the rank's strip of the result is in the other plane (see persistent_aborts)
    // the other strips: one broadcast per rank, in place in the result plane
    for (int q = 0; q < n; ++q) {
        const int cb = pl.tc0[q] * U.tc, ce = std::min(pl.tc0[q + 1] * U.tc, G.Wg);
        if (ce <= cb) continue;
        SRPS_TRY(comm_broadcast(ctx, G.d_x + (size_t)(cb + PAD) * G.Hs, (size_t)(ce - cb) * G.Hs, q));
    }
    return SRPS_OK;
}

// the smallest tile that still gives every tile a CU (more CUs at work, less arithmetic per CU and step);
// cg_resident_tile = 16 | 256 | 512 forces the 256 x 16, 256 x 32 or 256 x 64 shape
static int resident_shape(const srps_ctx* ctx) {      // 2: 256 x 16, 4: 256 x 16 with 512 threads, 3: 256 x 32 with 512 threads, 0: 256 x 32 with 256, 1: 256 x 64, -1: none fits
    const int want = ctx->cg_resident_tile;
    if (want == 2) return resident_supported_n512c2(ctx) ? 4 : -1;
    if (want == 16) return resident_supported_n256c4(ctx) ? 2 : -1;
    if (want == 32) return resident_supported_n512c4(ctx) ? 3 : -1;
    if (want == 256) return resident_supported_n256(ctx) ? 0 : -1;
    if (want == 512) return resident_supported_n512(ctx) ? 1 : -1;
    // 256 x 16 tiles.  With 512 threads and two columns per thread (sf 1, 2) they win up to 240 tiles (the whole Mitten frame,
    // 69 tiles: 5.6 us per step against 7.1 with 256 x 32 tiles; 512 x 512: 4.9 / 5.8; 768 x 1280 ragged, 240 tiles: 6.5 / 7.3;
    // 1024 x 1024 full, 256 tiles: 6.5 against 6.2 -- there the 256 x 32 tiles stay).  With 256 threads and four columns per
    // thread (sf 4) while they are few (640 x 480, 90 tiles: 6.5 / 7.0; 1024 x 512, 128 tiles: 6.1 against 5.9): a step's cost
    // outside the columns (ring, exchange, skew of more blocks) does not shrink with the tile.
    const long tiles16 = ctx->grid.n_occ[2];               // occupied 256 x 16 tiles
    if (tiles16 <= 240 && resident_supported_n512c2(ctx)) return 4;
    if (tiles16 <= 96 && resident_supported_n256c4(ctx)) return 2;
    // 256 x 32 tiles: eight waves of four columns each (two waves per SIMD: an instruction issues in 2.3 clocks) rather than four
    // waves of eight columns (one wave per SIMD: 4.5)
    if (resident_supported_n512c4(ctx)) return 3;
    if (resident_supported_n256(ctx)) return 0;
    return resident_supported_n512(ctx) ? 1 : -1;
}
bool resident_supported(const srps_ctx* ctx) { return resident_shape(ctx) >= 0; }
// would the next resident launch on the bound grid be the kernel without structure bits (every tile qualifies)?
bool resident_rect_active(const srps_ctx* ctx) {
    int shape = resident_shape(ctx);
    if (shape < 0 || !ctx->cg_resident_rect) return false;
    if (shape == 3) shape = 0;      // the same 256 x 32 tiling
    if (shape == 4) shape = 2;      // the same 256 x 16 tiling
    return ctx->grid.n_occ[shape] > 0 && ctx->grid.n_rect_tiles[shape] == ctx->grid.n_occ[shape];
}
int resident_cg(srps_ctx* ctx, int max_steps, bool fixed_steps) {
    switch (resident_shape(ctx)) {
        case 2: return resident_cg_n256c4(ctx, max_steps, fixed_steps);
        case 3: return resident_cg_n512c4(ctx, max_steps, fixed_steps);
        case 4: return resident_cg_n512c2(ctx, max_steps, fixed_steps);
        case 0: return resident_cg_n256(ctx, max_steps, fixed_steps);
        default: return resident_cg_n512(ctx, max_steps, fixed_steps);
    }
}
#endif

}  // namespace srps
