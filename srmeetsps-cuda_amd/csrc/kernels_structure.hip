// kernels_structure.hip -- the grid structure built ON THE DEVICE from the mask: what SRPS.cu:151-203 expresses as index lists and
// the COO matrices KT, Dx, Dy (a host double loop over h*w with push_back, SRPS.cu:10-71, 151-193) becomes a bounding box, the
// compact <-> grid index maps and one structure byte per pixel.  Round 2 built this with a single-threaded host loop (17 ms at
// 2048 x 2048, inside every solve); here it is a handful of kernels (one block per image column; counts, prefix sums, fill)
// that run while the images cross PCIe, and one 64-byte read-back for the sizes the host must allocate.
//
// Order of the compact indices (the reference's, SRPS.cu:157-162 / 176-183): ascending linear index i + j*h, i.e. column after
// column, rows ascending -- so pixel (i, j) has index col_start[j] + (masked pixels above it in column j), and the complete
// sf x sf blocks are numbered the same way on the low-resolution grid.
#include <climits>
#include "srps_internal.h"
#include "device_utils.h"

namespace srps {

namespace {

// exclusive prefix of one flag per thread over a 256-thread block; returns this thread's rank and the block's total
__device__ __forceinline__ int block_rank256(bool flag, int* sm /* [4] */, int& total) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long b = __ballot(flag);
    const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();                                   // sm may still be read from the previous call
    if (lane == 0) sm[wave] = __popcll(b);
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) before += (k < wave) ? sm[k] : 0;
    total = sm[0] + sm[1] + sm[2] + sm[3];
    return before + in_wave;
}

__device__ __forceinline__ int block_reduce_sum(int v, int* sm /* [4] */) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ int block_reduce_min(int v, int* sm) {
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_down(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return min(min(sm[0], sm[1]), min(sm[2], sm[3]));
}
__device__ __forceinline__ int block_reduce_max(int v, int* sm) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return max(max(sm[0], sm[1]), max(sm[2], sm[3]));
}

// A. one block per image column: masked pixels of the column, its first and last masked row; the first sample that is neither 0
// nor 1 (the reference indexes with mask != 0, SRPS.cu:158, but compacts with mask == 1, devicecalls.cuh:19-24: anything else
// silently corrupts it -- refused here).  Also the mask as bytes (later kernels read a quarter of the bytes).
__global__ __launch_bounds__(256) void k_struct_columns(const float* __restrict__ mask, int h, uint8_t* __restrict__ mb, int* __restrict__ col_count,
                                                        int* __restrict__ col_lo, int* __restrict__ col_hi, int* __restrict__ bad) {
    __shared__ int sm[4];
    const int j = blockIdx.x;
    const float* mc = mask + (size_t)j * h;
    uint8_t* bc = mb + (size_t)j * h;
    int cnt = 0, lo = INT_MAX, hi = -1, first_bad = INT_MAX;
    for (int i = threadIdx.x; i < h; i += 256) {
        const float m = mc[i];
        if (!(m == 0.f || m == 1.f)) first_bad = min(first_bad, i);
        const bool on = m != 0.f;
        bc[i] = on ? 1 : 0;
        if (on) { ++cnt; lo = min(lo, i); hi = i; }
    }
    cnt = block_reduce_sum(cnt, sm);
    lo = block_reduce_min(lo, sm);
    hi = block_reduce_max(hi, sm);
    first_bad = block_reduce_min(first_bad, sm);
    if (threadIdx.x == 0) {
        col_count[j] = cnt; col_lo[j] = lo; col_hi[j] = hi;
        if (first_bad != INT_MAX) atomicMin(bad, (int)((size_t)j * h + first_bad) & INT_MAX);
    }
}

// A2. one block per low-resolution column of the IMAGE (blocks are aligned to the image's sf grid, so a block is complete or not
// whatever the bounding box turns out to be): which sf x sf blocks are fully masked (rows of KT: D*mask == 1 exactly,
// SRPS.cu:110-111, 163-183), and how many per column
__global__ __launch_bounds__(256) void k_struct_lr_full(const uint8_t* __restrict__ mb, int h, int sf, uint8_t* __restrict__ lr_full, int* __restrict__ lr_count) {
    __shared__ int sm[4];
    const int bj = blockIdx.x, hs = h / sf;
    int cnt = 0;
    for (int bi = threadIdx.x; bi < hs; bi += 256) {
        bool full = true;
        for (int dj = 0; dj < sf; ++dj) {
            const uint8_t* c = mb + (size_t)(bj * sf + dj) * h + (size_t)bi * sf;
            for (int di = 0; di < sf; ++di) full &= c[di] != 0;
        }
        lr_full[(size_t)bj * hs + bi] = full ? 1 : 0;
        cnt += full ? 1 : 0;
    }
    cnt = block_reduce_sum(cnt, sm);
    if (threadIdx.x == 0) lr_count[bj] = cnt;
}

// B. one block: exclusive prefix sums of the per-column counts (HR and LR), the bounding box, the totals
__global__ __launch_bounds__(256) void k_struct_prefix(const int* __restrict__ col_count, const int* __restrict__ col_lo, const int* __restrict__ col_hi, int w,
                                                       int* __restrict__ col_start /* [w + 1] */, const int* __restrict__ lr_count, int ws,
                                                       int* __restrict__ lr_start /* [ws + 1] */, const int* __restrict__ bad, int* __restrict__ header) {
    __shared__ int part[256];
    __shared__ int sm[4];
    const int tid = threadIdx.x;
    auto scan = [&](const int* in, int n, int* out) -> int {
        const int per = (n + 255) / 256, b = tid * per, e = min(b + per, n);
        int s = 0;
        for (int k = b; k < e; ++k) s += in[k];
        __syncthreads();
        part[tid] = s;
        __syncthreads();
        int before = 0;
        for (int k = 0; k < tid; ++k) before += part[k];
        int total = 0;
        for (int k = 0; k < 256; ++k) total += part[k];
        for (int k = b; k < e; ++k) { out[k] = before; before += in[k]; }
        if (tid == 0) out[n] = total;
        return total;
    };
    const int P = scan(col_count, w, col_start);
    const int Ps = scan(lr_count, ws, lr_start);
    int imin = INT_MAX, imax = -1, jmin = INT_MAX, jmax = -1;
    for (int j = tid; j < w; j += 256)
        if (col_count[j] > 0) { imin = min(imin, col_lo[j]); imax = max(imax, col_hi[j]); jmin = min(jmin, j); jmax = max(jmax, j); }
    imin = block_reduce_min(imin, sm); imax = block_reduce_max(imax, sm);
    jmin = block_reduce_min(jmin, sm); jmax = block_reduce_max(jmax, sm);
    if (tid == 0) { header[0] = P; header[1] = Ps; header[2] = imin; header[3] = imax; header[4] = jmin; header[5] = jmax; header[6] = *bad; }
}

struct FillArgs {
    const uint8_t* mb;          // [w][h] mask bytes
    const uint8_t* lr_full;     // [w/sf][h/sf]
    const int* col_start;
    int h, w, sf;
    int i_lo, j_lo, Hs;
    int* imask;                 // [P]
    int* gofp;                  // [P]
    uint8_t* flags;             // [plane], zeroed
};

// C. one block per image column: the compact index of every masked pixel, its grid offset, its structure byte
// (make_gradient's case distinction, SRPS.cu:31-46: forward difference where the next pixel is masked, else backward where the
// previous one is, else none)
__global__ __launch_bounds__(256) void k_struct_fill(FillArgs a) {
    __shared__ int sm[4];
    const int j = blockIdx.x, h = a.h;
    const int n_col = a.col_start[j + 1] - a.col_start[j];
    if (n_col == 0) return;
    const uint8_t* c0 = a.mb + (size_t)j * h;
    const uint8_t* cl = (j > 0) ? c0 - h : nullptr;
    const uint8_t* cr = (j + 1 < a.w) ? c0 + h : nullptr;
    const int gbase = (j - a.j_lo + PAD) * a.Hs - a.i_lo + PAD;
    const int hs = h / a.sf;
    const uint8_t* lrc = a.lr_full + (size_t)(j / a.sf) * hs;
    int run = a.col_start[j];
    for (int i0 = 0; i0 < h; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool on = i < h && c0[i] != 0;
        int total;
        const int rank = block_rank256(on, sm, total);
        if (on) {
            uint8_t f = F_MASK;
            if (i + 1 < h && c0[i + 1]) f |= F_FY; else if (i > 0 && c0[i - 1]) f |= F_BY;      // SRPS.cu:31-38
            if (cr && cr[i]) f |= F_FX; else if (cl && cl[i]) f |= F_BX;                         // SRPS.cu:39-46
            if (lrc[i / a.sf]) f |= F_KB;                                                         // SRPS.cu:176-183
            const int p = run + rank;
            a.imask[p] = j * h + i;
            a.gofp[p] = gbase + i;
            a.flags[gbase + i] = f;
        }
        run += total;
    }
}

// D. one block per low-resolution column of the bounding box: compact LR index of every block (-1: not complete) and the LR
// linear index of every complete block (imasks, SRPS.cu:163-168)
__global__ __launch_bounds__(256) void k_struct_lr_index(const uint8_t* __restrict__ lr_full, const int* __restrict__ lr_start, int hs, int bi_lo, int bj_lo, int Hl,
                                                         int* __restrict__ lr_index /* [Wl][Hl] */, int* __restrict__ imasks /* [Ps] */) {
    __shared__ int sm[4];
    const int bjl = blockIdx.x, bj = bj_lo + bjl;
    const uint8_t* c = lr_full + (size_t)bj * hs;
    int run = lr_start[bj];
    // complete blocks of this column lie inside the box's rows (the box contains every masked pixel), so ranking the box's rows
    // ranks the column
    for (int b0 = 0; b0 < Hl; b0 += 256) {
        const int bil = b0 + threadIdx.x;
        const bool full = bil < Hl && c[bi_lo + bil] != 0;
        int total;
        const int rank = block_rank256(full, sm, total);
        if (bil < Hl) lr_index[(size_t)bjl * Hl + bil] = full ? run + rank : -1;
        if (full) imasks[run + rank] = bj * hs + bi_lo + bil;
        run += total;
    }
}

// E. classes of the resident CG's tiles (kernels_resident.hip; tile = 256 rows x tc columns, tile index = column of tiles * tiles
// per column + row of tiles), one block per tile, thread r = row r of the tile: TILE_RECT when every pixel of the tile is masked
// and inside a complete KT block and the ring row below / ring column to the right is either wholly masked (with no backward
// difference pointing into the tile) or wholly empty -- then the only backward differences of the tile are those of its last
// row / last column, which the RECT body handles without structure bits.
__global__ __launch_bounds__(256) void k_struct_tiles(const uint8_t* __restrict__ flags, int Hg, int Wg, int Hs, int Ws, int tc, int nbr,
                                                      uint8_t* __restrict__ cls, int* __restrict__ n_rect) {
    __shared__ int sm[4];
    const int tile = blockIdx.x, bc = tile / nbr, br = tile - bc * nbr;
    const int r0 = br * 256, c0 = bc * tc, r = threadIdx.x;
    auto F = [&](int rr, int cc) -> unsigned {
        const int sr = rr + PAD, sc = cc + PAD;
        return (sr < 0 || sr >= Hs || sc < 0 || sc >= Ws) ? 0u : (unsigned)flags[(size_t)sc * Hs + sr];
    };
    const bool inside = r0 + 256 <= Hg && c0 + tc <= Wg;
    int ok = 1, occ = 0;
    for (int c = 0; c < tc; ++c) {
        const unsigned f = F(r0 + r, c0 + c);
        ok &= (f & (F_MASK | F_KB)) == (F_MASK | F_KB);
        occ |= (f & F_MASK) ? 1 : 0;
    }
    const unsigned fr = F(r0 + r, c0 + tc);                              // ring column to the right: one pixel per thread
    const unsigned fb = (r < tc) ? F(r0 + 256, c0 + r) : 0u;             // ring row below: the first tc threads
    int bad = ((fr & F_BX) != 0) | ((r < tc) && (fb & F_BY) != 0);
    const int all_ok = block_reduce_min(ok, sm);
    const int any_occ = block_reduce_max(occ, sm);
    const int any_bad = block_reduce_max(bad, sm);
    const int nr = block_reduce_sum((fr & F_MASK) ? 1 : 0, sm);
    const int nb = block_reduce_sum((r < tc && (fb & F_MASK)) ? 1 : 0, sm);
    if (threadIdx.x == 0) {
        uint8_t v = any_occ ? TILE_OCCUPIED : 0;
        if (inside && all_ok && !any_bad && (nb == 0 || nb == tc) && (nr == 0 || nr == 256)) {
            v |= (uint8_t)(TILE_RECT | (nb == 0 ? TILE_BOTTOM_EMPTY : 0) | (nr == 0 ? TILE_RIGHT_EMPTY : 0));
            atomicAdd(n_rect, 1);
        }
        cls[tile] = v;
    }
}

__global__ void k_gather_int_index(const float* __restrict__ full, const int* __restrict__ index, int n, float* __restrict__ out) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) out[p] = full[index[p]];
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------------------------------------
// phase 1: mask (device, floats) -> per-column counts, prefix sums, LR completeness, the header (left in d_header; the caller
// copies it back and waits).  The scratch arrays live in `ws` (grow-only), laid out by struct_scratch().
size_t struct_scratch_bytes(int h, int w, int sf) {
    const size_t hw = (size_t)h * w, ws = (size_t)(w / sf), hs = (size_t)(h / sf);
    // mb [hw] | lr_full [ws*hs] | ints: col_count [w], col_lo [w], col_hi [w], col_start [w+1], lr_count [ws], lr_start [ws+1], bad [1], header [16], n_rect [4]
    return ((hw + 15) & ~(size_t)15) + ((ws * hs + 15) & ~(size_t)15) + sizeof(int) * (4 * (size_t)w + 2 * ws + 2 + 1 + 16 + 4 + 16);
}
StructScratch struct_scratch(void* base, int h, int w, int sf) {
    const size_t hw = (size_t)h * w, ws = (size_t)(w / sf), hs = (size_t)(h / sf);
    StructScratch s;
    uint8_t* b = (uint8_t*)base;
    s.mb = b; b += (hw + 15) & ~(size_t)15;
    s.lr_full = b; b += (ws * hs + 15) & ~(size_t)15;
    int* ip = (int*)b;
    s.col_count = ip; ip += w;
    s.col_lo = ip; ip += w;
    s.col_hi = ip; ip += w;
    s.col_start = ip; ip += w + 1;
    s.lr_count = ip; ip += ws;
    s.lr_start = ip; ip += ws + 1;
    s.bad = ip; ip += 1;
    s.header = ip; ip += 16;
    s.n_rect = ip; ip += 4;
    return s;
}

int struct_phase1(hipStream_t st, const float* d_mask, int h, int w, int sf, const StructScratch& s) {
    SRPS_HIP(hipMemsetD32Async((hipDeviceptr_t)s.bad, INT_MAX, 1, st));
    SRPS_HIP(hipMemsetAsync(s.n_rect, 0, 4 * sizeof(int), st));
    hipLaunchKernelGGL(k_struct_columns, dim3(w), dim3(256), 0, st, d_mask, h, s.mb, s.col_count, s.col_lo, s.col_hi, s.bad);
    hipLaunchKernelGGL(k_struct_lr_full, dim3(w / sf), dim3(256), 0, st, (const uint8_t*)s.mb, h, sf, s.lr_full, s.lr_count);
    hipLaunchKernelGGL(k_struct_prefix, dim3(1), dim3(256), 0, st, (const int*)s.col_count, (const int*)s.col_lo, (const int*)s.col_hi, w, s.col_start,
                       (const int*)s.lr_count, w / sf, s.lr_start, (const int*)s.bad, s.header);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// phase 2: with the grid's extent known and its arrays allocated: index maps, structure bytes (d_flags zeroed here), LR indices,
// tile classes (their counts are left in s.n_rect[0..2])
int struct_phase2(hipStream_t st, Grid& G, const StructScratch& s, int* d_imasks) {
    SRPS_HIP(hipMemsetAsync(G.d_flags, 0, G.plane, st));
    FillArgs a;
    a.mb = s.mb; a.lr_full = s.lr_full; a.col_start = s.col_start; a.h = G.h; a.w = G.w; a.sf = G.sf;
    a.i_lo = G.i_lo; a.j_lo = G.j_lo; a.Hs = G.Hs; a.imask = G.d_imask; a.gofp = G.d_gofp; a.flags = G.d_flags;
    hipLaunchKernelGGL(k_struct_fill, dim3(G.w), dim3(256), 0, st, a);
    if (G.Wl > 0 && G.Hl > 0)
        hipLaunchKernelGGL(k_struct_lr_index, dim3(G.Wl), dim3(256), 0, st, (const uint8_t*)s.lr_full, (const int*)s.lr_start, G.h / G.sf, G.i_lo / G.sf, G.j_lo / G.sf,
                           G.Hl, G.d_lr_index, d_imasks);
    for (int shape = 0; shape < 3; ++shape) {                    // [0] 256 x 32 tiles, [1] 256 x 64 tiles, [2] 256 x 16 tiles
        const int tc = shape == 0 ? 32 : shape == 1 ? 64 : 16, nbr = cdiv(G.Hg, 256), nbc = cdiv(G.Wg, tc);
        hipLaunchKernelGGL(k_struct_tiles, dim3(nbr * nbc), dim3(256), 0, st, (const uint8_t*)G.d_flags, G.Hg, G.Wg, G.Hs, G.Ws, tc, nbr, G.d_tile_cls[shape], s.n_rect + shape);
    }
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

int launch_gather_index(hipStream_t st, const float* d_full, const int* d_index, int n, float* d_out) {
    if (n <= 0) return SRPS_OK;
    hipLaunchKernelGGL(k_gather_int_index, dim3(std::min(cdiv(n, 256), 4096)), dim3(256), 0, st, d_full, d_index, n, d_out);
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

}  // namespace srps
