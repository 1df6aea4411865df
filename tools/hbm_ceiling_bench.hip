// hbm_ceiling_bench.hip -- what an HBM-bound kernel can reach on THIS box, and whether the number of concurrent plane streams of
// the streaming depth-CG step (8 planes read + 4 written, one 1-KiB wave access per plane and column) is what holds it at
// 0.53 - 0.62 of the 8 TB/s peak (round-2 review, weak #5):
//   linear_read / linear_copy / linear_rw21   grid-stride float4 streams over ~800 MB: read only, 1:1 copy, 2 reads : 1 write
//   march_planar                              the CG step's shape: a wave marches over columns, 8 planes read, 4 written per column
//   march_paired                              same bytes, (p, r) and (x, omega) stored as interleaved pairs: 5 read + 2 write streams
//   march_records                             same bytes, ONE 32-byte read record and ONE 16-byte write record per pixel: 1 + 1 streams
// All at two waves per SIMD (the occupancy of k_apply_march<., 3, 3>), loads of column c+1 and c+2 in flight.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_ceiling_bench.hip -o tools/hbm_ceiling_bench.bin ; tools/hbm_ceiling_bench.bin [rows=4096] [cols=4096] [reps=20] [guide_sweep=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));
#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_linear_read(const float4* __restrict__ a, size_t n4, float* __restrict__ out) {
    float s = 0.f;
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    for (; i + 3 * st < n4; i += 4 * st) {
        const float4 v0 = a[i], v1 = a[i + st], v2 = a[i + 2 * st], v3 = a[i + 3 * st];
        s += (v0.x + v1.y) + (v2.z + v3.w);
    }
    for (; i < n4; i += st) s += a[i].x;
    if (s == 1.2345f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_linear_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    for (; i + 3 * st < n4; i += 4 * st) {
        const float4 v0 = a[i], v1 = a[i + st], v2 = a[i + 2 * st], v3 = a[i + 3 * st];
        b[i] = v0; b[i + st] = v1; b[i + 2 * st] = v2; b[i + 3 * st] = v3;
    }
    for (; i < n4; i += st) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_linear_rw21(const float4* __restrict__ a, const float4* __restrict__ c, float4* __restrict__ b, size_t n4) {
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    for (; i + st < n4; i += 2 * st) {
        const float4 v0 = a[i], v1 = a[i + st], w0 = c[i], w1 = c[i + st];
        b[i] = make_float4(v0.x + w0.x, v0.y + w0.y, v0.z + w0.z, v0.w + w0.w);
        b[i + st] = make_float4(v1.x + w1.x, v1.y + w1.y, v1.z + w1.z, v1.w + w1.w);
    }
    for (; i < n4; i += st) b[i] = a[i];
}


// ---- round 4: the guide's recipe, swept (MI355X_MICROARCH.md:36 "6.29 TB/s measured (float4 copy)", :351 "1.2 GB table swept in order
// 6.0 - 6.1 TB/s"; cdna_hip_programming.md Guideline 11: 256 CUs x 8 blocks of 256 threads, grid-stride, 16 B per lane) ----
// U loads of 16 B in flight per lane; NT: non-temporal loads / stores; CONTIG: every block sweeps one contiguous chunk instead of
// striding over the whole array (the DRAM pages a block touches then stay its own).
template <int U, bool NT, bool CONTIG, bool WRITE>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ a_, float4* __restrict__ b_, size_t n4, float* __restrict__ out) {
    const vf4* __restrict__ a = reinterpret_cast<const vf4*>(a_);
    vf4* __restrict__ b = reinterpret_cast<vf4*>(b_);
    size_t i, end, st;
    if (CONTIG) {
        const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
        i = blockIdx.x * per + threadIdx.x; end = min(n4, (blockIdx.x + 1) * per); st = 256;
    } else { i = blockIdx.x * (size_t)256 + threadIdx.x; end = n4; st = (size_t)gridDim.x * 256; }
    float s = 0.f;
    for (; i + (U - 1) * st < end; i += U * st) {
        vf4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * st) : a[i + u * st];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (WRITE) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * st); else b[i + u * st] = v[u]; }
            else s += v[u].x + v[u].w;
        }
    }
    for (; i < end; i += st) { const vf4 v = a[i]; if (WRITE) b[i] = v; else s += v.x; }
    if (!WRITE && s == 1.2345f) out[0] = s;
}

struct F4 { float e[4]; };
template <bool NT = false>
__device__ __forceinline__ F4 ld4(const float* p) {
    if constexpr (NT) { const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p)); return F4{{t.x, t.y, t.z, t.w}}; }
    else { const float4 t = *reinterpret_cast<const float4*>(p); return F4{{t.x, t.y, t.z, t.w}}; }
}
template <bool NT = false>
__device__ __forceinline__ void st4(float* p, const F4& a) {
    if constexpr (NT) { vf4 t; t.x = a.e[0]; t.y = a.e[1]; t.z = a.e[2]; t.w = a.e[3]; __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p)); }
    else *reinterpret_cast<float4*>(p) = make_float4(a.e[0], a.e[1], a.e[2], a.e[3]);
}

// LAYOUT 0: 8 read planes, 4 write planes (element = float).  1: reads g0, g1, g2 (3 planes) + [p r] + [x w] (2 planes of pairs),
// writes [p r] + [x w].  2: reads one plane of 8-float records, writes one plane of 4-float records.
struct MArgs {
    const float* in[8];
    float* out[4];
    int Hs, cols_per_wave, n_items, n_seg, cols_total;
};
// LAYOUT 3 (round 4): the planar layout with every plane stored SEGMENT-major, [segment][column][256 rows]: a wave's march over its
// columns is one contiguous run of 1 KiB x columns per plane instead of 1 KiB every Hs * 4 bytes.  NT: non-temporal loads and stores.
template <int LAYOUT, bool NT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_march(MArgs a, float* chk_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (item >= a.n_items) return;
    const int strip = item / a.n_seg, seg = item - strip * a.n_seg;
    const int row = seg * 256 + lane * 4;
    const int c0 = strip * a.cols_per_wave;
    float chk = 0.f;
    struct Raw { F4 v[8]; };
    auto issue = [&](Raw& r, int col) {
        const size_t e = LAYOUT == 3 ? ((size_t)seg * a.cols_total + (c0 + col)) * 256 + lane * 4 : (size_t)(c0 + col) * a.Hs + row;      // element index of the lane's first row
        if (LAYOUT == 0 || LAYOUT == 3) {
#pragma unroll
            for (int p = 0; p < 8; ++p) r.v[p] = ld4<NT>(a.in[p] + e);
        } else if (LAYOUT == 1) {
#pragma unroll
            for (int p = 0; p < 3; ++p) r.v[p] = ld4<NT>(a.in[p] + e);
            r.v[3] = ld4<NT>(a.in[3] + 2 * e); r.v[4] = ld4<NT>(a.in[3] + 2 * e + 4);
            r.v[5] = ld4<NT>(a.in[4] + 2 * e); r.v[6] = ld4<NT>(a.in[4] + 2 * e + 4);
            r.v[7] = r.v[0];                                    // (the structure bytes: a quarter plane, left out)
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p) r.v[p] = ld4<NT>(a.in[0] + 8 * e + 4 * p);
        }
    };
    auto consume = [&](const Raw& r, int col) {
        const size_t e = LAYOUT == 3 ? ((size_t)seg * a.cols_total + (c0 + col)) * 256 + lane * 4 : (size_t)(c0 + col) * a.Hs + row;
        F4 o[4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) o[w].e[q] = fmaf(r.v[w].e[q], 0.5f, r.v[w + 4].e[q]);
        if (LAYOUT == 0 || LAYOUT == 3) {
#pragma unroll
            for (int w = 0; w < 4; ++w) st4<NT>(a.out[w] + e, o[w]);
        } else if (LAYOUT == 1) {
            st4<NT>(a.out[0] + 2 * e, o[0]); st4<NT>(a.out[0] + 2 * e + 4, o[1]);
            st4<NT>(a.out[1] + 2 * e, o[2]); st4<NT>(a.out[1] + 2 * e + 4, o[3]);
        } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) st4<NT>(a.out[0] + 4 * e + 4 * w, o[w]);
        }
        chk += o[0].e[0];
    };
    Raw b0, b1;
    issue(b0, 0); issue(b1, min(1, a.cols_per_wave - 1));
    for (int c = 0; c < a.cols_per_wave; c += 2) {
        { const Raw cur = b0; issue(b0, min(c + 2, a.cols_per_wave - 1)); consume(cur, c); }
        { const Raw cur = b1; issue(b1, min(c + 3, a.cols_per_wave - 1)); consume(cur, c + 1); }
    }
    if (chk == 1.2345f) chk_out[0] = chk;
}


// ---- round 4: the image sweeps' shape.  A block of 256 threads owns 1024 consecutive pixels and reads them from NP planes (the
// (image, channel) rows of I[n][c][P], P * 4 bytes apart), four loads in flight per lane (k_albedo_numden) -- against the same bytes
// stored TILE-major, [tile][plane][1024 pixels]: the block's NP x 4 KiB are one contiguous run.
template <bool NT, bool TILED>
__global__ __launch_bounds__(256) void k_sweep(const float* __restrict__ I, size_t P, int NP, float* __restrict__ out) {
    const size_t q = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (q >= P) return;
    float s = 0.f;
#pragma unroll 4
    for (int pl = 0; pl < NP; ++pl) {
        const float* src = TILED ? I + ((size_t)blockIdx.x * NP + pl) * 1024 + threadIdx.x * 4 : I + (size_t)pl * P + q;
        const F4 v = ld4<NT>(src);
        s = fmaf(v.e[0], v.e[1], s) + v.e[2] * v.e[3];
    }
    if (s == 1.2345f) out[0] = s;
}

// ---- round 6: what separates k_sweep (6.85 TB/s from the tiles) from the library's sweeps (5.4 - 5.5 TB/s, tiles or planes alike)?
// The same loop with (a) the OCCUPANCY of the library's kernels (blocks per CU capped by a dynamic LDS allocation: a block of 256 threads is
// one wave per SIMD), (b) U loads in flight per lane, (c) the albedo sweep's SIDE streams: 8 planes read before the images (normals, dz,
// xx, yy, the grid map), per channel a plane read and written (rho) and a plane stored (g), three planes stored at the end (q).
// SIDE bits: 1 side reads, 2 side writes, 4 the writes non-temporal, 8 the side reads non-temporal
template <bool NT, bool TILED, int U, int SIDE>
__global__ __launch_bounds__(256) void k_sweep_occ(const float* __restrict__ I, size_t P, int NP, const float* __restrict__ side_in, float* __restrict__ side_out, float* __restrict__ out) {
    extern __shared__ float lds_cap[];
    constexpr bool RD = SIDE & 1, WR = SIDE & 2, WNT = SIDE & 4, RNT = SIDE & 8;
    const size_t q = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (q >= P) return;
    float s = 0.f;
    if (RD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const F4 v = ld4<RNT>(side_in + (size_t)k * P + q); s += v.e[0] * v.e[3]; }
    }
    const int C = 3, NI = NP / C;
    for (int c = 0; c < C; ++c) {
        float a = 0.f;
#pragma unroll U
        for (int i = 0; i < NI; ++i) {
            const int pl = TILED ? c * NI + i : i * C + c;                       // tile-major rows are channel-major; the planes are image-major
            const float* src = TILED ? I + ((size_t)blockIdx.x * NP + pl) * 1024 + threadIdx.x * 4 : I + (size_t)pl * P + q;
            const F4 v = ld4<NT>(src);
            a = fmaf(v.e[0], v.e[1], a) + v.e[2] * v.e[3];
        }
        s += a;
        F4 r; r.e[0] = r.e[1] = r.e[2] = r.e[3] = a;
        if (RD) r = ld4<RNT>(side_in + (size_t)(8 + c) * P + q);
        if (WR) {
            r.e[0] += a; st4<WNT>(side_out + (size_t)c * P + q, r);              // rho: read, written
            r.e[1] += a; st4<WNT>(side_out + (size_t)(3 + c) * P + q, r);        // g_c
        } else s += r.e[2];
    }
    if (WR) {
#pragma unroll
        for (int t = 0; t < 3; ++t) { F4 r; r.e[0] = r.e[1] = r.e[2] = r.e[3] = s + t; st4<WNT>(side_out + (size_t)(6 + t) * P + q, r); }
    }
    if (s == 1.2345f) out[0] = s + lds_cap[0];
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 4096, cols = argc > 2 ? atoi(argv[2]) : 4096, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int Hs = rows + 32;
    const size_t pl = (size_t)Hs * (cols + 8);                 // elements of one plane
    // one arena of 12 planes, carved differently per layout
    float* arena; CHECK(hipMalloc(&arena, 12 * pl * sizeof(float)));
    CHECK(hipMemset(arena, 0, 12 * pl * sizeof(float)));
    float* chk; CHECK(hipMalloc(&chk, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto launch) {
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("{\"variant\": \"%s\", \"MB\": %.1f, \"us\": %.1f, \"GBs\": %.0f, \"frac_of_8TBs\": %.3f}\n", name, bytes * 1e-6, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
    };
    const size_t n4 = 12 * pl / 4;                              // float4 elements of the arena (~806 MB at 4096^2)
    for (int nb : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, sizeof nm, "linear_read_%dblk", nb);
        timeit(nm, 16.0 * n4, [&] { hipLaunchKernelGGL(k_linear_read, dim3(nb), dim3(256), 0, 0, (const float4*)arena, n4, chk); });
        snprintf(nm, sizeof nm, "linear_copy_%dblk", nb);
        timeit(nm, 16.0 * n4, [&] { hipLaunchKernelGGL(k_linear_copy, dim3(nb), dim3(256), 0, 0, (const float4*)arena, (float4*)arena + n4 / 2, n4 / 2); });
        snprintf(nm, sizeof nm, "linear_rw21_%dblk", nb);
        timeit(nm, 16.0 * n4, [&] { hipLaunchKernelGGL(k_linear_rw21, dim3(nb), dim3(256), 0, 0, (const float4*)arena, (const float4*)arena + n4 / 3, (float4*)arena + 2 * (n4 / 3), n4 / 3); });
    }
    MArgs a{};
    a.Hs = Hs; a.n_seg = rows / 256;
    const double bytes = (double)rows * cols * 4.0 * 12;
    const int LDSB = 80 * 1024;                                  // two blocks per CU = two waves per SIMD whatever the register count
    CHECK(hipFuncSetAttribute((const void*)k_march<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    CHECK(hipFuncSetAttribute((const void*)k_march<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    CHECK(hipFuncSetAttribute((const void*)k_march<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    CHECK(hipFuncSetAttribute((const void*)k_march<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    CHECK(hipFuncSetAttribute((const void*)k_march<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    CHECK(hipFuncSetAttribute((const void*)k_march<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    for (int waves_target : {2048, 1024}) {
        int strips = waves_target / a.n_seg; if (strips < 1) strips = 1;
        while (cols % strips) --strips;
        a.cols_per_wave = cols / strips; a.n_items = strips * a.n_seg;
        const int nb = (a.n_items + 3) / 4;
        printf("march: rows %d cols %d: %d waves, %d columns each, %.1f MB per pass\n", rows, cols, a.n_items, a.cols_per_wave, bytes * 1e-6);
        for (int p = 0; p < 8; ++p) a.in[p] = arena + (size_t)p * pl;
        for (int w = 0; w < 4; ++w) a.out[w] = arena + (size_t)(8 + w) * pl;
        a.cols_total = cols;
        timeit("march_planar_8r_4w", bytes, [&] { hipLaunchKernelGGL((k_march<0>), dim3(nb), dim3(256), LDSB, 0, a, chk); });
        timeit("march_planar_nt", bytes, [&] { hipLaunchKernelGGL((k_march<0, true>), dim3(nb), dim3(256), LDSB, 0, a, chk); });
        timeit("march_segmajor", bytes, [&] { hipLaunchKernelGGL((k_march<3>), dim3(nb), dim3(256), LDSB, 0, a, chk); });
        timeit("march_segmajor_nt", bytes, [&] { hipLaunchKernelGGL((k_march<3, true>), dim3(nb), dim3(256), LDSB, 0, a, chk); });
        // pairs: g0 g1 g2 | [p r] (2 planes) | [x w] (2 planes) read; [p r] | [x w] written
        a.in[0] = arena; a.in[1] = arena + pl; a.in[2] = arena + 2 * pl; a.in[3] = arena + 3 * pl; a.in[4] = arena + 5 * pl;
        a.out[0] = arena + 7 * pl; a.out[1] = arena + 9 * pl;
        timeit("march_paired_5r_2w", bytes * 11.0 / 12.0, [&] { hipLaunchKernelGGL((k_march<1>), dim3(nb), dim3(256), LDSB, 0, a, chk); });
        a.in[0] = arena; a.out[0] = arena + 8 * pl;
        timeit("march_records_1r_1w", bytes, [&] { hipLaunchKernelGGL((k_march<2>), dim3(nb), dim3(256), LDSB, 0, a, chk); });
    }
    // the guide's recipe swept: blocks x loads in flight x cache policy x block-contiguous chunks; read of 1.2 GB (guide :351) and
    // copy of 2 x 0.6 GB; also what the runtime's own device-to-device copy and memset reach
    const int mode = argc > 4 ? atoi(argv[4]) : 0;      // 1: everything; 2: the image sweeps' shapes only (what tools/tcc_sweeps.sh profiles)
    if (mode != 0) {
        const size_t big4 = ((size_t)1200 << 20) / 16;
        float4* big; CHECK(hipMalloc(&big, big4 * 16));
        CHECK(hipMemset(big, 0, big4 * 16));
#define SWEEP(U, NT, CONTIG)                                                                                                       \
        for (int nb : {1024, 2048, 4096, 16384}) {                                                                                   \
            char nm[96];                                                                                                           \
            snprintf(nm, sizeof nm, "guide_read_u%d%s%s_%dblk", U, NT ? "_nt" : "", CONTIG ? "_contig" : "", nb);                  \
            timeit(nm, 16.0 * big4, [&] { hipLaunchKernelGGL((k_stream<U, NT, CONTIG, false>), dim3(nb), dim3(256), 0, 0, (const float4*)big, (float4*)nullptr, big4, chk); });   \
            snprintf(nm, sizeof nm, "guide_copy_u%d%s%s_%dblk", U, NT ? "_nt" : "", CONTIG ? "_contig" : "", nb);                  \
            timeit(nm, 16.0 * big4, [&] { hipLaunchKernelGGL((k_stream<U, NT, CONTIG, true>), dim3(nb), dim3(256), 0, 0, (const float4*)big, big + big4 / 2, big4 / 2, chk); });  \
        }
        if (mode == 1) {
        SWEEP(1, false, false) SWEEP(4, false, false) SWEEP(8, false, false) SWEEP(4, true, false) SWEEP(8, true, false)
        SWEEP(4, false, true) SWEEP(8, true, true)
        }
#undef SWEEP
        {
            const size_t Pp = (size_t)2048 * 2048; const int NP = 60;          // 1.0066 GB: the images of the metric's configuration
            float* img; CHECK(hipMalloc(&img, Pp * NP * 4)); CHECK(hipMemset(img, 0, Pp * NP * 4));
            const int nbs = (int)(Pp / 1024);
            timeit("sweep_60planes", 4.0 * Pp * NP, [&] { hipLaunchKernelGGL((k_sweep<false, false>), dim3(nbs), dim3(256), 0, 0, img, Pp, NP, chk); });
            timeit("sweep_60planes_nt", 4.0 * Pp * NP, [&] { hipLaunchKernelGGL((k_sweep<true, false>), dim3(nbs), dim3(256), 0, 0, img, Pp, NP, chk); });
            timeit("sweep_tilemajor", 4.0 * Pp * NP, [&] { hipLaunchKernelGGL((k_sweep<false, true>), dim3(nbs), dim3(256), 0, 0, img, Pp, NP, chk); });
            timeit("sweep_tilemajor_nt", 4.0 * Pp * NP, [&] { hipLaunchKernelGGL((k_sweep<true, true>), dim3(nbs), dim3(256), 0, 0, img, Pp, NP, chk); });
            // round 6: occupancy x loads in flight x side streams (see k_sweep_occ)
            float *sin, *sout; CHECK(hipMalloc(&sin, Pp * 11 * 4)); CHECK(hipMalloc(&sout, Pp * 9 * 4)); CHECK(hipMemset(sin, 0, Pp * 11 * 4));
#define OCC_CASE(NTV, TIL, UU, SD, nm_) do {                                                                                            \
                CHECK(hipFuncSetAttribute((const void*)k_sweep_occ<NTV, TIL, UU, SD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));   \
                for (int occ : {8, 4}) {                                                                                                   \
                    char nm[96]; snprintf(nm, sizeof nm, "%s_u%d_%dwaves", nm_, UU, occ);                                               \
                    const int lds = occ >= 8 ? 0 : (160 * 1024 / occ) - 1024;                                                          \
                    const double sb = 4.0 * Pp * (((SD) & 1 ? 11 : 0) + ((SD) & 2 ? 9 : 0));                                           \
                    timeit(nm, 4.0 * Pp * NP + sb, [&] { hipLaunchKernelGGL((k_sweep_occ<NTV, TIL, UU, SD>), dim3(nbs), dim3(256), lds, 0, img, Pp, NP, sin, sout, chk); }); \
                } } while (0)
            OCC_CASE(true, true, 4, 0, "occ_tiles_nt");
            OCC_CASE(true, false, 4, 0, "occ_planes_nt");
            OCC_CASE(true, false, 8, 0, "occ_planes_nt");
            OCC_CASE(true, true, 4, 1, "occ_tiles_nt_sideR");
            OCC_CASE(true, true, 4, 9, "occ_tiles_nt_sideRnt");
            OCC_CASE(true, true, 4, 2, "occ_tiles_nt_sideW");
            OCC_CASE(true, true, 4, 6, "occ_tiles_nt_sideWnt");
            OCC_CASE(true, true, 4, 3, "occ_tiles_nt_sideRW");
            OCC_CASE(true, true, 4, 7, "occ_tiles_nt_sideRWnt");
            OCC_CASE(true, true, 4, 15, "occ_tiles_nt_sideRntWnt");
            OCC_CASE(true, false, 4, 3, "occ_planes_nt_sideRW");
            OCC_CASE(true, false, 4, 7, "occ_planes_nt_sideRWnt");
            OCC_CASE(true, false, 4, 15, "occ_planes_nt_sideRntWnt");
#undef OCC_CASE
            CHECK(hipFree(sin)); CHECK(hipFree(sout));
            CHECK(hipFree(img));
        }
        if (mode == 1) timeit("hipMemcpyDtoD_600MB", 16.0 * big4, [&] { CHECK(hipMemcpyAsync(big + big4 / 2, big, big4 / 2 * 16, hipMemcpyDeviceToDevice, 0)); });
        if (mode == 1) timeit("hipMemset_1200MB", 16.0 * big4, [&] { CHECK(hipMemsetAsync(big, 0, big4 * 16, 0)); });
        CHECK(hipFree(big));
    }
    CHECK(hipFree(arena));
    return 0;
}
