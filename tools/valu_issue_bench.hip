// valu_issue_bench.hip -- how many SIMD clocks does one wave64 VALU instruction occupy on gfx950?
//
// Settles the assumption behind the VALU roofline of k_cg_resident (profiles/README.md): the guide says a wave64
// v_fma_f32 issues over 2 cycles on CDNA4's SIMD-32 when the SIMD has two or more waves to pick from and 4 when a wave
// is alone; what v_pk_fma_f32, v_pk_mul_f32, v_bfe_i32, v_and_b32 and DPP moves cost is measured here as well.
//
// One block per CU (the dynamic LDS allocation excludes a second one), 256 / 512 / 1024 threads = 1 / 2 / 4 waves per
// SIMD.  Every wave runs ITER iterations of 64 independent instructions (8 accumulators x 8, no dependent chain
// shorter than 8 instructions).  Time by HIP events around the launch (minimum of 5); clocks = time x f / (instructions
// per wave x waves per SIMD), f = the shader clock, which the driver script reads from sysfs while the kernel runs
// (2.4 GHz nominal when it cannot).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_issue_bench.bin tools/valu_issue_bench.hip
//   tools/valu_issue_bench.bin [GHz]          -> one JSON object per line
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X

enum { K_FMA = 0, K_PK_FMA, K_PK_MUL, K_PK_ADD, K_ADD, K_BFE, K_AND, K_DPP_MOV, K_FMA_DPP_MIX, K_READLANE, K_CNDMASK_SGPR, K_FMAC_SGPR, K_PK_FMA_SGPR, K_CNDMASK_VCC, K_ADD_SGPR, K_FMA_LITERAL, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_add_f32", "v_bfe_i32",
                                      "v_and_b32", "v_mov_b32_dpp(wave_shr:1)", "v_pk_fma_f32+v_bfe_i32 (1:1)", "v_readlane_b32 (SGPR spill reload)",
                                      "v_cndmask_b32 (mask in an SGPR pair)", "v_fmac_f32 (one SGPR operand)",
                                      "v_pk_fma_f32 (one SGPR-pair operand)", "v_cndmask_b32 (mask in vcc)", "v_add_f32 (one SGPR operand)",
                                      "v_fma_f32 (one literal constant)"};

template <int KIND>
__global__ void k_issue(float* out, int iters, float seed) {
    extern __shared__ float lds[];
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float x = 0.999f, y = 1e-3f;
    const v2f xx = {x, x}, yy = {y, y};
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_FMA) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (KIND == K_ADD) {
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
        } else if (KIND == K_PK_FMA) {
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(xx), "v"(yy));
        } else if (KIND == K_PK_MUL) {
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(xx));
        } else if (KIND == K_PK_ADD) {
            asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(yy));
        } else if (KIND == K_BFE) {
            asm volatile(REP8("v_bfe_i32 %0, %0, 3, 29\n v_bfe_i32 %1, %1, 3, 29\n v_bfe_i32 %2, %2, 3, 29\n v_bfe_i32 %3, %3, 3, 29\n"
                              "v_bfe_i32 %4, %4, 3, 29\n v_bfe_i32 %5, %5, 3, 29\n v_bfe_i32 %6, %6, 3, 29\n v_bfe_i32 %7, %7, 3, 29\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (KIND == K_AND) {
            asm volatile(REP8("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                              "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(0x7fffffff));
        } else if (KIND == K_DPP_MOV) {
            asm volatile(REP8("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (KIND == K_READLANE) {
            int s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile(REP8("v_readlane_b32 %0, %8, 1\n v_readlane_b32 %1, %9, 2\n v_readlane_b32 %2, %10, 3\n v_readlane_b32 %3, %11, 4\n"
                              "v_readlane_b32 %4, %12, 5\n v_readlane_b32 %5, %13, 6\n v_readlane_b32 %6, %14, 7\n v_readlane_b32 %7, %15, 8\n")
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
                         : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));
            i0 += s0 & 1; i1 += s1 & 1; i2 += s2 & 1; i3 += s3 & 1; i4 += s4 & 1; i5 += s5 & 1; i6 += s6 & 1; i7 += s7 & 1;      // 8 more (full-rate) instructions per 64
        } else if (KIND == K_CNDMASK_SGPR) {
            const unsigned long long m = 0x5555555555555555ull + (unsigned long long)it;
            asm volatile(REP8("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                              "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "s"(m));
        } else if (KIND == K_FMAC_SGPR) {
            float sc = x;
            asm volatile("" : "+s"(sc));
            asm volatile(REP8("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc), "v"(y));
        } else if (KIND == K_PK_FMA_SGPR) {
            v2f sc = xx;
            asm volatile("" : "+s"(sc));
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(sc), "v"(yy));
        } else if (KIND == K_CNDMASK_VCC) {
            asm volatile("s_mov_b64 vcc, 0x55555555\n"
                         REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y) : "vcc");
        } else if (KIND == K_ADD_SGPR) {
            float sc = y;
            asm volatile("" : "+s"(sc));
            asm volatile(REP8("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                              "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc));
        } else if (KIND == K_FMA_LITERAL) {
            asm volatile(REP8("v_fmac_f32 %0, 0x3a83126f, %8\n v_fmac_f32 %1, 0x3a83126f, %8\n v_fmac_f32 %2, 0x3a83126f, %8\n v_fmac_f32 %3, 0x3a83126f, %8\n"
                              "v_fmac_f32 %4, 0x3a83126f, %8\n v_fmac_f32 %5, 0x3a83126f, %8\n v_fmac_f32 %6, 0x3a83126f, %8\n v_fmac_f32 %7, 0x3a83126f, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
        } else {      // the resident kernel's mix: packed arithmetic interleaved with mask formation
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_bfe_i32 %4, %4, 3, 29\n v_pk_fma_f32 %1, %1, %8, %9\n v_bfe_i32 %5, %5, 3, 29\n"
                              "v_pk_fma_f32 %2, %2, %8, %9\n v_bfe_i32 %6, %6, 3, 29\n v_pk_fma_f32 %3, %3, %8, %9\n v_bfe_i32 %7, %7, 3, 29\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(xx), "v"(yy));
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y +
              p6.x + p6.y + p7.x + p7.y + (float)(i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7);
    if (r == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = r + lds[threadIdx.x];
}

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int KIND>
static int run(int cus, double ghz, float* d_out) {
    const int iters = 20000;
    const size_t lds = 96 * 1024;                       // one block per CU
    CHECK(hipFuncSetAttribute((const void*)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int threads : {256, 512, 1024}) {
        double best = 1e30;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_issue<KIND>, dim3(cus), dim3(threads), lds, 0, d_out, iters, 1.0f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double instr_per_wave = 64.0 * iters, waves_per_simd = threads / 256.0;
        const double ns_per_instr = best * 1e6 / (instr_per_wave * waves_per_simd);
        printf("{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.4f, \"ns_per_wave_instruction_per_simd\": %.4f, "
               "\"assumed_GHz\": %.3f, \"simd_clocks_per_wave_instruction\": %.3f}\n",
               kNames[KIND], threads / 256, best, ns_per_instr, ghz, ns_per_instr * ghz);
        fflush(stdout);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    int cus = 0;
    CHECK(hipSetDevice(0));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float* d_out = nullptr;
    CHECK(hipMalloc(&d_out, (size_t)cus * 1024 * sizeof(float)));
    if (run<K_FMA>(cus, ghz, d_out) || run<K_PK_FMA>(cus, ghz, d_out) || run<K_PK_MUL>(cus, ghz, d_out) || run<K_PK_ADD>(cus, ghz, d_out) ||
        run<K_ADD>(cus, ghz, d_out) || run<K_BFE>(cus, ghz, d_out) || run<K_AND>(cus, ghz, d_out) || run<K_DPP_MOV>(cus, ghz, d_out) ||
        run<K_FMA_DPP_MIX>(cus, ghz, d_out) || run<K_READLANE>(cus, ghz, d_out) || run<K_CNDMASK_SGPR>(cus, ghz, d_out) || run<K_FMAC_SGPR>(cus, ghz, d_out) ||
        run<K_PK_FMA_SGPR>(cus, ghz, d_out) || run<K_CNDMASK_VCC>(cus, ghz, d_out) || run<K_ADD_SGPR>(cus, ghz, d_out) || run<K_FMA_LITERAL>(cus, ghz, d_out))
        return 1;
    (void)hipFree(d_out);
    return 0;
}
