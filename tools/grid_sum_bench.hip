// grid_sum_bench.hip -- what a grid-wide sum of three numbers costs 256 resident blocks (one per CU), per generation, for two
// ways of exchanging the blocks' partial sums:
//   library:   the exchange k_cg_resident uses, called from device_utils.h unchanged (grid_sum3_publish / grid_sum3_collect: one
//              generation-tagged 16-byte granule per block, 256 B apart; one wave per block polls all of them, four per lane in
//              one batch, and adds them in a fixed order)
//   granules:  a plain restatement of the same idea (polls one granule after the other)
//   atomics R: every block adds its three partial sums, as 64-bit fixed-point numbers whose low 9 bits count the contributions,
//              to one of R records (device-scope atomic adds: exact and order-independent, so still bit-reproducible); a lane
//              per record polls until the counts are complete.  The accumulators are never reset: a reader subtracts the value
//              it saw two generations ago.
// Between two sums every block "computes" for a fixed time (s_sleep), as the CG step does; the figure printed is the time per
// generation minus that.
//   hipcc -O3 --offload-arch=gfx950 tools/grid_sum_bench.hip -o tools/grid_sum_bench.bin && tools/grid_sum_bench.bin
#include <hip/hip_runtime.h>
#include "../srmeetsps-cuda_amd/csrc/device_utils.h"      // the library's own exchange: variant "library"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NBMAX = 256, NT = 512, GENS = 200;
constexpr int STRIDE = 32;                       // 8-byte words between granules (256 B)

__device__ __forceinline__ void compute_for(unsigned long long ticks) {      // 100 MHz ticks
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

__global__ __launch_bounds__(NT) void k_granules(unsigned long long* ent, unsigned long long* t_out, float* sink, int work_ticks) {
    extern __shared__ float lds[];
    __shared__ float tot[4];
    const int tid = threadIdx.x, b = blockIdx.x, NB = gridDim.x;
    float acc = 0.f;
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    for (unsigned gen = 1; gen <= GENS; ++gen) {
        compute_for(work_ticks + ((b * 7 + gen * 3) & 15));                   // a little skew between the blocks
        __syncthreads();
        const float v0 = 1.f + b * 1e-3f + acc * 1e-9f, v1 = 0.5f, v2 = 0.25f;
        if (tid == 0) {
            unsigned long long* g = ent + ((size_t)(gen & 1u) * NBMAX + b) * STRIDE;
            const unsigned long long w0 = ((unsigned long long)gen << 32) | __float_as_uint(v0);
            const unsigned long long w1 = ((unsigned long long)__float_as_uint(v2) << 32) | __float_as_uint(v1);
            // one 16-byte store
            typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
            v2u val = {w0, w1};
            __builtin_nontemporal_store(val, reinterpret_cast<v2u*>(g));
            __threadfence();
        }
        if (tid < 64) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int q = 0; q < (NB + 63) / 64; ++q) {
                if (q * 64 + tid >= NB) break;
                const unsigned long long* g = ent + ((size_t)(gen & 1u) * NBMAX + q * 64 + tid) * STRIDE;
                unsigned long long w0, w1;
                do {
                    w0 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    w1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((unsigned)(w0 >> 32) != gen && __builtin_amdgcn_s_memrealtime() - t_begin < 5000000ull);      // 50 ms: never hang
                s0 += __uint_as_float((unsigned)w0); s1 += __uint_as_float((unsigned)w1); s2 += __uint_as_float((unsigned)(w1 >> 32));
            }
            for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            if (tid == 0) { tot[0] = s0; tot[1] = s1; tot[2] = s2; }
        }
        __syncthreads();
        acc += tot[0] + tot[1] + tot[2];
    }
    if (tid == 0) { t_out[b] = __builtin_amdgcn_s_memrealtime() - t_begin; sink[b] = acc; }
}

// the exchange k_cg_resident uses, unchanged: grid_sum3_publish<8, true> / grid_sum3_collect<true> of device_utils.h
__global__ __launch_bounds__(NT) void k_library(unsigned long long* ent3, unsigned long long* t_out, float* sink, int work_ticks, int* status) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    float acc = 0.f;
    srps::spin_guard_init(5000000ull, status, 1);                             // 50 ms: never hang
    srps::grid_sum3_prepare();
    __syncthreads();
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    for (unsigned gen = 1; gen <= GENS; ++gen) {
        compute_for(work_ticks + ((b * 7 + gen * 3) & 15));
        const float v0 = (tid == 0) ? 1.f + b * 1e-3f + acc * 1e-9f : 0.f, v1 = (tid == 0) ? 0.5f : 0.f, v2 = (tid == 0) ? -0.25f : 0.f;
        srps::grid_sum3_publish<NT / 64, true>(v0, v1, v2, ent3, gen);
        double o0, o1, o2;
        srps::grid_sum3_collect<true>(ent3, gen, o0, o1, o2);
        acc += (float)(o0 + o1 + o2);
    }
    if (tid == 0) { t_out[b] = __builtin_amdgcn_s_memrealtime() - t_begin; sink[b] = acc; }
}

template <int R>
__global__ __launch_bounds__(NT) void k_atomics(unsigned long long* rec, unsigned long long* t_out, float* sink, int work_ticks) {
    extern __shared__ float lds[];
    __shared__ double tot[4];
    const int tid = threadIdx.x, b = blockIdx.x, NB = gridDim.x;
    float acc = 0.f;
    // rec[parity][R][4 words, 256 B apart]; what the reader saw two generations ago, per parity and record
    long long seen[2][3] = {{0, 0, 0}, {0, 0, 0}};
    const int my = (b & 7) % R + ((b >> 3) % ((R + 7) / 8)) * 8;               // R <= 8: by XCD (blocks are dealt round-robin)
    const int PER = NB / R;                                                   // contributions per record (NB a multiple of R)
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    for (unsigned gen = 1; gen <= GENS; ++gen) {
        compute_for(work_ticks + ((b * 7 + gen * 3) & 15));
        __syncthreads();
        const double v[3] = {1.0 + b * 1e-3 + acc * 1e-9, 0.5, -0.25};
        if (tid < 3) {
            unsigned long long* a = rec + (((size_t)(gen & 1u) * R + my) * STRIDE) + tid;
            const long long fx = (long long)(v[tid] * 4294967296.0);            // 2^32 fixed point
            __hip_atomic_fetch_add(a, (unsigned long long)((fx << 9) | 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < 64) {
            long long d[3] = {0, 0, 0};
            if (tid < R) {
                const unsigned long long* a = rec + (((size_t)(gen & 1u) * R + tid) * STRIDE);
                bool done;
                do {
                    done = true;
                    for (int i = 0; i < 3; ++i) {
                        const long long cur = (long long)__hip_atomic_load(a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        d[i] = cur - seen[gen & 1u][i];
                        done &= (d[i] & 511) == (PER & 511);
                    }
                } while (!done && __builtin_amdgcn_s_memrealtime() - t_begin < 5000000ull);
                for (int i = 0; i < 3; ++i) { seen[gen & 1u][i] += d[i]; d[i] >>= 9; }
            }
            for (int o = 32; o > 0; o >>= 1)
                for (int i = 0; i < 3; ++i) d[i] += __shfl_xor(d[i], o);
            if (tid == 0) for (int i = 0; i < 3; ++i) tot[i] = (double)d[i] * (1.0 / 4294967296.0);
        }
        __syncthreads();
        acc += (float)(tot[0] + tot[1] + tot[2]);
    }
    if (tid == 0) { t_out[b] = __builtin_amdgcn_s_memrealtime() - t_begin; sink[b] = acc; }
}

int main(int argc, char** argv) {
    const int work_ticks = argc > 1 ? atoi(argv[1]) : 500;                    // 5 us of "compute" per generation
    const int NB = argc > 2 ? atoi(argv[2]) : 256;                            // resident blocks (a multiple of 32, at most 256)
    unsigned long long *ws, *t_out; float* sink;
    const size_t ws_n = (size_t)2 * NBMAX * STRIDE + 64;
    CHECK(hipMalloc(&ws, ws_n * 8)); CHECK(hipMalloc(&t_out, NB * 8)); CHECK(hipMalloc(&sink, NB * 4));
    const int LDSB = 100 * 1024;                                              // one block per CU
    auto run = [&](const char* name, const void* fn) {
        CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(ws, 0, ws_n * 8));
            int wt = work_ticks;
            void* args[] = {&ws, &t_out, &sink, &wt};
            CHECK(hipLaunchCooperativeKernel(fn, dim3(NB), dim3(NT), args, LDSB, 0));
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> t(NB); std::vector<float> s(NB);
            CHECK(hipMemcpy(t.data(), t_out, NB * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(s.data(), sink, NB * 4, hipMemcpyDeviceToHost));
            double mean = 0; for (auto v : t) mean += (double)v; mean /= NB;
            bool same = true; for (int b = 1; b < NB; ++b) same &= s[b] == s[0];
            if (rep == 2)
                printf("{\"blocks\": %d, \"variant\": \"%s\", \"us_per_generation\": %.3f, \"of_which_work\": %.2f, \"sum_cost_us\": %.3f, \"all_blocks_same_total\": %s, \"total\": %.6f}\n",
                       NB, name, mean * 0.01 / GENS, (work_ticks + 7.5) * 0.01, mean * 0.01 / GENS - (work_ticks + 7.5) * 0.01, same ? "true" : "false", s[0]);
        }
    };
    {   // the library's exchange: ent3 = [2][256] granules SRPS_G3_STRIDE apart
        unsigned long long* ent3; int* status;
        CHECK(hipMalloc(&ent3, (size_t)2 * 256 * SRPS_G3_STRIDE)); CHECK(hipMalloc(&status, 64));
        CHECK(hipFuncSetAttribute((const void*)k_library, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(ent3, 0, (size_t)2 * 256 * SRPS_G3_STRIDE)); CHECK(hipMemset(status, 0, 64));
            int wt = work_ticks;
            void* args[] = {&ent3, &t_out, &sink, &wt, &status};
            CHECK(hipLaunchCooperativeKernel((const void*)k_library, dim3(NB), dim3(NT), args, LDSB, 0));
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> t(NB); std::vector<float> sk(NB);
            CHECK(hipMemcpy(t.data(), t_out, NB * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(sk.data(), sink, NB * 4, hipMemcpyDeviceToHost));
            double mean = 0; for (auto v : t) mean += (double)v; mean /= NB;
            bool same = true; for (int b = 1; b < NB; ++b) same &= sk[b] == sk[0];
            if (rep == 2)
                printf("{\"blocks\": %d, \"variant\": \"library\", \"us_per_generation\": %.3f, \"of_which_work\": %.2f, \"sum_cost_us\": %.3f, \"all_blocks_same_total\": %s, \"total\": %.6f}\n",
                       NB, mean * 0.01 / GENS, (work_ticks + 7.5) * 0.01, mean * 0.01 / GENS - (work_ticks + 7.5) * 0.01, same ? "true" : "false", sk[0]);
        }
    }
    run("granules", (const void*)k_granules);
    run("atomics_1", (const void*)k_atomics<1>);
    run("atomics_8", (const void*)k_atomics<8>);
    run("atomics_32", (const void*)k_atomics<32>);
    return 0;
}
