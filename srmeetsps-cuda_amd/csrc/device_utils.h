// device_utils.h -- wave-64 reductions and small vector helpers (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace srps {

// ---- wave-64 shuffle reductions ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {          // total in every lane (butterfly)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block sum for blockDim.x*blockDim.y*blockDim.z = NT threads (multiple of 64, <= 1024).
// sm must hold 16 floats. Result valid in every thread. Fixed order => deterministic.
__device__ __forceinline__ float block_sum(float v, float* sm) {
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int nw = (blockDim.x * blockDim.y * blockDim.z) >> 6;
    v = wave_sum(v);
    if (nw == 1) return v;
    __syncthreads();                      // protect sm from a previous call
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sm[i];
    return t;
}

// Sum of a partial-sum array written by a PREVIOUS kernel, by a block of exactly 256 threads
// (every kernel that calls this is launched with 256 threads): thread t adds elements t, t+256, ...
// in double, then a wave butterfly, then the four wave sums in a fixed order.  The pattern does not
// depend on anything but n, so every block of every kernel obtains the bit-identical value.
__device__ __forceinline__ double sum_partials(const float* __restrict__ part, int n, double* sm_d /* [4] */) {
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double acc = 0.0;
    for (int i = tid; i < n; i += 256) acc += (double)part[i];
    acc = wave_sum(acc);
    __syncthreads();                       // protect sm_d from a previous call
    if ((tid & 63) == 0) sm_d[tid >> 6] = acc;
    __syncthreads();
    return ((sm_d[0] + sm_d[1]) + sm_d[2]) + sm_d[3];
}

// ---- grid-wide sum of a persistent (cooperative) launch: blocks of up to 1024 threads (a multiple of 64) ----
// Grid-wide sum without read-modify-write atomics (256 device-scope atomics on one address serialise in the
// fabric: the library's grid barrier costs 33 us on 256 CUs). Every block publishes {generation, partial sum} as
// one 64-bit device-scope store; thread t of every block polls entry t until it carries this generation; then
// every block adds the same values in the same order. Nothing but these entries travels between blocks, so no
// other fences are needed. Two slots: a block can be at most one reduction ahead of the slowest one.
__device__ __forceinline__ void grid_sum_publish(float v, unsigned long long* ent, unsigned gen, float* sm) {
    unsigned long long* slot = ent + (size_t)(gen & 1u) * gridDim.x;
    const float t = block_sum(v, sm);
    if (threadIdx.x == 0)
        __hip_atomic_store(&slot[blockIdx.x], ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(t),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float grid_sum_collect(unsigned long long* ent, unsigned gen, double* smd) {
    const int nb = gridDim.x, tid = threadIdx.x;
    unsigned long long* slot = ent + (size_t)(gen & 1u) * nb;
    double a = 0.0;
    for (int i = tid; i < nb; i += (int)blockDim.x) {
        unsigned long long w;
        while ((unsigned)((w = __hip_atomic_load(&slot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != gen)
            __builtin_amdgcn_s_sleep(1);
        a += (double)__uint_as_float((unsigned)w);
    }
    a = wave_sum(a);
    __syncthreads();
    if ((tid & 63) == 0) smd[tid >> 6] = a;
    __syncthreads();
    double tot = 0.0;
    const int nw = (int)blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) tot += smd[i];
    return (float)tot;
}
__device__ __forceinline__ float grid_sum(float v, unsigned long long* ent, unsigned gen, float* sm, double* smd) {
    grid_sum_publish(v, ent, gen, sm);
    return grid_sum_collect(ent, gen, smd);
}

// p = beta p + r the way the reference's CG does it: Sscal (dc.cu:263) then Saxpy (dc.cu:264), two roundings
__device__ __forceinline__ float scal_then_axpy(float beta, float p, float r) {
#pragma clang fp contract(off)
    const float t = beta * p;
    return t + r;
}

// ---- V-wide loads of consecutive floats (V = 1, 2 or 4) ---------------------------------------
template <int V>
struct Vec {
    float v[V];
};
template <int V>
__device__ __forceinline__ Vec<V> ldv(const float* __restrict__ p);
template <>
__device__ __forceinline__ Vec<1> ldv<1>(const float* __restrict__ p) {
    Vec<1> r;
    r.v[0] = *p;
    return r;
}
template <>
__device__ __forceinline__ Vec<4> ldv<4>(const float* __restrict__ p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    Vec<4> r;
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    return r;
}
template <>
__device__ __forceinline__ Vec<2> ldv<2>(const float* __restrict__ p) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    Vec<2> r;
    r.v[0] = t.x; r.v[1] = t.y;
    return r;
}
template <int V>
__device__ __forceinline__ void stv(float* __restrict__ p, const Vec<V>& a);
template <>
__device__ __forceinline__ void stv<1>(float* __restrict__ p, const Vec<1>& a) { *p = a.v[0]; }
template <>
__device__ __forceinline__ void stv<4>(float* __restrict__ p, const Vec<4>& a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}

}  // namespace srps
