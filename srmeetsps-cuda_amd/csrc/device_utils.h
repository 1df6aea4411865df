// device_utils.h -- wave-64 reductions and small vector helpers (gfx950).
#pragma once
#ifndef SRPS_POLL_SLEEP
#define SRPS_POLL_SLEEP 1
#endif
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

// the same sum of partials that OTHER blocks of the running kernel have written (the caller has seen them arrive through an atomic
// ticket and fenced): loads at device scope, the same order, the same bits
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// One value written THROUGH to device scope (sc1) / to the host (sc0 sc1), and waited for: what follows in program order -- the atomic
// ticket, the sequence word -- is then behind it.  Instead of a release fence: at device scope that is a write-back of the whole L2
// (buffer_wbl2), and in a sweep that has just stored 64 MB of normals, issued by each of its 512 blocks, it cost 90 us per pass.
__device__ __forceinline__ void st_agent_done(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_system_done(void* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ double sum_partials_agent(const float* part, int n, double* sm_d /* [4] */) {
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double acc = 0.0;
    // eight loads in flight per lane (a load at device scope comes from memory: ~1.5 us; one after the other, the 2 + 4 rounds of the
    // energy sweep's two arrays took 9 us); entries past the end re-read the last one and are not added
    for (int base = 0; base < n; base += 8 * 256) {
        float v[8];
        const float* q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = part + min(base + j * 256 + tid, n - 1);
        asm volatile("global_load_dword %0, %8, off sc1\n\t"
                     "global_load_dword %1, %9, off sc1\n\t"
                     "global_load_dword %2, %10, off sc1\n\t"
                     "global_load_dword %3, %11, off sc1\n\t"
                     "global_load_dword %4, %12, off sc1\n\t"
                     "global_load_dword %5, %13, off sc1\n\t"
                     "global_load_dword %6, %14, off sc1\n\t"
                     "global_load_dword %7, %15, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
                     : "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (base + j * 256 + tid < n) acc += (double)v[j];      // the order of sum_partials: i = tid, tid + 256, ...
    }
    acc = wave_sum(acc);
    __syncthreads();
    if ((tid & 63) == 0) sm_d[tid >> 6] = acc;
    __syncthreads();
    return ((sm_d[0] + sm_d[1]) + sm_d[2]) + sm_d[3];
}

// four partial-sum arrays [4][stride] at once (one pair of barriers), the first n entries of each -- the ones the producing
// launch wrote: entries behind them may be left over from a launch with more blocks (another strip width) --: the same pattern
// and the same bits per array as sum_partials
__device__ __forceinline__ void sum_partials4(const float* __restrict__ part, int n, int stride, double (&out)[4], double (*sm_d)[4] /* [4][4] */) {
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = tid; i < n; i += 256) {
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] += (double)part[(size_t)v * stride + i];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = wave_sum(acc[v]);
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) sm_d[v][tid >> 6] = acc[v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 4; ++v) out[v] = ((sm_d[v][0] + sm_d[v][1]) + sm_d[v][2]) + sm_d[v][3];
}
// block sums of four values of a 256-thread block: results in sm[0..3] (sm holds 4 + 16 floats), valid after the call
__device__ __forceinline__ void block_sum4(const float (&v)[4], float* sm) {
    const int tid = threadIdx.x;
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = wave_sum(v[i]);
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sm[4 + i * 4 + (tid >> 6)] = t[i];
    }
    __syncthreads();
    if (tid < 4) sm[tid] = ((sm[4 + tid * 4] + sm[4 + tid * 4 + 1]) + sm[4 + tid * 4 + 2]) + sm[4 + tid * 4 + 3];
    __syncthreads();
}

// ---- wave totals through DPP (no LDS traffic; ds_bpermute-based shuffles cost ~150 cycles each) -------------
// Fixed order: four row_shr steps leave each row's total in its lane 15, row_bcast:15 / row_bcast:31 carry them
// to lane 63, v_readlane broadcasts.  Invalid source lanes read 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ float wave_total(float v) {
    v += dpp_f<0x111, 0xf>(v);       // row_shr:1
    v += dpp_f<0x112, 0xf>(v);       // row_shr:2
    v += dpp_f<0x114, 0xf>(v);       // row_shr:4
    v += dpp_f<0x118, 0xf>(v);       // row_shr:8
    v += dpp_f<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v += dpp_f<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_total(double v) {
    v += dpp_d<0x111, 0xf>(v);
    v += dpp_d<0x112, 0xf>(v);
    v += dpp_d<0x114, 0xf>(v);
    v += dpp_d<0x118, 0xf>(v);
    v += dpp_d<0x142, 0xa>(v);
    v += dpp_d<0x143, 0xc>(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

// ---- bounded waits of the persistent kernels ---------------------------------------------------------------
// A kernel whose blocks wait for each other hangs forever when one of its blocks never becomes resident (a co-tenant
// process holds the CU, a CU mask, the occupancy API admitting one block more than the hardware).  Every wait below is
// therefore bounded: the kernel computes ONE deadline at its start (s_memrealtime, 100 MHz, plus the budget the host
// passes: option "spin_budget_ms"), a waiting wave looks at the clock every 64th failed poll, and once the deadline has
// passed the wait gives up: it marks the block dead (an LDS flag every thread reads at the end of the kernel: a dead block
// stores no results), records in the status words which wait failed and how many blocks had arrived, and returns NaN --
// the CG loops end on a NaN r.r by themselves.  A dead block publishes nothing further, so its neighbours time out in
// turn and the whole launch ends within about two budgets.  The host finds the abort flag in the report record and
// repeats the solve with the streaming kernels (srps_api.hip: persistent_recover).
// The deadline, the status pointer and the dead flag live in LDS (a fixed address: no registers are held for them in
// kernels that have none to spare); only failed polls ever look at them.
struct SpinState {
    unsigned long long deadline;   // s_memrealtime value after which a failed poll gives up
    int* status;                   // [3] abort flags | blocks that had arrived at the failing wait | its generation
    int bit;                       // this kernel's bit in the abort flags (1 depth CG, 2 albedo CG)
    int dead;                      // some wait of this block gave up
};
__device__ __forceinline__ SpinState* spin_state() {
    __shared__ SpinState st;
    return &st;
}
// called by every thread at the start of a persistent kernel; a barrier must follow before the first wait
__device__ __forceinline__ void spin_guard_init(unsigned long long budget_ticks, int* status, int bit) {
    if (threadIdx.x == 0) {
        SpinState* st = spin_state();
        st->deadline = __builtin_amdgcn_s_memrealtime() + budget_ticks;
        st->status = status; st->bit = bit; st->dead = 0;
    }
}
__device__ __forceinline__ bool spin_deadline_passed() { return __builtin_amdgcn_s_memrealtime() > spin_state()->deadline; }
__device__ __forceinline__ bool spin_expired(unsigned& polls) {
    if ((++polls & 63u) != 0u) return false;
    return spin_deadline_passed();
}
__device__ __forceinline__ void spin_give_up(int arrived, unsigned gen) {
    SpinState* st = spin_state();
    st->dead = 1;
    int* status = st->status;
    if (__hip_atomic_fetch_or(status, st->bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) { status[1] = arrived; status[2] = (int)gen; }
}
// after the last barrier-separated wait of the kernel: true when some wait of this block gave up (block-uniform)
__device__ __forceinline__ bool spin_block_dead() {
    __syncthreads();
    return spin_state()->dead != 0;
}

// Several launches as ONE grid (round 4: the resident depth CG on the column strips of a multi-GPU partition, kernels_resident.hip): the
// granule arrays of the exchange exist once per launch ("rank"), every block publishes its granule into ITS slot of every rank's
// array -- its own and, through peer pointers, the others' -- and polls its own rank's array only.  nb slots for the blocks of the
// whole group, in the order a single launch over all tiles would give them: the sums are added in that order, bit for bit as there.
// Stores and polls are system scope then (the arrays may sit in another device's memory).  A null pointer where the functions below take
// a GridPeers means: one launch, slots = blocks, device scope -- and folds away at compile time.
struct GridPeers {
    int nb;                              // granule slots = blocks of the whole group
    int slot;                            // this block's slot
    int n_peers;                         // the OTHER ranks
    unsigned long long* ent[7];          // their 8-byte granule arrays [2][nb]
    unsigned long long* ent3[7];         // their 16-byte granule arrays [2][nb rounded up to 256]
};

// ---- grid-wide sum of a persistent (cooperative) launch: blocks of up to 1024 threads (a multiple of 64) ----
// No read-modify-write atomics (256 device-scope atomics on one address serialise in the fabric: the library's grid
// barrier costs 33 us on 256 CUs).  Every block publishes {generation, partial sum} as one 64-bit device-scope store;
// the first wave of every block then collects all the entries (lane l polls entries 4l..4l+3 until they carry this
// generation; more polling waves cost fabric bandwidth: 8 per block made the step 15 % slower) and adds them in a fixed
// order, in double: all blocks obtain the same bits.  Nothing but these entries travels between blocks, so no other fences are needed.  Two slots per
// block: a block can be at most one reduction ahead of the slowest one.  sm: 40 floats (blocks of at most 16 waves).
__device__ __forceinline__ void grid_sum_publish(float v, unsigned long long* ent, unsigned gen, float* sm, const GridPeers* gp = nullptr) {
    const int tid = threadIdx.x, nw = (int)blockDim.x >> 6;
    float* s = sm + (gen & 1u) * 16;                       // alternating: the previous reduction may still be read
    const float t = wave_total(v);
    if ((tid & 63) == 0) s[tid >> 6] = t;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int i = 0; i < nw; ++i) tot += s[i];
        const unsigned long long g = ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(tot);
        if (gp) {
            const size_t at = (size_t)(gen & 1u) * gp->nb + gp->slot;
            __hip_atomic_store(&ent[at], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int q = 0; q < gp->n_peers; ++q) __hip_atomic_store(&gp->ent[q][at], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else
            __hip_atomic_store(&ent[(size_t)(gen & 1u) * gridDim.x + blockIdx.x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ float grid_sum_collect(unsigned long long* ent, unsigned gen, float* sm, const GridPeers* gp = nullptr) {
    const int nb = gp ? gp->nb : (int)gridDim.x, tid = threadIdx.x, lane = tid & 63;
    const unsigned long long* slot = ent + (size_t)(gen & 1u) * nb;
    float* res = sm + 32 + (gen & 1u);                     // behind the two sets of wave partials
    if (tid < 64) {                                        // one polling wave per block: pollers cost fabric bandwidth
        double a = 0.0;
        unsigned polls = 0;
        bool gave_up = false;
        for (int base = 0; base < nb; base += 256) {
            unsigned long long w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = base + 4 * lane + i;
                w[i] = (idx < nb) ? (gp ? __hip_atomic_load(&slot[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(&slot[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                  : ((unsigned long long)gen << 32);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = base + 4 * lane + i;
                while (!gave_up && (unsigned)(w[i] >> 32) != gen) {
                    if (spin_expired(polls)) { gave_up = true; break; }
                    __builtin_amdgcn_s_sleep(SRPS_POLL_SLEEP);
                    w[i] = gp ? __hip_atomic_load(&slot[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(&slot[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                a += (double)__uint_as_float((unsigned)w[i]);
            }
        }
        float tot = (float)wave_total(a);
        const unsigned long long missing = __builtin_amdgcn_ballot_w64(gave_up);
        if (missing != 0ull) {                             // wave-uniform: some lane's granule never came
            tot = __builtin_nanf("");
            if (lane == 0) spin_give_up(nb - (int)__builtin_popcountll(missing), gen);
        }
        if (lane == 0) *res = tot;
    }
    __syncthreads();
    return *res;
}
__device__ __forceinline__ float grid_sum(float v, unsigned long long* ent, unsigned gen, float* sm, const GridPeers* gp = nullptr) {
    grid_sum_publish(v, ent, gen, sm, gp);
    return grid_sum_collect(ent, gen, sm, gp);
}

// Three sums in one exchange. A block publishes ONE 16-byte granule {generation, v0, v1, v2} (partial sums float per
// thread and per wave, double across the waves, rounded to float), written by a single global_store_dwordx4 and read by a single
// global_load_dwordx4 (agent scope, sc1), so the three values and their tag arrive together: the collecting wave needs one
// round trip when all blocks have published -- with three 8-byte granules per block it needed three.
// ent3: [2][blocks rounded up to 256] granules SRPS_G3_STRIDE bytes apart, 16-byte aligned, zeroed before the launch.
typedef unsigned srps_v4u __attribute__((ext_vector_type(4)));
#ifndef SRPS_G3_STRIDE
#define SRPS_G3_STRIDE 256       // bytes between the granules of consecutive blocks: every block polls all of them at the same
                                 // time, and packed into 4 KB they queue on a few memory channels (16: 11.75, 64: 11.5, 256 and
                                 // 1024: 11.45 us per CG step at 2048 x 2048)
#endif
// Round 5, the collect's round trip (every block's polling wave asks for all 256 granules at the same moment: 65 536 16-byte requests
// for 256 lines per step).  SRPS_G3_REPLICAS = R: every block stores its granule R times, a block polls copy (slot mod R) -- the same
// requests over R times the lines.  Same-box A/B at 2048 x 2048 (tools/ab_variants.sh, three rounds): R = 1 7.95 - 7.99 us per CG step,
// R = 2 7.81 - 7.84, R = 8 7.79 - 7.84: two copies are the default (one more 16-byte store per block and step).  SRPS_G3_GROUP = 4 (four
// consecutive granules in one 64-byte segment, so that the four lanes that read them in one instruction are ONE request, a quarter of
// the requests on the same footprint) measured 8.03 - 8.09: off.
#ifndef SRPS_G3_GROUP
#define SRPS_G3_GROUP 1
#endif
#ifndef SRPS_G3_REPLICAS
#define SRPS_G3_REPLICAS 2
#endif
// byte offset of granule g (of nbr = blocks rounded up to 256) in copy `rep` of generation parity `par`
__device__ __forceinline__ size_t g3_offset(unsigned par, int rep, int nbr, int g) {
    const size_t copy = ((size_t)par * SRPS_G3_REPLICAS + rep) * (size_t)nbr * SRPS_G3_STRIDE;
    if (SRPS_G3_GROUP > 1) return copy + (size_t)(g / SRPS_G3_GROUP) * (SRPS_G3_STRIDE * SRPS_G3_GROUP) + (size_t)(g % SRPS_G3_GROUP) * 16;
    return copy + (size_t)g * SRPS_G3_STRIDE;
}
// FLOAT32: the sums across the waves of a block and across the blocks in fp32 instead of fp64 -- for the depth CG, whose
// critical path after the last block has published runs through this arithmetic (one thread's 24 dependent fp64 additions,
// then the polling wave's fp64 DPP totals: ~0.25 us per CG step).  The per-thread and per-wave sums are fp32 either way; the
// order is fixed, all blocks obtain the same bits.
// arrival counters of the barrier-free publish below (one per generation parity)
__device__ __forceinline__ unsigned* grid_sum3_arrived() {
    __shared__ unsigned arrived[2];
    return arrived;
}
// called by every thread of a kernel that uses grid_sum3_publish<NW, true>, before its first barrier
__device__ __forceinline__ void grid_sum3_prepare() {
    if (threadIdx.x < 2) grid_sum3_arrived()[threadIdx.x] = 0u;
}
template <int NW = 0, bool FLOAT32 = false>      // NW: waves per block when known at compile time (the cross-wave sum is then one LDS round trip)
__device__ __forceinline__ void grid_sum3_publish(float v0, float v1, float v2, unsigned long long* ent3, unsigned gen, unsigned long long* st = nullptr,
                                                  const GridPeers* gp = nullptr) {
    const int tid = threadIdx.x, nw = NW ? NW : (int)blockDim.x >> 6, nb = gp ? gp->nb : (int)gridDim.x, myslot = gp ? gp->slot : (int)blockIdx.x;
    // the granule into this block's slot of the rank's own array and, in a group, of every other rank's
    auto store_granule = [&](const srps_v4u& g) {
        const int nbr = (nb + 255) & ~255;
#pragma unroll
        for (int rep = 0; rep < SRPS_G3_REPLICAS; ++rep) {
        const size_t at = g3_offset(gen & 1u, rep, nbr, myslot);
        const char* dst = reinterpret_cast<const char*>(ent3) + at;
        // s_nop 1: a store of more than 8 bytes must not be followed within two wait states by a write of its data registers
        // (gfx940 and later; the compiler inserts them for its own stores, it cannot see into the asm)
        if (gp) {
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(dst), "v"(g) : "memory");
            for (int q = 0; q < gp->n_peers; ++q) {
                const char* d2 = reinterpret_cast<const char*>(gp->ent3[q]) + at;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(d2), "v"(g) : "memory");
            }
        } else
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(g) : "memory");
        }
    };
    if constexpr (NW > 0 && FLOAT32) {
        // Every wave leaves its three totals in LDS and counts itself in; the wave that arrives LAST adds the NW entries in
        // their fixed order and publishes the granule -- the block's sums leave as soon as its slowest wave is done, without
        // the wake-up of a summing thread behind a barrier.  (One wave's LDS accesses execute in order and the LDS is
        // coherent within the CU; the buffers alternate with the generation, and a grid-wide collect -- with its barrier --
        // lies between two uses of the same one.)
        __shared__ float sf[2][NW][4];
        unsigned* arrived = grid_sum3_arrived();            // zeroed by grid_sum3_prepare() before the kernel's first barrier
        const float t0 = wave_total(v0), t1 = wave_total(v1), t2 = wave_total(v2);
        if ((tid & 63) == 0) {
            float* d = sf[gen & 1u][tid >> 6];
            d[0] = t0; d[1] = t1; d[2] = t2;
            // relaxed, between two compiler barriers: an acquire-release here also waits for the edge granules just stored to
            // global memory (s_waitcnt vmcnt(0)), which is what this form is meant to avoid; the LDS itself keeps the order
            asm volatile("" ::: "memory");
            const unsigned before = __hip_atomic_fetch_add(&arrived[gen & 1u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            if (before % NW == NW - 1) {
                float f[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NW; ++i) { const float* e = sf[gen & 1u][i]; f[0] += e[0]; f[1] += e[1]; f[2] += e[2]; }
                const srps_v4u g = {gen, __float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2])};
                store_granule(g);
                if (st) st[0] = __builtin_amdgcn_s_memrealtime();
            }
        }
        // The barrier stays -- without it the block's polling wave starts to poll while its other waves still compute, takes
        // issue slots from the wave it shares a SIMD with and loads the fabric for longer (measured: +2.5 us per CG step) --
        // but it now opens the moment the last wave has published, with no summing thread to wake up behind it.
        __syncthreads();
        return;
    }
    __shared__ double sd[2][16][4];                        // per wave: three totals (+ pad)
    const double t0 = (double)wave_total(v0), t1 = (double)wave_total(v1), t2 = (double)wave_total(v2);
    if ((tid & 63) == 0) { double* d = sd[gen & 1u][tid >> 6]; d[0] = t0; d[1] = t1; d[2] = t2; }
    __syncthreads();
    if (tid == 0) {
        double tot[3] = {0.0, 0.0, 0.0};
        if (NW && FLOAT32) {
            float f[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < (NW ? NW : 1); ++i) { const double* d = sd[gen & 1u][i]; f[0] += (float)d[0]; f[1] += (float)d[1]; f[2] += (float)d[2]; }
            tot[0] = f[0]; tot[1] = f[1]; tot[2] = f[2];
        } else if (NW) {
#pragma unroll
            for (int i = 0; i < (NW ? NW : 1); ++i) { const double* d = sd[gen & 1u][i]; tot[0] += d[0]; tot[1] += d[1]; tot[2] += d[2]; }
        } else {
            for (int i = 0; i < nw; ++i) { const double* d = sd[gen & 1u][i]; tot[0] += d[0]; tot[1] += d[1]; tot[2] += d[2]; }
        }
        const srps_v4u g = {gen, __float_as_uint((float)tot[0]), __float_as_uint((float)tot[1]), __float_as_uint((float)tot[2])};
        store_granule(g);
        if (st) st[0] = __builtin_amdgcn_s_memrealtime();
    }
}
template <bool FLOAT32 = false>
__device__ __forceinline__ void grid_sum3_collect(unsigned long long* ent3, unsigned gen, double& o0, double& o1, double& o2, unsigned long long* st = nullptr,
                                                  const GridPeers* gp = nullptr) {
    const int nb = gp ? gp->nb : (int)gridDim.x, tid = threadIdx.x, lane = tid & 63;
    __shared__ double res3[2][3];
    if (tid < 64) {                                        // one polling wave per block
        double acc[3] = {0.0, 0.0, 0.0};
        float facc[3] = {0.f, 0.f, 0.f};
        const int nbr = (nb + 255) & ~255;
        const int myrep = SRPS_G3_REPLICAS > 1 ? (int)((gp ? gp->slot : (int)blockIdx.x) % SRPS_G3_REPLICAS) : 0;
        const char* slot = reinterpret_cast<const char*>(ent3);
        for (int base = 0; base < nb; base += 256) {
            // lane l takes granules l, l + 64, l + 128, l + 192 of this group, all requested before the first is looked at
            const char* src = slot + g3_offset(gen & 1u, myrep, nbr, base + lane);
            const char *s1 = slot + g3_offset(gen & 1u, myrep, nbr, base + lane + 64), *s2 = slot + g3_offset(gen & 1u, myrep, nbr, base + lane + 128),
                       *s3 = slot + g3_offset(gen & 1u, myrep, nbr, base + lane + 192);
            srps_v4u w[4];
            // every round asks for all four again: a lane whose granules arrive in a different order than it looks at them
            // would otherwise pay one more round trip per granule
            const unsigned want0 = (base + lane < nb) ? gen : 0u, want1 = (base + lane + 64 < nb) ? gen : 0u;
            const unsigned want2 = (base + lane + 128 < nb) ? gen : 0u, want3 = (base + lane + 192 < nb) ? gen : 0u;
            bool first = true;
            for (;;) {
                if (gp)
                    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
                                 "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
                                 "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
                                 "s_waitcnt vmcnt(0)"
                                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(src), "v"(s1), "v"(s2), "v"(s3) : "memory");
                else
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                             "global_load_dwordx4 %1, %5, off sc1\n\t"
                             "global_load_dwordx4 %2, %6, off sc1\n\t"
                             "global_load_dwordx4 %3, %7, off sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(src), "v"(s1), "v"(s2), "v"(s3) : "memory");
                if (first && st && tid == 0) st[1] = __builtin_amdgcn_s_memrealtime();
                first = false;
                // out-of-range granules are never written (tag 0): they are not waited for
                const bool ok = (want0 == 0u || w[0].x == gen) && (want1 == 0u || w[1].x == gen) && (want2 == 0u || w[2].x == gen) &&
                                (want3 == 0u || w[3].x == gen);
                if (ok) break;
                if (spin_deadline_passed()) {          // the clock is read by the scalar unit; only lanes still waiting get here
                    const int n_missing = (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(want0 != 0u && w[0].x != gen)) +
                                          (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(want1 != 0u && w[1].x != gen)) +
                                          (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(want2 != 0u && w[2].x != gen)) +
                                          (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(want3 != 0u && w[3].x != gen));
                    spin_give_up(nb - n_missing, gen);
                    break;
                }
                __builtin_amdgcn_s_sleep(SRPS_POLL_SLEEP);
            }
            if (FLOAT32) {
                if (want0) { facc[0] += __uint_as_float(w[0].y); facc[1] += __uint_as_float(w[0].z); facc[2] += __uint_as_float(w[0].w); }
                if (want1) { facc[0] += __uint_as_float(w[1].y); facc[1] += __uint_as_float(w[1].z); facc[2] += __uint_as_float(w[1].w); }
                if (want2) { facc[0] += __uint_as_float(w[2].y); facc[1] += __uint_as_float(w[2].z); facc[2] += __uint_as_float(w[2].w); }
                if (want3) { facc[0] += __uint_as_float(w[3].y); facc[1] += __uint_as_float(w[3].z); facc[2] += __uint_as_float(w[3].w); }
            } else {
            if (want0) { acc[0] += (double)__uint_as_float(w[0].y); acc[1] += (double)__uint_as_float(w[0].z); acc[2] += (double)__uint_as_float(w[0].w); }
            if (want1) { acc[0] += (double)__uint_as_float(w[1].y); acc[1] += (double)__uint_as_float(w[1].z); acc[2] += (double)__uint_as_float(w[1].w); }
            if (want2) { acc[0] += (double)__uint_as_float(w[2].y); acc[1] += (double)__uint_as_float(w[2].z); acc[2] += (double)__uint_as_float(w[2].w); }
            if (want3) { acc[0] += (double)__uint_as_float(w[3].y); acc[1] += (double)__uint_as_float(w[3].z); acc[2] += (double)__uint_as_float(w[3].w); }
            }
        }
        if (st && tid == 0) st[2] = __builtin_amdgcn_s_memrealtime();
        const double t0 = FLOAT32 ? (double)wave_total(facc[0]) : wave_total(acc[0]), t1 = FLOAT32 ? (double)wave_total(facc[1]) : wave_total(acc[1]),
                     t2 = FLOAT32 ? (double)wave_total(facc[2]) : wave_total(acc[2]);
        if (lane == 0) { res3[gen & 1u][0] = t0; res3[gen & 1u][1] = t1; res3[gen & 1u][2] = t2; }
    }
    __syncthreads();
    o0 = res3[gen & 1u][0]; o1 = res3[gen & 1u][1]; o2 = res3[gen & 1u][2];
    if (spin_state()->dead) o0 = o1 = o2 = (double)__builtin_nanf("");      // a wait gave up: the CG loops end on a NaN
}

// The same exchange for three independent triples at once (the three colour channels of the albedo CG advance in lockstep):
// one granule per block and triple, ent9: [2][3][blocks rounded up to 256] granules SRPS_G3_STRIDE apart; all twelve
// granules of a lane are requested before the first is looked at, so the collect is still one round trip.  Per triple the
// arithmetic (float per thread and wave, double across waves and blocks, fixed order) is that of grid_sum3.
template <int NW>
__device__ __forceinline__ void grid_sum9_publish(const float (&v)[3][3], unsigned long long* ent9, unsigned gen) {
    const int tid = threadIdx.x, nb = gridDim.x;
    __shared__ double sd9[2][NW][3][4];
    double t[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) t[c][k] = (double)wave_total(v[c][k]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) sd9[gen & 1u][tid >> 6][c][k] = t[c][k];
    }
    __syncthreads();
    if (tid < 3) {                                         // thread c publishes the granule of triple c
        double tot[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NW; ++i) { const double* d = sd9[gen & 1u][i][tid]; tot[0] += d[0]; tot[1] += d[1]; tot[2] += d[2]; }
        const srps_v4u g = {gen, __float_as_uint((float)tot[0]), __float_as_uint((float)tot[1]), __float_as_uint((float)tot[2])};
        const int nbr = (nb + 255) & ~255;
        const char* dst = reinterpret_cast<const char*>(ent9) + (((size_t)(gen & 1u) * 3 + tid) * nbr + blockIdx.x) * SRPS_G3_STRIDE;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(g) : "memory");
    }
}
__device__ __forceinline__ void grid_sum9_collect(unsigned long long* ent9, unsigned gen, double (&o)[3][3]) {
    const int nb = gridDim.x, tid = threadIdx.x, lane = tid & 63;
    __shared__ double res9[2][3][4];
    if (tid < 64) {
        double acc[3][3] = {};
        unsigned polls = 0;
        int missing = 0;
        const int nbr = (nb + 255) & ~255;
        const char* slot = reinterpret_cast<const char*>(ent9) + (size_t)(gen & 1u) * 3 * nbr * SRPS_G3_STRIDE;
        const size_t cstep = (size_t)nbr * SRPS_G3_STRIDE;
        for (int base = 0; base < nb; base += 256) {
            const char* a0 = slot + (size_t)(base + lane) * SRPS_G3_STRIDE;
            const char *a1 = a0 + 64 * SRPS_G3_STRIDE, *a2 = a0 + 128 * SRPS_G3_STRIDE, *a3 = a0 + 192 * SRPS_G3_STRIDE;
            const bool want[4] = {base + lane < nb, base + lane + 64 < nb, base + lane + 128 < nb, base + lane + 192 < nb};
            srps_v4u w[3][4];
            for (;;) {
                asm volatile("global_load_dwordx4 %0, %12, off sc1\n\t"
                             "global_load_dwordx4 %1, %13, off sc1\n\t"
                             "global_load_dwordx4 %2, %14, off sc1\n\t"
                             "global_load_dwordx4 %3, %15, off sc1\n\t"
                             "global_load_dwordx4 %4, %16, off sc1\n\t"
                             "global_load_dwordx4 %5, %17, off sc1\n\t"
                             "global_load_dwordx4 %6, %18, off sc1\n\t"
                             "global_load_dwordx4 %7, %19, off sc1\n\t"
                             "global_load_dwordx4 %8, %20, off sc1\n\t"
                             "global_load_dwordx4 %9, %21, off sc1\n\t"
                             "global_load_dwordx4 %10, %22, off sc1\n\t"
                             "global_load_dwordx4 %11, %23, off sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(w[0][0]), "=&v"(w[0][1]), "=&v"(w[0][2]), "=&v"(w[0][3]), "=&v"(w[1][0]), "=&v"(w[1][1]), "=&v"(w[1][2]),
                               "=&v"(w[1][3]), "=&v"(w[2][0]), "=&v"(w[2][1]), "=&v"(w[2][2]), "=&v"(w[2][3])
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a0 + cstep), "v"(a1 + cstep), "v"(a2 + cstep), "v"(a3 + cstep),
                               "v"(a0 + 2 * cstep), "v"(a1 + 2 * cstep), "v"(a2 + 2 * cstep), "v"(a3 + 2 * cstep)
                             : "memory");
                bool ok = true;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ok = ok && (!want[i] || w[c][i].x == gen);
                if (ok) break;
                if (spin_expired(polls)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) missing += (int)(want[i] && (w[0][i].x != gen || w[1][i].x != gen || w[2][i].x != gen));
                    break;
                }
                __builtin_amdgcn_s_sleep(SRPS_POLL_SLEEP);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (want[i]) {
                        acc[c][0] += (double)__uint_as_float(w[c][i].y);
                        acc[c][1] += (double)__uint_as_float(w[c][i].z);
                        acc[c][2] += (double)__uint_as_float(w[c][i].w);
                    }
        }
        const bool gave_up = __builtin_amdgcn_ballot_w64(missing != 0) != 0ull;      // wave-uniform
        if (gave_up) {
            const int n_missing = (int)wave_total((float)missing);
            if (lane == 0) spin_give_up(nb - n_missing, gen);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double t = gave_up ? (double)__builtin_nanf("") : wave_total(acc[c][k]);
                if (lane == 0) res9[gen & 1u][c][k] = t;
            }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) o[c][k] = res9[gen & 1u][c][k];
}

// p = beta p + r the way the reference's CG does it: Sscal (dc.cu:263) then Saxpy (dc.cu:264), two roundings
__device__ __forceinline__ float scal_then_axpy(float beta, float p, float r) {
#pragma clang fp contract(off)
    const float t = beta * p;
    return t + r;
}

// ---- perspective normal (devicecalls.cu:171-223) -----------------------------------------------
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
// V consecutive compact pixels onto a grid plane through the compact -> grid map: the V grid indices of a lane are consecutive and
// 16-byte aligned wherever the lane's pixels lie in one column segment (everywhere in a full-frame mask, nearly everywhere in
// any other) -- then ONE 16-byte store; else V scalar stores.  (One dword store per pixel makes every store instruction of a
// wave write 64 x 4 bytes with a 16-byte stride: a quarter of the bytes per request.)
template <int V>
struct GridIdx {
    int go[V];
    bool vec;
};
template <int V>
__device__ __forceinline__ GridIdx<V> grid_idx(const int* __restrict__ gofp, int q) {
    GridIdx<V> r;
    if constexpr (V == 4) {
        const int4 t = *reinterpret_cast<const int4*>(gofp + q);
        r.go[0] = t.x; r.go[1] = t.y; r.go[2] = t.z; r.go[3] = t.w;
        r.vec = ((t.x & 3) == 0) && (t.y == t.x + 1) && (t.z == t.x + 2) && (t.w == t.x + 3);
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) r.go[e] = gofp[q + e];
        r.vec = false;
    }
    return r;
}
template <int V>
__device__ __forceinline__ void scatter_store(float* __restrict__ plane, const GridIdx<V>& gi, const float (&val)[V]) {
    if constexpr (V == 4) {
        if (gi.vec) { *reinterpret_cast<float4*>(plane + gi.go[0]) = make_float4(val[0], val[1], val[2], val[3]); return; }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) plane[gi.go[e]] = val[e];
}

// Four image samples kept as bytes (the 8-bit image store, srps_api.hip: image_store_prepare): the float the reference's
// loader forms from a byte is k / 255.f (Utilities.cpp:343).  The division is replaced by k R_hi + k R_lo with R_hi + R_lo =
// 1/255 to 48 bits, the second product rounded, the sum formed by one fused multiply-add: the correctly rounded quotient for
// each of the 256 values (tests/test_gpu_image_store.py: exact rational arithmetic on the CPU, bit comparison on the GPU).
__device__ __forceinline__ float unit_from_byte(float k) {
    constexpr float R_HI = 1.f / 255.f;
    constexpr float R_LO = (float)(1.0 / 255.0 - (double)R_HI);
    return __builtin_fmaf(k, R_HI, k * R_LO);
}
__device__ __forceinline__ Vec<4> ld_img8(const unsigned char* __restrict__ p) {
    const unsigned w = *reinterpret_cast<const unsigned*>(p);
    Vec<4> r;
    r.v[0] = unit_from_byte((float)(w & 0xffu));            // v_cvt_f32_ubyte0 .. 3
    r.v[1] = unit_from_byte((float)((w >> 8) & 0xffu));
    r.v[2] = unit_from_byte((float)((w >> 16) & 0xffu));
    r.v[3] = unit_from_byte((float)(w >> 24));
    return r;
}
// the image samples of one (image, channel) row at pixel q: floats, or bytes when the context holds the 8-bit store
// Cache policy of the streams (round 4).  tools/hbm_ceiling_bench.hip on the same box: a grid-stride float4 read reaches 5.3 - 6.4 TB/s
// with the default policy and 6.5 - 7.15 with NON-TEMPORAL loads (`global_load_dwordx4 ... nt`: the lines are not kept in the L2 /
// Infinity Cache, which a stream larger than the caches only thrashes); a copy 4.2 - 5.7 against 6.5 - 7.1.  The images (1 GB per
// sweep at the metric's configuration, every byte read once per sweep) are such a stream: SRPS_NT_IMAGES.  Planes that one kernel
// writes for the next (num, den, the image sums; N and dz of the next pass) are stored non-temporally under SRPS_NT_STORES.
#ifndef SRPS_NT_IMAGES
#define SRPS_NT_IMAGES 1
#endif
#ifndef SRPS_NT_STORES
#define SRPS_NT_STORES 0
#endif
#ifndef SRPS_NT_NORMALS
#define SRPS_NT_NORMALS 1
#endif
typedef float srps_vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Vec<4> ldv4_nt(const float* __restrict__ p) {
    const srps_vf4 t = __builtin_nontemporal_load(reinterpret_cast<const srps_vf4*>(p));
    Vec<4> r;
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    return r;
}
// a load of image samples (read once per sweep, never again before the caches have turned over)
template <int V>
__device__ __forceinline__ Vec<V> ldv_stream(const float* __restrict__ p) {
    if constexpr (V == 4 && SRPS_NT_IMAGES) return ldv4_nt(p);
    else return ldv<V>(p);
}
template <int V, bool U8>
__device__ __forceinline__ Vec<V> ld_img(const float* __restrict__ I, const unsigned char* __restrict__ I8, size_t row, int P, int q, int rows = 0) {
    if constexpr (U8) {
        static_assert(V == 4, "the 8-bit image store is read four pixels at a time");
        if constexpr (SRPS_NT_IMAGES) {
            const unsigned w = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(I8 + row * (size_t)P + q));
            Vec<4> r;
            r.v[0] = unit_from_byte((float)(w & 0xffu)); r.v[1] = unit_from_byte((float)((w >> 8) & 0xffu));
            r.v[2] = unit_from_byte((float)((w >> 16) & 0xffu)); r.v[3] = unit_from_byte((float)(w >> 24));
            return r;
        } else
            return ld_img8(I8 + row * (size_t)P + q);
    } else {
        return ldv_stream<V>(I + row * (size_t)P + q);
    }
}
// the same bytes in two steps -- the raw load, and the conversion where the samples are used: a software pipeline keeps the dword in
// flight, not the four floats (whose conversion would wait for the load where it is issued)
__device__ __forceinline__ unsigned ld_bytes4_stream(const unsigned char* __restrict__ p) {
    if constexpr (SRPS_NT_IMAGES) return __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p));
    else return *reinterpret_cast<const unsigned*>(p);
}
__device__ __forceinline__ Vec<4> bytes4_to_unit(unsigned w) {
    Vec<4> r;
    r.v[0] = unit_from_byte((float)(w & 0xffu)); r.v[1] = unit_from_byte((float)((w >> 8) & 0xffu));
    r.v[2] = unit_from_byte((float)((w >> 16) & 0xffu)); r.v[3] = unit_from_byte((float)(w >> 24));
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
// a store of a plane that the writing kernel does not read again.  Same-box A/B at 2048 x 2048, 20 images (gpurun_out/r04b): the normals
// and dz the lighting sweep leaves for the next pass stored non-temporally: 0.284 -> 0.267 ms for that sweep (NT = SRPS_NT_NORMALS);
// num / den / the image sums of the albedo sweep, which the next two kernels read back at once: 0.234 -> 0.253 ms (SRPS_NT_STORES, off).
template <int V, bool NT = SRPS_NT_STORES>
__device__ __forceinline__ void stv_stream(float* __restrict__ p, const Vec<V>& a) {
    if constexpr (V == 4 && NT) {
        srps_vf4 t; t.x = a.v[0]; t.y = a.v[1]; t.z = a.v[2]; t.w = a.v[3];
        __builtin_nontemporal_store(t, reinterpret_cast<srps_vf4*>(p));
    } else stv<V>(p, a);
}

}  // namespace srps
