// cu_holder.hip -- a co-tenant that holds CUs: test aid for the bounded waits of the persistent kernels
// (tests/test_gpu_persistent_abort.py).  Launches `blocks` blocks of 64 threads that each pin `lds_kb` KiB of LDS and spin
// until `ms` milliseconds have passed (s_memrealtime), prints "holding" once the kernel is in flight, waits for it, exits.
// A CU whose LDS is partly taken cannot host a block of k_cg_resident (157 of 160 KiB), so that kernel's grid no longer
// fits on the device while this one runs.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/cu_holder.bin tools/cu_holder.hip ;  tools/cu_holder.bin 64 3000 16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_hold(unsigned long long ticks, int* out) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (lds[threadIdx.x] == -1) out[0] = 1;
}

static hipStream_t g_stream = nullptr;
static int* g_out = nullptr;

// asynchronous: the kernel runs on its own non-blocking stream until the time is up (in-process co-tenant for the tests)
extern "C" int cu_holder_launch(int blocks, int ms, int lds_kb) {
    if (!g_out && hipMalloc(&g_out, 64) != hipSuccess) return 1;
    if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 1;
    (void)hipFuncSetAttribute((const void*)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipLaunchKernelGGL(k_hold, dim3(blocks), dim3(64), (size_t)lds_kb * 1024, g_stream, (unsigned long long)ms * 100000ull, g_out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
extern "C" int cu_holder_wait(void) { return g_stream ? (int)hipStreamSynchronize(g_stream) : 0; }

// Do two streams of this process run side by side?  The runtime deals streams to a few hardware queues (four per priority level, by use
// count), and kernels of two streams that share a queue run one after the other.  A one-block kernel spins on stream a for 30 ms; a
// one-block kernel that returns at once follows on stream b: 1 if b's finished while a's was still spinning, 0 if it had to wait, -1 on
// an error.  (Test aid: ranks of one process on ONE device need streams that pass this, tests/test_gpu_strips.py.)
__global__ void k_nop(int* out) { if (out == nullptr) __builtin_trap(); }
extern "C" int cu_streams_concurrent(void* stream_a, void* stream_b) {
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    if (!g_out && hipMalloc(&g_out, 64) != hipSuccess) return -1;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) return -1;
    (void)hipFuncSetAttribute((const void*)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, 1024);
    hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 1024, a, 30ull * 100000ull, g_out);
    (void)hipEventRecord(ea, a);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, b, g_out);
    (void)hipEventRecord(eb, b);
    int side_by_side = 0;
    for (;;) {
        const hipError_t qb = hipEventQuery(eb), qa = hipEventQuery(ea);
        if (qb == hipSuccess) { side_by_side = (qa == hipErrorNotReady) ? 1 : 0; break; }
        if (qa == hipSuccess && qb != hipSuccess) {              // a is done, b not yet: b waited behind it (or is just finishing)
            side_by_side = 0; break;
        }
        if ((qa != hipErrorNotReady && qa != hipSuccess) || (qb != hipErrorNotReady && qb != hipSuccess)) { side_by_side = -1; break; }
    }
    (void)hipGetLastError();
    (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    return side_by_side;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 64;
    const int ms = argc > 2 ? atoi(argv[2]) : 2000;
    const int lds_kb = argc > 3 ? atoi(argv[3]) : 16;
    if (hipSetDevice(0) != hipSuccess || cu_holder_launch(blocks, ms, lds_kb)) { fprintf(stderr, "launch failed\n"); return 1; }
    printf("holding\n");
    fflush(stdout);
    const int e = cu_holder_wait();
    printf("released %d\n", e);
    return e == 0 ? 0 : 1;
}
