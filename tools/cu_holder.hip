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
