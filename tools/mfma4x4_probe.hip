// Which lane supplies / receives what in v_mfma_f32_4x4x1_16b_f32 (16 blocks of D[4][4] += A[4][1] B[1][4]) on gfx950: a probe run
// once before kernels_lighting.hip's matrix-pipe variant of the lighting sweep was written (hipcc --offload-arch=gfx950 -o mfma4x4_probe.bin).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, int mode) {
    const int l = threadIdx.x;
    const float a = mode == 0 ? (float)(l + 1) : 1.f, b = mode == 0 ? 1.f : (float)(l + 1);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 256 * sizeof(float));
    float h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (%s = lane + 1, the other operand 1): D[lane][reg] = the supplying lane + 1\n", mode, mode == 0 ? "A" : "B");
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int r = 0; r < 4; ++r) printf(" %3.0f", h[l * 4 + r]); printf("%s", (l & 3) == 3 ? "\n" : "   "); }
    }
    return 0;
}
