// DVFS probe: sustained fp32-MFMA rate with CONSTANT vs RANDOM operand registers (one wave per SIMD, 4
// accumulators, registers only).  On a power-capped part the random-operand rate is the real ceiling of a GEMM.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, int random, float scale) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    unsigned s = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u; a[i] = random ? scale * ((int)(s >> 8) / 8388608.f - 1.f) : 1.f;
        s = s * 1664525u + 1013904223u; b[i] = random ? scale * ((int)(s >> 8) / 8388608.f - 1.f) : 2.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u + 2 * i) & 7], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 12345.678f) out[threadIdx.x] = t;
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps)
        for (int mode = 0; mode < 3; ++mode) {
            const int blocks = 256 * wps;
            const float scale = mode == 2 ? 0.01f : 1.f;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, mode > 0, scale); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, mode > 0, scale);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = 5.0 * blocks * 4 * iters * 32 * (2.0 * 32 * 32 * 2);
            printf("waves/SIMD %d operands %s: %.1f TF (%.2f ms per launch)\n", wps,
                   mode == 0 ? "constant" : mode == 1 ? "random [-1,1)" : "random small", flop / (ms * 1e-3) / 1e12, ms / 5);
        }
    return 0;
}
