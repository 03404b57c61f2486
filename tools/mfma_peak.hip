// Calibration probe: sustained fp32-MFMA rate of this MI355X (v_mfma_f32_32x32x2_f32 and
// v_mfma_f32_16x16x4_f32) for 1/2/4 independent accumulators per wave and 1..4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <typename F>
double run(F launch, double flop) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return flop * 5 / (ms * 1e-3) / 1e12;
}
int main() {
    float* out; hipMalloc(&out, 4096);
    const int iters = 2000;
    for (int bpc = 1; bpc <= 4; ++bpc) {
        const int blocks = 256 * bpc;   // 256 threads = 1 wave per SIMD per block
        const double n32 = (double)blocks * 4 * iters * 8, f32 = 2.0 * 32 * 32 * 2, f16 = 2.0 * 16 * 16 * 4;
        printf("waves/SIMD %d | 32x32x2 acc1 %.1f acc2 %.1f acc4 %.1f TF | 16x16x4 acc1 %.1f acc2 %.1f acc4 %.1f TF\n", bpc,
               run([&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, n32 * 1 * f32),
               run([&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, n32 * 2 * f32),
               run([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, n32 * 4 * f32),
               run([&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, n32 * 1 * f16),
               run([&] { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, n32 * 2 * f16),
               run([&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, n32 * 4 * f16));
    }
    return 0;
}
