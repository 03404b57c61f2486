// How many VALU "filler" instructions of the bf16x3 split hide under one v_mfma_f32_32x32x16_bf16 — within ONE wave,
// and with two such waves per SIMD?  (Design input for the software-pipelined bf16x3 GEMM, round 4.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_fill.hip -o tools/mfma_fill
// Each wave runs ITER x { 4 MFMAs on 4 independent accumulators, after each MFMA NF fillers } and reports
// shader cycles per MFMA (s_memtime around the loop, wave 0 of block 0).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <int NF, int KIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.f + i * 0.01f); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.37f + i;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 77u + i;
    f32x2 xp[4];
    for (int i = 0; i < 4; ++i) { xp[i].x = threadIdx.x * 0.11f + i; xp[i].y = i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // everything in asm volatile: program order IS issue order (fillers never touch the accumulators)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int q = (j * NF + f) & 7;
                if (KIND == 0) {                      // the split's mix: cvt_pk, shift, and, sub
                    const int m = (j * NF + f) & 3;
                    if (m == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[q]) : "v"(x[q]), "v"(x[(q + 1) & 7]));
                    else if (m == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[q]) : "v"(u[(q + 3) & 7]));
                    else if (m == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[q]) : "v"(u[(q + 5) & 7]));
                    else asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[q]) : "v"(x[q]), "v"(u[(q + 2) & 7]));
                } else if (KIND == 1) {
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[q]) : "v"(x[q]), "v"(x[(q + 1) & 7]), "v"(x[(q + 2) & 7]));
                } else if (KIND == 2) {                // v_pk_add_f32 on register pairs
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(xp[q & 3]) : "v"(xp[q & 3]), "v"(xp[(q + 1) & 3]));
                } else {                               // the fp16x2 split's mix: v_mul, v_cvt_pk_f16_f32, v_fma_mix_f32
                    const int m = (j * NF + f) % 3;
                    if (m == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[q]) : "v"(x[(q + 3) & 7]), "v"(x[(q + 5) & 7]));
                    else if (m == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[q]) : "v"(x[q]), "v"(x[(q + 1) & 7]));
                    else asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(x[q]) : "v"(x[(q + 2) & 7]), "v"(x[(q + 4) & 7]), "v"(u[(q + 6) & 7]));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += xp[i].x + xp[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int NF, int KIND>
static void run(int threads, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NF, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NF, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c[16]; (void)hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    unsigned long long lo = ~0ull, hi = 0; double own = 0;
    for (int w = 0; w < nw; ++w) { lo = c[2 * w] < lo ? c[2 * w] : lo; hi = c[2 * w + 1] > hi ? c[2 * w + 1] : hi; own += (double)(c[2 * w + 1] - c[2 * w]) / nw; }
    printf("  waves/SIMD %d  fillers/MFMA %2d (%s): %.1f matrix-pipe cycles per MFMA (workgroup span / MFMAs per SIMD), a wave's own loop %.1f cycles per MFMA\n",
           threads / 256, NF, KIND == 0 ? "split mix" : (KIND == 1 ? "v_fma" : (KIND == 2 ? "v_pk_add_f32" : "fp16x2 split mix")),
           (double)(hi - lo) / (iters * 4.0 * (threads / 256)), own / (iters * 4.0));
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 128);
    for (int threads : {256, 512}) {
        run<0, 0>(threads, out, cyc); run<2, 0>(threads, out, cyc); run<3, 0>(threads, out, cyc);
        run<4, 0>(threads, out, cyc); run<5, 0>(threads, out, cyc); run<6, 0>(threads, out, cyc);
        run<8, 0>(threads, out, cyc); run<12, 0>(threads, out, cyc);
        run<4, 1>(threads, out, cyc); run<6, 1>(threads, out, cyc); run<8, 1>(threads, out, cyc);
        run<2, 2>(threads, out, cyc); run<4, 2>(threads, out, cyc);
        run<3, 3>(threads, out, cyc); run<5, 3>(threads, out, cyc); run<6, 3>(threads, out, cyc); run<8, 3>(threads, out, cyc);
    }
    return 0;
}
