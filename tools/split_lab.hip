// Lab: the fp32 -> two scaled fp16 planes split of gi_x2.h (6 VALU per pair: mul x 2, cvt_pk, fma_mix x 2, cvt_pk) against
// a 4-instruction form on v_fma_mixlo/hi_f16 (the product x s rounded straight to fp16; the residual x s - h1 formed by ONE
// fma from the fp16 half and rounded straight to fp16).  Checks bit equality of both planes over magnitudes that cover
// fp16 normals, subnormals and flush-to-zero, and times both forms.   hipcc --offload-arch=gfx950 -O3 tools/split_lab.hip -o tools/split_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_old(float x0, float x1, float s, unsigned& p0, unsigned& p1) {
    const float y0 = x0 * s, y1 = x1 * s;
    f32x2 v = {y0, y1};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    p0 = __builtin_bit_cast(unsigned, h);
    f32x2 r = {y0 - (float)h.x, y1 - (float)h.y};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ void split_new(float x0, float x1, float s, unsigned& p0, unsigned& p1) {
    unsigned h, r;
    asm volatile("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
                 "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
                 "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
                 "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                 : "=&v"(h), "=&v"(r) : "v"(x0), "v"(x1), "v"(s));
    p0 = h; p1 = r;
}
template <bool NEW> __global__ void k(const float* x, float s, unsigned* o, int n, int reps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float a = x[2 * i], b = x[2 * i + 1];
    unsigned p0 = 0, p1 = 0, acc0 = 0, acc1 = 0;
    for (int r = 0; r < reps; ++r) {
        if (NEW) split_new(a, b, s, p0, p1); else split_old(a, b, s, p0, p1);
        acc0 ^= p0; acc1 ^= p1;
        a = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) ^ (acc1 & 0));   // (keeps the loop from folding)
    }
    o[2 * i] = reps & 1 ? acc0 : p0; o[2 * i + 1] = reps & 1 ? acc1 : p1;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const float m = (float)rand() / RAND_MAX * 2 - 1;
        const int e = rand() % 60 - 45;                       // 2^-45 .. 2^14 after the scale
        h[i] = ldexpf(m, e);
        if (i % 97 == 0) h[i] = 0.f;
        if (i % 101 == 0) h[i] = ldexpf(1.f, e);              // exact powers of two (ties)
    }
    float *dx; unsigned *o0, *o1;
    hipMalloc(&dx, n * 4); hipMalloc(&o0, n * 4); hipMalloc(&o1, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    const float s = 1.0f;                                     // (values are generated at their scaled magnitude)
    hipLaunchKernelGGL(k<false>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, s, o0, n, 1);
    hipLaunchKernelGGL(k<true>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, s, o1, n, 1);
    std::vector<unsigned> a(n), b(n);
    hipMemcpy(a.data(), o0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, n * 4, hipMemcpyDeviceToHost);
    long long diff = 0; int shown = 0;
    for (int i = 0; i < n; ++i) if (a[i] != b[i]) { ++diff; if (shown++ < 8) printf("differ at %d (plane %d): x = %g %g  old %08x new %08x\n", i, i & 1, h[i & ~1], h[i | 1], a[i], b[i]); }
    printf("planes compared: %d words, differing: %lld\n", n, diff);
    for (int which = 0; which < 2; ++which) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (which) hipLaunchKernelGGL(k<true>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, s, o1, n, 2001);
            else hipLaunchKernelGGL(k<false>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, s, o0, n, 2001);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s split: %.3f ms for %d x 2001 pair splits -> %.1f G pairs/s\n", which ? "4-instruction" : "6-instruction", ms, n / 2, n / 2 * 2001.0 / ms / 1e6);
        }
    }
    return diff != 0;
}
