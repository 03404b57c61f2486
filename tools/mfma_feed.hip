// Probe: which feeding path costs MFMA issue slots?  One wave per workgroup, 64x64 tile (4 acc),
// per "k16 tile": 32 MFMAs fed from (a) registers only, (b) ds_read_b128 fragments from a static LDS
// image, (c) + 8 ds_write_b128 per tile, (d) + 8 global_load_dwordx4 per tile (full GEMM feed).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void k(const float* __restrict__ g, float* out, int tiles) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 20];
    const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
    for (int i = lane; i < 2 * 64 * 20; i += 64) lds[i] = 0.001f * i;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    v4f st[8];
    for (int i = 0; i < 8; ++i) st[i] = (v4f){1.f, 2.f, 3.f, 4.f};
    const float* gp = g + (size_t)blockIdx.x * 4096 + lane * 4;
    // MODE 4/5: every wave streams its own rows of a large matrix (row = 64 B segment per 4 lanes,
    // 16 rows per load instruction, like the GEMM's contig staging) -> real L2/MALL/HBM latency
    const float* sp = g + ((size_t)(blockIdx.x % 2048) * 64 + (lane >> 2)) * 4096 + (lane & 3) * 4;
    v4f st2[8];
    if (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 8; ++i) st2[i] = *(const v4f*)(sp + (size_t)(16 * (i & 3)) * 4096 + (i >> 2) * 2048);
    }
    for (int t = 0; t < tiles; ++t) {
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) st[i] = *(const v4f*)(gp + ((t * 8 + i) & 15) * 256);
        }
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                st[i] = *(const v4f*)(sp + (size_t)(16 * (i & 3)) * 4096 + (i >> 2) * 2048 + ((t * 16) & 2047));
        }
        if (MODE == 5) {                      // consume the tile loaded one iteration ago, fetch two ahead
#pragma unroll
            for (int i = 0; i < 8; ++i) st[i] = st2[i];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                st2[i] = *(const v4f*)(sp + (size_t)(16 * (i & 3)) * 4096 + (i >> 2) * 2048 + (((t + 1) * 16) & 2047));
        }
#pragma unroll
        for (int k8 = 0; k8 < 2; ++k8) {
            v4f a[2], b[2];
            if (MODE >= 1) {
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    a[x] = *(const v4f*)&lds[(x * 32 + l31) * 20 + k8 * 8 + 4 * lhi];
                    b[x] = *(const v4f*)&lds[64 * 20 + (x * 32 + l31) * 20 + k8 * 8 + 4 * lhi];
                }
            } else {
#pragma unroll
                for (int x = 0; x < 2; ++x) { a[x] = st[x + 2 * k8]; b[x] = st[4 + x + 2 * k8]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x][j], b[y][j], acc[x][y], 0, 0, 0);
        }
        if (MODE >= 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *(v4f*)&lds[((lane >> 2) + 16 * i) * 20 + 4 * (lane & 3)] = st[i];
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[lane] = s;
}
template <int MODE>
void run(const float* g, float* out, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd, tiles = 128;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, g, out, tiles); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, g, out, tiles);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double flop = 3.0 * blocks * tiles * 32 * (2.0 * 32 * 32 * 2);
    printf("  mode %d waves/SIMD %d: %.1f TF\n", MODE, waves_per_simd, flop / (ms * 1e-3) / 1e12);
}
int main() {
    float *g, *out; (void)hipMalloc(&g, 2048ull * 64 * 4096 * 4 + (1 << 20)); (void)hipMalloc(&out, 4096);
    (void)hipMemset(g, 0, 2048ull * 64 * 4096 * 4 + (1 << 20));
    for (int w = 1; w <= 3; ++w) { run<3>(g, out, w); run<4>(g, out, w); run<5>(g, out, w); }
    return 0;
}
