// Stream lab: how fast can ONE workgroup per CU pull a weight image from L2 into LDS, on G of the 256 CUs?
// The chain kernels (graphinvent_amd/csrc/gi_chain.hip) stream a ~1 MB image per workgroup through LDS by LDS-DMA
// (global_load_lds_dwordx4); the fp16x2 dZ chain does so on half the CUs and is as long as the fp32 chain on all of
// them.  Is that the per-CU rate of the LDS-DMA path, or bytes in flight?  Variants:
//   dma  D : LDS-DMA, 16 KB tiles, D tiles in flight (ring of D + 1 slots), one barrier per tile (the chain's structure)
//   reg  D : the same tiles through VGPRs (2 x dwordx4 per thread and tile, D tiles in flight in registers), ds_write_b128
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_lab.hip -o tools/stream_lab
//   usage: stream_lab <grid> [image_KB per group = 1150] [groups = 3]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int TILE = 4096;                                   // floats per 16 KB tile

__device__ __forceinline__ void lds_dma_1k(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// consume a tile: every thread reads 32 bytes of it from LDS and folds them into a checksum (stands for the fragment reads)
__device__ __forceinline__ float consume(const float* tile, int tid) {
    const v4f a = *(const v4f*)(tile + 4 * tid), b = *(const v4f*)(tile + 2048 + 4 * tid);
    return a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
}

template <int D>
__global__ __launch_bounds__(512) void dma_kernel(const float* img, long long group_floats, int groups, int tiles, float* out) {
    __shared__ __attribute__((aligned(1024))) float Bs[(D + 1) * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* src = img + (long long)(blockIdx.x % groups) * group_floats + (wid * 2) * 256 + lane * 4;
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)Bs + (unsigned)(wid * 2) * 1024u));
    auto dma = [&](int t) {
        t = min(t, tiles - 1);
        const float* s = src + (long long)t * TILE;
        const unsigned d = base + (unsigned)__builtin_amdgcn_readfirstlane(t % (D + 1)) * (unsigned)(TILE * 4);
        lds_dma_1k(s, d); lds_dma_1k(s + 256, d + 1024u);
    };
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < D; ++t) dma(t);
    for (int s = 0; s < tiles; ++s) {
        // this wave's two pieces of tile s have landed when at most 2 (D - 1) younger loads are outstanding
        if (D == 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (D == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        if (D == 3) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        if (D == 4) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        if (D == 6) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
        dma(s + D);
        acc += consume(Bs + (s % (D + 1)) * TILE, tid);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

template <int D>
__global__ __launch_bounds__(512) void reg_kernel(const float* img, long long group_floats, int groups, int tiles, float* out) {
    __shared__ __attribute__((aligned(1024))) float Bs[2 * TILE];
    const int tid = threadIdx.x;
    const float* src = img + (long long)(blockIdx.x % groups) * group_floats + 4 * tid;
    v4f r[D][2];
    auto load = [&](int t, v4f (&x)[2]) {
        t = min(t, tiles - 1);
        x[0] = *(const v4f*)(src + (long long)t * TILE);
        x[1] = *(const v4f*)(src + (long long)t * TILE + 2048);
    };
#pragma unroll
    for (int t = 0; t < D; ++t) load(t, r[t]);
    float acc = 0.f;
    for (int s0 = 0; s0 < tiles; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int s = s0 + u;
            float* slot = Bs + (s & 1) * TILE;
            *(v4f*)(slot + 4 * tid) = r[u][0];
            *(v4f*)(slot + 2048 + 4 * tid) = r[u][1];
            load(s + D, r[u]);
            __syncthreads();
            acc += consume(slot, tid ^ 37);
        }
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 131;
    const int kb = argc > 2 ? atoi(argv[2]) : 1150;
    const int groups = argc > 3 ? atoi(argv[3]) : 3;
    const int tiles = kb * 1024 / (TILE * 4);
    const long long gf = (long long)tiles * TILE;
    float *img, *out;
    (void)hipMalloc(&img, gf * groups * 4); (void)hipMalloc(&out, 4096 * 4);
    (void)hipMemset(img, 0, gf * groups * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) launch();
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms / 10 < best ? ms / 10 : best;
        }
        const double bytes = (double)grid * tiles * TILE * 4;
        printf("%-8s grid %3d: %7.1f us per launch, %6.1f GB/s per workgroup, %5.2f TB/s aggregate\n", name, grid,
               best * 1e3, bytes / grid / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
    };
#define L(K, D) time(#K " " #D, [&] { hipLaunchKernelGGL((K##_kernel<D>), dim3(grid), dim3(512), 0, 0, img, gf, groups, tiles, out); })
    L(dma, 1); L(dma, 2); L(dma, 3); L(dma, 4); L(dma, 6);
    L(reg, 2); L(reg, 4); L(reg, 6); L(reg, 8);
    return 0;
}
