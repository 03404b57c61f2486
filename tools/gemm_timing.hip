// Per-phase cycle accounting of the wave-tile GEMM (debug build of gi_gemm.hip).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGI_GEMM_TIMING -Iinclude -Igraphinvent_amd/csrc tools/gemm_timing.hip tools/prof_stub.cpp -o tools/gemm_timing
#include "../graphinvent_amd/csrc/gi_gemm.hip"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
bool gi_prof_on() { return false; }
void gi_prof_push(int, double, hipEvent_t, hipEvent_t) {}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 7300, N = argc > 2 ? atoi(argv[2]) : 500, K = argc > 3 ? atoi(argv[3]) : 500;
    const int tm = argc > 4 ? atoi(argv[4]) : 2, tn = argc > 5 ? atoi(argv[5]) : 2;
    float *X, *W, *b, *Y;
    (void)hipMalloc(&X, (size_t)M * K * 4); (void)hipMalloc(&W, (size_t)N * K * 4); (void)hipMalloc(&b, N * 4);
    (void)hipMalloc(&Y, (size_t)M * N * 4);
    (void)hipMemset(X, 0, (size_t)M * K * 4); (void)hipMemset(W, 0, (size_t)N * K * 4); (void)hipMemset(b, 0, N * 4);
    gi_gemm_params p; memset(&p, 0, sizeof(p));
    p.A = X; p.B = W; p.C = Y; p.bias = b; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
    p.flags = GI_EPI_BIAS | GI_EPI_SELU; p.tm = tm; p.tn = tn; p.nsplit = 1; p.ones_col = -1;
    for (int i = 0; i < 3; ++i) gi_gemm(&p, 0);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms;
#ifndef GI_GEMM_TIMING
    const int reps = 50;                                  // clean build: back-to-back launches from C++
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) gi_gemm(&p, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("M=%d N=%d K=%d tile=(%d,%d): %.2f us per launch, %.1f TF\n", M, N, K, tm, tn, ms * 1e3 / reps,
           2.0 * M * N * K * reps / (ms * 1e-3) / 1e12);
    return 0;
#else
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(gi_dbg), z, sizeof(z));
    (void)hipEventRecord(e0); gi_gemm(&p, 0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(gi_dbg), sizeof(z));
    const double w = (double)z[6];
    printf("M=%d N=%d K=%d tile=(%d,%d): kernel %.1f us, waves %.0f; per wave cycles: prologue %.0f  load-issue %.0f  "
           "lds-read+mfma %.0f  wait+lds-write %.0f  epilogue %.0f  lifetime %.0f (%.1f us @2.4GHz)\n",
           M, N, K, tm, tn, ms * 1e3, w, z[0] / w, z[1] / w, z[2] / w, z[3] / w, z[4] / w, z[5] / w, z[5] / w / 2400.0);
    return 0;
#endif
}
