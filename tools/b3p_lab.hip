// Small lab for the ping-pong bf16x3 kernel (gi_gemm_b3p.hip) alone: compiles in seconds, for phase-time experiments.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igraphinvent_amd/csrc [-DGI_B3P_TRACE] [-DGP_DBG=n] tools/b3p_lab.hip -o tools/b3p_lab
//   GP_DBG bit 0: no MFMAs / fragment reads;  bit 1: no split (planes = raw halves);  bit 2: no LDS writes;  bit 3: no global loads in the loop
// usage: b3p_lab [M] [trace.txt]      (2 x Mx500x500 + 2 x Mx250x250, forward epilogue)
#include "../graphinvent_amd/csrc/gi_gemm_b3p.hip"
#include <stdio.h>
#include <vector>
#include <math.h>
bool gi_prof_on() { return false; }
void gi_prof_push(int, double, hipEvent_t, hipEvent_t) {}
void gi_gemm_log_launch(const char*, const gi_gemm_params*, int, int, double) {}
static float* dev(size_t n, unsigned seed, float scale, std::vector<float>* keep = nullptr) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) / 8388608.f - 1.f); }
    float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    if (keep) *keep = h;
    return d;
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 7258;
    const char* trace_path = argc > 2 ? argv[2] : nullptr;
    const int dims[4] = {500, 500, 250, 250};
    gi_gemm_params p[4];
    std::vector<float> hA, hB, hb;
    double flops = 0;
    for (int i = 0; i < 4; ++i) {
        memset(&p[i], 0, sizeof(p[i]));
        const int N = dims[i], K = dims[i];
        p[i].A = dev((size_t)M * K, 11 + i, 1.f, i == 0 ? &hA : nullptr); p[i].lda = K;
        p[i].B = dev((size_t)N * K, 31 + i, 0.06f, i == 0 ? &hB : nullptr); p[i].ldb = K;
        p[i].bias = dev(N, 41 + i, 0.1f, i == 0 ? &hb : nullptr);
        p[i].C = dev((size_t)M * N, 1, 0.f); p[i].ldc = N;
        p[i].M = M; p[i].N = N; p[i].K = K; p[i].nsplit = 1; p[i].ones_col = -1;
        p[i].flags = GI_EPI_BIAS | GI_EPI_SELU | GI_GEMM_BF3 | GI_GEMM_BF3B_F32;
        flops += 2.0 * M * N * K;
    }
    auto launch = [&] { const int rc = gi_b3p_launch(p, 4, 0); if (rc) { printf("rc %d\n", rc); exit(1); } };
    for (int i = 0; i < 5; ++i) launch();
    (void)hipDeviceSynchronize();
    std::vector<float> c((size_t)M * 500);
    (void)hipMemcpy(c.data(), p[0].C, c.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    unsigned s = 777;
    for (int t = 0; t < 200; ++t) {
        s = s * 1664525u + 1013904223u; const int r = (s >> 8) % M;
        s = s * 1664525u + 1013904223u; const int n = (s >> 8) % 500;
        double ref = hb[n];
        for (int k = 0; k < 500; ++k) ref += (double)hA[(size_t)r * 500 + k] * hB[(size_t)n * 500 + k];
        ref = 1.0507009873554804934193349852946 * (ref > 0 ? ref : 1.6732632423543772848170429916717 * (exp(ref) - 1));
        const double e = fabs(c[(size_t)r * 500 + n] - ref) / (1 + fabs(ref));
        worst = e > worst ? e : worst;
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 7; ++r) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    printf("b3p M=%d dbg=%d: %.2f us per launch (best %.2f), %.1f TF   max rel err %.2e\n", M,
#ifdef GP_DBG
           GP_DBG,
#else
           0,
#endif
           sum / 140 * 1e3, best / 20 * 1e3, flops * 140 / (sum * 1e-3) / 1e12, worst);
#ifdef GI_B3P_TRACE
    if (trace_path) {
        unsigned long long* buf; (void)hipMalloc(&buf, 8192 * 8); (void)hipMemset(buf, 0, 8192 * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gp_trace_buf), &buf, sizeof(buf));
        launch(); (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(8192);
        (void)hipMemcpy(h.data(), buf, 8192 * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(trace_path, "w");
        if (!f) { printf("cannot write %s\n", trace_path); return 1; }
        for (int g = 0; g < 2; ++g) {
            fprintf(f, "wave %d: cycles per k tile iteration (48 MFMAs per wave, 2 waves per SIMD):", 4 * g);
            for (int i = 0; i + 1 < 1024 && h[g * 1024 + i + 1]; ++i) fprintf(f, " %llu", h[g * 1024 + i + 1] - h[g * 1024 + i]);
            fprintf(f, "\n");
        }
        unsigned long long t0 = ~0ull;
        for (int bk = 0; bk < 1024 && h[2048 + 4 * bk]; ++bk) t0 = h[2048 + 4 * bk] < t0 ? h[2048 + 4 * bk] : t0;
        fprintf(f, "workgroups (cycles from the first start): block start loop-start loop-end end\n");
        for (int bk = 0; bk < 1024 && h[2048 + 4 * bk]; ++bk)
            fprintf(f, "  %4d %7llu %7llu %7llu %7llu\n", bk, h[2048 + 4 * bk] - t0, h[2048 + 4 * bk + 1] - t0, h[2048 + 4 * bk + 2] - t0, h[2048 + 4 * bk + 3] - t0);
        fclose(f);
    }
#endif
    return 0;
}
