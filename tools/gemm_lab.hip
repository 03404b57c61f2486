// GEMM lab: the batched launches of the training step as a standalone program (no torch), for
// kernel experiments and per-workgroup timelines.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igraphinvent_amd/csrc tools/gemm_lab.hip -o tools/gemm_lab
//   (add -DGI_GEMM_TRACE for the per-tile trace: tools/gemm_lab_trace <class> <tm> <tn> <persist> trace.csv)
// usage: gemm_lab <fwd|dgrad|wgrad|fwd1|tier2> <tm> <tn> [persist_tenths [trace.csv]]
//   fwd   : the node-level hidden-layer launch  (2 x 7258x500x500 + 2 x 7258x250x250, contig/contig)
//   dgrad : its backward                         (the same, B operand reduction-major)
//   wgrad : a weight-gradient batch              (3 x 500x501 + 2 x 250x251 + 3x501 + 2 x 100x251 over 7258 rows, split-K slabs)
//   fwd1  : one 7258x500x500 problem
//   fwd3 / dgrad3 / fwd13 / tier23: the same launches on the bf16 MFMA pipe (GI_GEMM_BF3: B pre-split by gi_bf3_pack)
//   fwd3f / dgrad3f with GI_B3V=1 (default) run on the 32-deep-tile kernels of gi_gemm_b3v.hip, GI_B3V=0 on the round-3
//   kernel; dgrad3m: W as stored (b_major, no transposed copy; gi_gemm_b3v.hip only); wgrad3: the weight-gradient batch
//   on the bf16 pipe (128x128 tiles, GI_LAB_WMUL (default 4) times the slabs of the 64x64-tile launch)
//   tier2 : the graph-level hidden-layer launch (3 x 1000x500x500)
//   tier2s: the same as GI_LAB_NSPLIT (default 2) split-K slabs per problem, no epilogue (what a k-split of
//           the launch would buy: more, shorter workgroups per CU)
// persist_tenths: 0 = one workgroup per tile, 11 = library default (persistent grid for launches of more
// than 1.1 rounds of resident workgroups), 1 = always persistent.  Every run checks 64 random outputs per
// problem against a double-precision dot product.
#include "../graphinvent_amd/csrc/gi_gemm.hip"
#include "../graphinvent_amd/csrc/gi_gemm_bf3.hip"
#include "../graphinvent_amd/csrc/gi_gemm_b3v.hip"
#include "../graphinvent_amd/csrc/gi_gemm_b3p.hip"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
bool gi_prof_on() { return false; }
void gi_prof_push(int, double, hipEvent_t, hipEvent_t) {}

struct Mat { std::vector<float> h; float* d; };
static float g_fill = 1.f;                         // GI_LAB_FILL=0: zero-filled operands (DVFS probe)
static Mat make(size_t n, unsigned seed, float scale) {
    Mat m; m.h.resize(n);
    scale *= g_fill;
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; m.h[i] = scale * ((int)(s >> 8) / 8388608.f - 1.f); }
    (void)hipMalloc(&m.d, n * 4); (void)hipMemcpy(m.d, m.h.data(), n * 4, hipMemcpyHostToDevice);
    return m;
}
static inline int r4(int x) { return (x + 3) & ~3; }
static double selu(double x) { return 1.0507009873554804934193349852946 * (x > 0 ? x : 1.6732632423543772848170429916717 * (exp(x) - 1)); }

int main(int argc, char** argv) {
    char clsbuf[32]; strncpy(clsbuf, argc > 1 ? argv[1] : "fwd", 31); clsbuf[31] = 0;
    bool bf3 = false, bf3a = false, bf3f = false, bf3m = false;        // "...3": B pre-split; "...3a": A pre-split as well; "...3f": B as fp32; "...3m": operands as stored
    if (strlen(clsbuf) > 2 && !strcmp(clsbuf + strlen(clsbuf) - 2, "3m")) { bf3 = bf3m = true; clsbuf[strlen(clsbuf) - 2] = 0; }
    else
    if (strlen(clsbuf) > 2 && !strcmp(clsbuf + strlen(clsbuf) - 2, "3f")) { bf3 = bf3f = true; clsbuf[strlen(clsbuf) - 2] = 0; }
    else
    if (strlen(clsbuf) > 2 && !strcmp(clsbuf + strlen(clsbuf) - 2, "3a")) { bf3 = bf3a = true; clsbuf[strlen(clsbuf) - 2] = 0; }
    else if (strlen(clsbuf) > 1 && clsbuf[strlen(clsbuf) - 1] == '3') { bf3 = true; clsbuf[strlen(clsbuf) - 1] = 0; }
    const char* cls = clsbuf;
    const int tm = argc > 2 ? atoi(argv[2]) : 1, tn = argc > 3 ? atoi(argv[3]) : 1;
    const int persist = argc > 4 ? atoi(argv[4]) : 11;
    const char* trace_path = argc > 5 ? argv[5] : nullptr;
    if (getenv("GI_LAB_FILL")) g_fill = (float)atof(getenv("GI_LAB_FILL"));
    gi_gemm_config(persist, 0);
    // GI_LAB_M: rows of the node-level classes (default: the headline batch)
    int M = getenv("GI_LAB_M") ? atoi(getenv("GI_LAB_M")) : 7258, n = 4, dims[8][3] = {{500, 500, 0}, {500, 500, 0}, {250, 250, 0}, {250, 250, 0}};
    const bool dgrad = !strcmp(cls, "dgrad"), wgrad = !strcmp(cls, "wgrad");
    const int wmul = (wgrad && bf3) ? (getenv("GI_LAB_WMUL") ? atoi(getenv("GI_LAB_WMUL")) : 4) : 1;
    if (wgrad && bf3) bf3m = true;
    if (!strcmp(cls, "fwd1")) n = 1;
    const bool tier2s = !strcmp(cls, "tier2s");
    const int lab_nsplit = getenv("GI_LAB_NSPLIT") ? atoi(getenv("GI_LAB_NSPLIT")) : 2;
    if (!strcmp(cls, "tier2") || tier2s) { M = 1000; n = 3; dims[2][0] = dims[2][1] = 500; }
    if (wgrad) {
        n = 8;
        const int w[8][3] = {{500, 500, 3}, {500, 500, 3}, {500, 500, 3}, {250, 250, 12}, {250, 250, 12}, {3, 500, 24}, {100, 250, 24}, {100, 250, 24}};
        memcpy(dims, w, sizeof(w));
    }
    if (getenv("GI_LAB_N")) n = atoi(getenv("GI_LAB_N")) < n ? atoi(getenv("GI_LAB_N")) : n;   // only the first problems
    if (getenv("GI_LAB_SQ") && !wgrad) {               // GI_LAB_SQ=250: three square problems (one message layer, one per bond type)
        n = 3;
        for (int i = 0; i < n; ++i) dims[i][0] = dims[i][1] = atoi(getenv("GI_LAB_SQ"));
    }
    if (getenv("GI_LAB_DIMS") && !wgrad) {             // GI_LAB_DIMS="500x128,500x128,250x136": N x K of every problem (up to 8)
        n = 0;
        for (const char* q = getenv("GI_LAB_DIMS"); *q && n < 8; ++n) {
            dims[n][0] = atoi(q); while (*q && *q != 'x') ++q; if (*q) ++q;
            dims[n][1] = atoi(q); while (*q && *q != ',') ++q; if (*q) ++q;
        }
    }
    gi_gemm_params probs[8];
    Mat A[8], B[8], Cm[8], bias[8], act[8];
    int ldb_host[8] = {};                              // leading dimension of the host copy B[i].h (p.ldb may change with the operand form)
    double flops = 0;
    for (int i = 0; i < n; ++i) {
        gi_gemm_params& p = probs[i];
        memset(&p, 0, sizeof(p));
        p.nsplit = 1; p.ones_col = -1; p.tm = tm; p.tn = tn;
        if (!wgrad) {
            const int N = dims[i][0], K = dims[i][1], ldk = r4(K), ldn = r4(N);
            A[i] = make((size_t)M * ldk, 11 + i, 1.f); p.A = A[i].d; p.lda = ldk;
            Cm[i] = make((size_t)M * ldn * (tier2s ? lab_nsplit : 1), 21 + i, 1.f); p.C = Cm[i].d; p.ldc = ldn;
            p.M = M; p.N = N; p.K = K;
            if (tier2s) {
                B[i] = make((size_t)N * K, 31 + i, 0.06f); p.B = B[i].d; p.ldb = K;
                p.flags = GI_GEMM_SPLITK; p.nsplit = lab_nsplit; p.c_split_stride = (long long)M * ldn;
            } else if (!dgrad) {
                B[i] = make((size_t)N * K, 31 + i, 0.06f); p.B = B[i].d; p.ldb = K;
                bias[i] = make(N, 41 + i, 0.1f); p.bias = bias[i].d;
                p.flags = GI_EPI_BIAS | GI_EPI_SELU;
            } else {                                   // dX[M, N] = dZ[M, K] W[K, N] * selu'(act)
                B[i] = make((size_t)K * N, 31 + i, 0.06f); p.B = B[i].d; p.ldb = N; p.b_major = 1;
                act[i] = make((size_t)M * ldn, 51 + i, 1.f); p.act = act[i].d; p.ldact = ldn;
                p.flags = GI_EPI_DSELU;
            }
            flops += 2.0 * M * N * K;
        } else {                                       // [dW | db] = dZ^T [X | 1], reduction over M rows
            const int no = dims[i][0], ni = dims[i][1];
            const int ns = getenv("GI_LAB_NSPLIT_ALL") ? atoi(getenv("GI_LAB_NSPLIT_ALL")) : dims[i][2] * wmul;   // the same slab count for every problem
            A[i] = make((size_t)M * r4(no), 11 + i, 1.f); p.A = A[i].d; p.lda = r4(no); p.a_major = 1;
            B[i] = make((size_t)M * r4(ni), 31 + i, 1.f); p.B = B[i].d; p.ldb = r4(ni); p.b_major = 1;
            p.M = no; p.N = ni + 1; p.K = M; p.ones_col = ni; p.ldc = r4(ni + 1);
            p.nsplit = ns; p.c_split_stride = r4(no * p.ldc); p.flags = GI_GEMM_SPLITK;
            Cm[i] = make((size_t)ns * p.c_split_stride, 21 + i, 1.f); p.C = Cm[i].d;
            flops += 2.0 * M * no * (ni + 1);
        }
    }
    for (int i = 0; i < n; ++i) ldb_host[i] = probs[i].ldb;
    if (bf3) {
        if (tier2s) { printf("no bf3 variant of this class\n"); return 1; }
        for (int i = 0; i < n; ++i) {
            gi_gemm_params& p = probs[i];
            if (bf3m) { p.flags |= GI_GEMM_BF3 | (p.b_major ? 0 : GI_GEMM_BF3B_F32); continue; }    // operands as stored
            if (bf3f && !p.b_major) { p.flags |= GI_GEMM_BF3 | GI_GEMM_BF3B_F32; continue; }    // forward: W as stored
            gi_bf3_pack_desc d = {};
            d.as_f32 = bf3f ? 1 : 0;                              // dgrad "3f": W^T as a plain fp32 copy
            d.W = p.B; d.rows = p.N; d.cols = p.K; d.ld = p.ldb; d.transpose = p.b_major;
            const long long ne = gi_bf3_image_elems(p.N, p.K);
            (void)hipMalloc(&d.image, ne * 2);
            const int rc = gi_bf3_pack(&d, 1, 0);
            if (rc) { printf("pack rc %d\n", rc); return 1; }
            p.B = (const float*)d.image; p.b_major = 0; p.flags |= GI_GEMM_BF3;
            if (bf3f) { p.flags |= GI_GEMM_BF3B_F32; p.ldb = r4(p.K); }
            if (bf3a) {
                gi_bf3_pack_desc a = {};
                a.W = p.A; a.rows = p.M; a.cols = p.K; a.ld = p.lda; a.transpose = 0;
                (void)hipMalloc(&a.image, gi_bf3_image_elems(p.M, p.K) * 2);
                if (gi_bf3_pack(&a, 1, 0)) { printf("pack A failed\n"); return 1; }
                p.A = (const float*)a.image; p.flags |= GI_GEMM_BF3A;
            }
        }
        (void)hipDeviceSynchronize();
    }
    if (bf3 && getenv("GI_LAB_X2") && atoi(getenv("GI_LAB_X2"))) {     // fp16x2: two scaled fp16 planes, three products
        float* am; (void)hipMalloc(&am, 64 * 4 * GI_AMAX_WORDS);
        std::vector<float> hm(64 * GI_AMAX_WORDS, 0.f);
        for (int i = 0; i < n; ++i) {
            gi_gemm_params& p = probs[i];
            for (float v : A[i].h) hm[2 * i * GI_AMAX_WORDS] = fmaxf(hm[2 * i * GI_AMAX_WORDS], fabsf(v));
            for (float v : B[i].h) hm[(2 * i + 1) * GI_AMAX_WORDS] = fmaxf(hm[(2 * i + 1) * GI_AMAX_WORDS], fabsf(v));
            p.a_amax = am + 2 * i * GI_AMAX_WORDS; p.b_amax = am + (2 * i + 1) * GI_AMAX_WORDS; p.flags |= GI_GEMM_X2;
        }
        (void)hipMemcpy(am, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
        printf("[fp16x2] ");
    }
    // GI_LAB_TWO=1 (4-problem classes): problems {0, 2} and {1, 3} as two launches on two streams, the timed region
    // ends when both streams have finished (do two half launches that drift out of phase beat one full launch?)
    const bool two = getenv("GI_LAB_TWO") && atoi(getenv("GI_LAB_TWO")) && n == 4;
    hipStream_t s2[2] = {nullptr, nullptr};
    hipEvent_t ej[2] = {nullptr, nullptr}, es = nullptr;
    gi_gemm_params half[2][2];
    if (two) {
        for (int i = 0; i < 2; ++i) { (void)hipStreamCreateWithFlags(&s2[i], hipStreamNonBlocking); (void)hipEventCreateWithFlags(&ej[i], hipEventDisableTiming); }
        (void)hipEventCreateWithFlags(&es, hipEventDisableTiming);
        half[0][0] = probs[0]; half[0][1] = probs[2]; half[1][0] = probs[1]; half[1][1] = probs[3];
        printf("[two streams] ");
    }
    int lab_reps = 1;
    auto launch = [&] {
        if (two) {                                  // lab_reps launches per stream between one fork and one join
            (void)hipEventRecord(es, 0);
            for (int h = 0; h < 2; ++h) {
                (void)hipStreamWaitEvent(s2[h], es, 0);
                for (int i = 0; i < lab_reps; ++i) { const int rc = gi_gemm_batch(half[h], 2, s2[h]); if (rc) { printf("rc %d\n", rc); exit(1); } }
                (void)hipEventRecord(ej[h], s2[h]);
                (void)hipStreamWaitEvent(0, ej[h], 0);
            }
            return;
        }
        for (int i = 0; i < lab_reps; ++i) { const int rc = gi_gemm_batch(probs, n, 0); if (rc) { printf("rc %d\n", rc); exit(1); } }
    };
    for (int i = 0; i < 5; ++i) launch();
    (void)hipDeviceSynchronize();
    // ---- spot check against double-precision dot products ------------------------------------------
    double worst = 0;
    for (int i = 0; i < n; ++i) {
        const gi_gemm_params& p = probs[i];
        const size_t csz = (wgrad || tier2s) ? (size_t)p.nsplit * p.c_split_stride : (size_t)M * p.ldc;
        std::vector<float> c(csz);
        (void)hipMemcpy(c.data(), p.C, csz * 4, hipMemcpyDeviceToHost);
        unsigned s = 777 + i;
        for (int t = 0; t < 64; ++t) {
            s = s * 1664525u + 1013904223u; const int r = (t < 2 ? (t ? p.M - 1 : 0) : (s >> 8) % p.M);
            s = s * 1664525u + 1013904223u; const int cidx = (t < 2 ? (t ? p.N - 1 : 0) : (s >> 8) % p.N);
            double ref = 0, got = 0, scale = 1;
            if (tier2s) {
                for (int k = 0; k < p.K; ++k) ref += (double)A[i].h[(size_t)r * p.lda + k] * B[i].h[(size_t)cidx * ldb_host[i] + k];
                for (int sp = 0; sp < p.nsplit; ++sp) got += c[(size_t)sp * p.c_split_stride + (size_t)r * p.ldc + cidx];
            } else if (!wgrad && !dgrad) {
                for (int k = 0; k < p.K; ++k) ref += (double)A[i].h[(size_t)r * p.lda + k] * B[i].h[(size_t)cidx * ldb_host[i] + k];
                ref = selu(ref + bias[i].h[cidx]); got = c[(size_t)r * p.ldc + cidx];
            } else if (dgrad) {
                for (int k = 0; k < p.K; ++k) ref += (double)A[i].h[(size_t)r * p.lda + k] * B[i].h[(size_t)k * ldb_host[i] + cidx];
                const double y = act[i].h[(size_t)r * p.ldact + cidx];
                ref *= (y > 0 ? 1.0507009873554804934193349852946 : y + 1.0507009873554804934193349852946 * 1.6732632423543772848170429916717);
                got = c[(size_t)r * p.ldc + cidx];
            } else {
                for (int k = 0; k < p.K; ++k)
                    ref += (double)A[i].h[(size_t)k * p.lda + r] * (cidx == p.ones_col ? 1.0 : (double)B[i].h[(size_t)k * p.ldb + cidx]);
                for (int sp = 0; sp < p.nsplit; ++sp) got += c[(size_t)sp * p.c_split_stride + (size_t)r * p.ldc + cidx];
                scale = sqrt((double)p.K);
            }
            const double err = fabs(got - ref) / (scale * 1.0 + fabs(ref));
            if (err > worst) worst = err;
        }
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    const int rounds = 7, reps = 20;
    for (int r = 0; r < rounds; ++r) {
        (void)hipEventRecord(e0);
        if (two) { lab_reps = reps; launch(); lab_reps = 1; }
        else for (int i = 0; i < reps; ++i) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    printf("%s%s tile=(%d,%d) persist=%d: %.2f us per launch (best %.2f), %.1f TF (best %.1f) = %.3f of 157.3   max rel err %.2e %s\n",
           cls, bf3m ? "3m (bf16x3, operands as stored)" : bf3f ? "3f (bf16x3, B fp32)" : bf3a ? "3a (bf16x3, A pre-split)" : (bf3 ? "3 (bf16x3)" : ""), tm, tn, persist, sum / rounds / reps * 1e3, best / reps * 1e3, flops * reps * rounds / (sum * 1e-3) / 1e12,
           flops * reps / (best * 1e-3) / 1e12, flops * reps * rounds / (sum * 1e-3) / 1e12 / 157.3, worst,
           worst < 2e-5 ? "OK" : "MISMATCH");
#ifdef GI_B3P_TRACE
    if (trace_path) {
        unsigned long long* buf; (void)hipMalloc(&buf, 2048 * 8); (void)hipMemset(buf, 0, 2048 * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gp_trace_buf), &buf, sizeof(buf));
        launch(); (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(2048);
        (void)hipMemcpy(h.data(), buf, 2048 * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(trace_path, "w");
        // six stamps per iteration.  group 0: compute start, compute end | stage start, loads landed, LDS written, loads issued;
        // group 1: stage start, loads landed, LDS written, loads issued | compute start, compute end
        for (int g = 0; g < 2; ++g) {
            fprintf(f, "group %d: durations (cycles) per iteration: %s\n", g, g == 0 ?
                    "compute, barrier, wait-loads, split+write, issue-loads, barrier" : "wait-loads, split+write, issue-loads, barrier, compute, barrier");
            for (int i = 0; i + 6 < 1024 && h[g * 1024 + i + 6]; i += 6) {
                const unsigned long long* t = &h[g * 1024 + i];
                if (g == 0) fprintf(f, "  %6llu: %5llu %5llu %5llu %5llu %5llu %5llu\n", t[0] - h[0], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5]);
                else fprintf(f, "  %6llu: %5llu %5llu %5llu %5llu %5llu %5llu\n", t[0] - h[0], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5]);
            }
        }
        fclose(f);
        printf("phase trace -> %s\n", trace_path);
    }
#endif
#ifdef GI_GEMM_TRACE
    if (trace_path) {
        int total = 0;
        for (int i = 0; i < n; ++i) {
            const gi_gemm_params& p = probs[i];
            total += ((p.N + 64 * tn - 1) / (64 * tn)) * ((p.M + 64 * tm - 1) / (64 * tm)) * (wgrad ? p.nsplit : 1);
        }
        unsigned long long* buf; (void)hipMalloc(&buf, (size_t)total * 64); (void)hipMemset(buf, 0, (size_t)total * 64);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gi_trace_buf), &buf, sizeof(buf));
        launch(); (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)total * 8);
        (void)hipMemcpy(h.data(), buf, (size_t)total * 64, hipMemcpyDeviceToHost);
        FILE* f = fopen(trace_path, "w");
        fprintf(f, "wg,t_start,t_prologue,t_loop,t_end,hw_id,xcc_id,block\n");
        for (int w = 0; w < total; ++w)
            fprintf(f, "%d,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3],
                    h[w * 8 + 4], h[w * 8 + 5], h[w * 8 + 6]);
        fclose(f);
        printf("trace of %d tiles -> %s\n", total, trace_path);
    }
#endif
    return worst < 2e-5 ? 0 : 2;
}
