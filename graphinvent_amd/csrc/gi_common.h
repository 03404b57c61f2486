// Shared device/host helpers for libgraphinvent_amd (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "graphinvent_amd.h"

#define GI_SELU_ALPHA 1.6732632423543772848170429916717f
#define GI_SELU_SCALE 1.0507009873554804934193349852946f

// exp(x) for x <= 0 as v_exp_f32 (2^t, 1 ulp) on t = fl(x * log2(e)): 2 instructions.  The rounded product costs exp up to
// |t| ulp, i.e. an ABSOLUTE error of at most exp(x) |x| 8.6e-8 <= 3.2e-8 — invisible in SELU, whose exp(x) - 1 rounds at
// 6e-8 whatever exp's last bits are (measured on the device against the fp64 value over x in [-20, -1e-6]: max |err|
// 2.04e-7, mean signed error -6.4e-9; with the first-order correction of rounds 2-5 — the product's rounding error and
// the low part of log2(e) folded back in, three more instructions — 1.93e-7 and -7.0e-9: the same).  Round 6 dropped the
// correction: 60 -> 48 cycles per value in every forward epilogue of the path, headline step -0.75 %
// (profiles/r06/ab_head_selu_bare_exp.txt).  -DGI_EXP_CORRECTED brings it back.  (libm expf costs ~25 VALU instructions.)
__device__ __forceinline__ float gi_exp_nonpos(float x) {
    const float l2e_hi = 1.44269502162933349609375f;
#ifdef GI_EXP_CORRECTED
    const float l2e_lo = 1.925963033500011e-8f;
    const float t = x * l2e_hi;
    const float r = fmaf(x, l2e_hi, -t) + x * l2e_lo;          // (exact product - t) + low part
    return __builtin_amdgcn_exp2f(t) * fmaf(r, 0.693147182464599609375f, 1.f);
#else
    return __builtin_amdgcn_exp2f(x * l2e_hi);
#endif
}
// SELU as torch.nn.SELU (gnn/modules.py:126,164): scale * (x > 0 ? x : alpha * (exp(x) - 1)) —
// the same exp(x) - 1 form ATen's CPU/GPU elu kernels evaluate.
__device__ __forceinline__ float gi_selu(float x) {
    return GI_SELU_SCALE * (x > 0.f ? x : GI_SELU_ALPHA * (gi_exp_nonpos(fminf(x, 0.f)) - 1.f));
}
// d selu(x)/dx through y = selu(x):  y > 0 -> scale,  else y + scale*alpha  (x <= 0 <=> y <= 0)
__device__ __forceinline__ float gi_selu_grad(float y) {
    return y > 0.f ? GI_SELU_SCALE : y + GI_SELU_SCALE * GI_SELU_ALPHA;
}
__device__ __forceinline__ float gi_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

static inline int gi_launch_status() {
    hipError_t e = hipGetLastError();
    return (int)e;
}
static inline int gi_r4(int x) { return (x + 3) & ~3; }
static inline long long gi_r4l(long long x) { return (x + 3) & ~3LL; }
static inline int gi_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- library-internal launchers of the bounded (host-sync-free) forward: the exported functions with the real
// row count read on the device (rows_dev != NULL: `rows` only sizes the grid) ------------------------------
int gi_seg_sum_n(const float* vals, int ldv, const int* perm, const int* off, int rows, int cols, float* out,
                 int ldo, int accumulate, const int* rows_dev, void* stream);
int gi_seg_softmax_fwd_n(const float* en, const float* emb, int ld, const int* perm, const int* off, int rows,
                         int cols, float* out, int ldo, const int* rows_dev, void* stream);
int gi_gru_gates_fwd_n(float* gi, float* gh, int ldg, const float* hx_prev, float* hx_new, int ldh,
                       const int* seg_off, int rows, int H, int Fn, const int* rows_dev, void* stream);

// the fused GRU update of the forward (gi_gru.hip): both projections + the gate arithmetic in one launch
bool gi_gru_fused_ok(int H, int M, int lda, int ldh, int ldg);
int gi_gru_fused_fwd(const float* agg, int lda, const float* hx, int ldh, const float* Wih, const float* Whh,
                     const float* bih, const float* bhh, float* gi, float* gh, int ldg, float* hx_new,
                     const int* seg_off, int rows, const int* rows_dev, int H, int M, void* stream);

// gi_gemm_batch hands launches whose problems all carry GI_GEMM_BF3 to gi_gemm_bf3.hip
int gi_gemm_bf3_launch(const gi_gemm_params* probs, int n, void* stream);
// ... and gi_gemm_bf3_launch those with plain fp32 operands (no images, no gathers) to gi_gemm_b3v.hip: forward,
// dgrad (W as stored, b_major) and weight-gradient (a_major + b_major, split-K slabs) layouts on 32-deep k tiles
bool gi_b3v_eligible(const gi_gemm_params* probs, int n);
bool gi_b3v_wants(const gi_gemm_params* probs, int n);      // every problem carries GI_GEMM_T128 and the kernel can run them
int gi_b3v_launch(const gi_gemm_params* probs, int n, void* stream);
extern "C" int gi_b3v_enable(int on);
// ... and forward / dgrad launches (both operands fp32 with k contiguous) to the 512-thread ping-pong kernel of
// gi_gemm_b3p.hip (128 x 256 tiles, two wave groups alternating between the MFMA pipe and the staging work)
bool gi_b3p_eligible(const gi_gemm_params* probs, int n);
int gi_b3p_launch(const gi_gemm_params* probs, int n, void* stream);
extern "C" int gi_b3p_enable(int on);
// GI_GEMM_LOG line of a launch (gi_gemm.hip); cls: two characters, "00" forward / "01" dgrad / "11" wgrad layouts,
// "b0" / "b1" the bf16x3 launches of the forward / dgrad
void gi_gemm_log_launch(const char* cls, const gi_gemm_params* probs, int n, int blocks, double flops);

// ---- bias-gradient column of weight-gradient slabs, computed on its own (gi_ops.hip) ---------------------------------
// A weight-gradient GEMM [dW | db] = dZ^T [X | 1] carries the bias gradient as an extra "ones" column: n_in + 1 output
// columns.  When n_in is a multiple of the 64-wide tile that one column costs a whole extra column of tiles (129 -> 3
// tiles instead of 2: the GRU projections, every stack's first layer at H = 128).  Such problems run as plain
// n_out x n_in GEMMs and this launch writes column `col` of their slabs: slab s of a problem gets the column sums of dZ
// over the s-th of `nsplit` equal row chunks (fixed order: deterministic), so the slab reduction finds db as before.
struct GiBiasSlab {
    const float* dZ; int lddz;           // [rows, n_out] row-major
    const int* grp_off; int g;           // rows [grp_off[g], grp_off[g + 1]) on the device, or NULL: rows [0, rows)
    int rows, n_out;
    float* slab; long long stride;       // slab s at slab + s * stride, [n_out, ld]
    int ld, col, nsplit;
};
int gi_bias_slabs(const GiBiasSlab* descs, int n, hipStream_t st);

// ---- pass-0 row cache (gi_graph.p0_cache, gi_compact.hip): lookup before the pass-0 stack launch (words[0] =
// hit flag, rows copied into m0 / e0 on a hit), insert after it (no-op on a hit).  nfam = 1 (message rows) or
// 2 (message + energy rows); rows are ldm floats per family.
long long gi_p0_cache_words_for(int row_floats);
int gi_p0_cache_lookup(const int* gfix, int B, int N, int Fe, int* cache, int nfam, float* m0, float* e0, int ldm,
                       void* stream);
int gi_p0_cache_insert(const int* gfix, int B, int N, int Fe, int* cache, int nfam, const float* m0,
                       const float* e0, int ldm, void* stream);

// ---- optional per-launch timing (bench.py roofline leg) -----------------------------------------
// When enabled, gi_gemm / gi_seg_sum bracket each kernel launch with hipEvents on the launch
// stream; gi_prof_collect synchronises and sums the elapsed times.  Off by default (zero cost).
enum { GI_PROF_GEMM = 0, GI_PROF_SEGSUM = 1, GI_PROF_KINDS = 2 };
// matrix pipe of a GEMM launch, in bits 8.. of the kind: 0 fp32 MFMA, 1 bf16 MFMA (bf16x3 split), 2 f16 MFMA (fp16x2 split)
enum { GI_PROF_PIPE_BF3 = 1 << 8, GI_PROF_PIPE_X2 = 2 << 8, GI_PROF_PIPES = 3 };
bool gi_prof_on();
void gi_prof_push(int kind, double work, hipEvent_t start, hipEvent_t stop);
struct GiProfScope {
    hipStream_t st; int kind; double work; hipEvent_t a, b; bool on;
    GiProfScope(hipStream_t s, int k, double w) : st(s), kind(k), work(w), a(nullptr), b(nullptr), on(gi_prof_on()) {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    }
    ~GiProfScope() { if (on) { (void)hipEventRecord(b, st); gi_prof_push(kind, work, a, b); } }
};
