// Fused GRU update of a message pass (gfx950): both input projections AND the gate arithmetic in ONE launch.
//
//   gi = agg W_ih^T + b_ih,  gh = h W_hh^T + b_hh                       (torch.nn.GRUCell, gnn/mpnn.py:296-297)
//   r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h
//
// Before (rounds 1-5): one batched GEMM launch for the two projections (2 x [R, 3H] written to HBM/L2: 22 MB at the
// headline batch), then gru_gates_fwd reading both back — 24.7 + 8.6 us and two launch latencies per pass on the
// forward's critical path for 1.4 GFLOP.  Here a workgroup owns 64 rows x 64 hidden units: it accumulates the SIX
// 64 x 64 blocks that those units need (gates r, z, n of both projections; r and z of the two projections share an
// accumulator: four accumulators per wave) on v_mfma_f32_32x32x2_f32 — exact fp32 products, like every narrow GEMM of
// the path — and applies the gates in registers.  It stores what the backward reads (gru_gates_bwd: r, z, n in gi,
// gh_n in gh) and h'.  Nodes without incoming edges keep their state (gnn/summation_mpnn.py:146: node_mask).
//
// Structure: 256 threads = 2 x 2 waves, a wave = one 32 x 32 quadrant of the 64 x 64 unit tile for all gates.  The
// reduction runs in 16-deep chunks: first over the message width M (A = agg, B = W_ih), then over H (A = h, B = W_hh);
// per chunk a thread stages one float4 of A and three of B (192 weight rows: 3 gates x 64 units) through registers into
// a double-buffered LDS image (rows padded to 18 floats: conflict-free 8-byte fragment reads); the MFMA's two k slots
// take elements (4q, 4q + 2) and (4q + 1, 4q + 3) of a 4-group, so that a lane's fragment is one ds_read_b64.
#include <stdlib.h>

#include "gi_common.h"
#include "gi_mfma.h"

namespace {

#ifndef GI_GRU_KC
#define GI_GRU_KC 16
#endif
constexpr int GU_TM = 64, GU_TN = 64, GU_KC = GI_GRU_KC, GU_LD = GU_KC + 2;
constexpr int GU_QPR = GU_KC / 4;                               // float4 per row of a chunk
constexpr int GU_NA = GU_TM * GU_QPR / 256, GU_NW = 3 * GU_TN * GU_QPR / 256;    // float4 per thread and chunk: A, W
typedef float gu_f32x2 __attribute__((ext_vector_type(2)));

struct GruArgs {
    const float* agg; int lda;              // [R, lda]: aggregated messages (M columns)
    const float* hx; int ldh;               // [R, ldh]: [h (H) | x | padding]
    const float* Wih; const float* Whh;     // [3H, M], [3H, H] row-major (torch.nn.GRUCell)
    const float* bih; const float* bhh;     // [3H]
    float* gi; float* gh; int ldg;          // [R, ldg]: r, z, n -> gi;  gh_n -> gh[:, 2H:3H]
    float* hx_new;                          // [R, ldh]
    const int* seg_off;                     // [R + 1]: incoming-edge CSR offsets (node mask)
    int rows; const int* rows_dev;          // rows_dev != NULL: the real row count lives on the device (bounded forward)
    int H, M;
};

// 512 threads = TWO groups of 2 x 2 waves: group 0 accumulates the input projection (agg W_ih^T, reduction over M), group 1
// the hidden projection (h W_hh^T, over H) — the two are independent GEMMs, so every SIMD hosts one wave of each and
// their LDS / barrier / load latencies overlap (one group alone: 35 us per launch for 12 us of MFMA time, measured).
// Group 1 hands its three accumulators to group 0 through LDS (over the staging buffers, which are dead by then), and
// group 0 applies the gates.
__global__ __launch_bounds__(512) void gru_fused_fwd_kernel(const GruArgs a) {
    constexpr int A_FL = 2 * GU_TM * GU_LD, W_FL = 2 * 3 * GU_TN * GU_LD, G_FL = A_FL + W_FL;     // floats per group
    __shared__ __attribute__((aligned(16))) float lds[2 * G_FL > 3 * 16 * 256 ? 2 * G_FL : 3 * 16 * 256];
    __shared__ int live_s[GU_TM];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, tg = tid & 255;
    const int rh = (wid >> 1) & 1, ch = wid & 1, l31 = lane & 31, lg = lane >> 5;
    float (*As)[GU_TM][GU_LD] = reinterpret_cast<float (*)[GU_TM][GU_LD]>(lds + grp * G_FL);
    float (*Ws)[3 * GU_TN][GU_LD] = reinterpret_cast<float (*)[3 * GU_TN][GU_LD]>(lds + grp * G_FL + A_FL);
    const int rows = a.rows_dev ? min(a.rows, *a.rows_dev) : a.rows;
    const int row0 = blockIdx.x * GU_TM, j0 = blockIdx.y * GU_TN;
    if (row0 >= rows) return;
    const int H = a.H, M = a.M;
    if (tid < GU_TM) {
        const int row = row0 + tid;
        live_s[tid] = (row < rows && a.seg_off[row + 1] > a.seg_off[row]) ? 1 : 0;
    }
    // feature tail / padding of the state rows: plain copy (the workgroups of the first unit tile)
    if (blockIdx.y == 0) {
        const int tail4 = (a.ldh - H) >> 2;                      // (H % 4 == 0, ldh % 4 == 0)
        for (int i = tid; i < GU_TM * tail4; i += 512) {
            const int r = i / tail4, c = H + 4 * (i - r * tail4);
            if (row0 + r < rows)
                *(v4f*)(a.hx_new + (long long)(row0 + r) * a.ldh + c) = *(const v4f*)(a.hx + (long long)(row0 + r) * a.ldh + c);
        }
    }

    // ---- staging (per group): thread -> GU_NA float4 of the A chunk, GU_NW of the B chunk ------------------------------
    const int K = grp ? H : M;
    const float* const Abase = grp ? a.hx : a.agg;
    const int lda_ = grp ? a.ldh : a.lda;
    const float* const W = grp ? a.Whh : a.Wih;
    const int nci = (M + GU_KC - 1) / GU_KC, nch = (H + GU_KC - 1) / GU_KC;
    const int mine = grp ? nch : nci, nc = max(nci, nch);
    v4f ra_[2][GU_NA], rb_[2][GU_NW];                            // TWO register stages: chunk c + 2 is requested while chunk c is multiplied
    // (loads are UNCONDITIONAL — addresses clamped into the row, values beyond K zeroed by a select when they are written
    // to LDS: a conditional load is a branch in the k loop)
    auto gload = [&](int c, int set) {
        const int kc0 = min(c, mine - 1) * GU_KC;
#pragma unroll
        for (int i = 0; i < GU_NA; ++i) {
            const int idx = tg + 256 * i, ar = idx / GU_QPR, aq = idx % GU_QPR;
            const long long row = min(row0 + ar, rows - 1);
            ra_[set][i] = *(const v4f*)(Abase + row * lda_ + min(kc0 + 4 * aq, K - 4));
        }
#pragma unroll
        for (int i = 0; i < GU_NW; ++i) {
            const int idx = tg + 256 * i, wr = idx / GU_QPR;      // 192 weight rows x GU_QPR float4
            const int unit = min(j0 + (wr & 63), H - 1);         // (units beyond H: any readable row, discarded)
            const float* wrow = W + (long long)((wr >> 6) * H + unit) * K;
            rb_[set][i] = *(const v4f_u*)(wrow + min(kc0 + 4 * (idx % GU_QPR), K - 4));
        }
    };
    auto sstore = [&](int c, int set) {
        const int buf = c & 1, kc0 = c * GU_KC;
        const v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < GU_NA; ++i) {
            const int idx = tg + 256 * i, ar = idx / GU_QPR, aq = idx % GU_QPR;
            *(v4f_u*)&As[buf][ar][4 * aq] = (kc0 + 4 * aq < K) ? ra_[set][i] : zero;      // (K % 4 == 0: a float4 is inside or outside)
        }
#pragma unroll
        for (int i = 0; i < GU_NW; ++i) {
            const int idx = tg + 256 * i;
            *(v4f_u*)&Ws[buf][idx / GU_QPR][4 * (idx % GU_QPR)] = (kc0 + 4 * (idx % GU_QPR) < K) ? rb_[set][i] : zero;
        }
    };

    f32x16 acc[3];                                               // gates r, z, n of this group's projection
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    auto compute = [&](int c) {
        const int buf = c & 1;
#pragma unroll
        for (int q = 0; q < GU_QPR; ++q) {
            const gu_f32x2 av = *(const gu_f32x2*)&As[buf][rh * 32 + l31][4 * q + 2 * lg];
            gu_f32x2 bv[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) bv[g] = *(const gu_f32x2*)&Ws[buf][g * 64 + ch * 32 + l31][4 * q + 2 * lg];
            // (three independent accumulators between two MFMAs on the same one)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[g].x, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[g].y, acc[g], 0, 0, 0);
        }
    };
    // Pipeline (chunk c is in LDS buffer c & 1, chunk c + 1 in register set (c + 1) & 1, chunk c + 2 being loaded into set
    // c & 1): request c + 2, multiply c, write c + 1 into the other buffer (its readers finished a barrier ago), barrier.
    // Both groups run the same number of iterations (the shorter reduction idles through the barriers).
    gload(0, 0);
    gload(1, 1);
    sstore(0, 0);
    __syncthreads();
    for (int c = 0; c < nc; c += 2) {                            // (unrolled by two: the register sets are compile-time;
        gload(c + 2, 0);                                         //  past the end: the last chunk again, never used)
        if (c < mine) compute(c);
        if (c + 1 < mine) sstore(c + 1, 1);
        __syncthreads();
        if (c + 1 < nc) {
            gload(c + 3, 1);
            if (c + 1 < mine) compute(c + 1);
            if (c + 2 < mine) sstore(c + 2, 0);
            __syncthreads();
        }
    }

    // ---- group 1 -> group 0: the hidden projection's accumulators, [gate][r][thread of the group] ----------------
    if (grp == 1) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[(g * 16 + r) * 256 + tg] = acc[g][r];
    }
    __syncthreads();
    if (grp == 1) return;

    // ---- gates (C/D layout of a 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----------
    const int j = j0 + ch * 32 + l31;
    const bool j_ok = j < H;
    const int jc = j_ok ? j : H - 1;
    const float br = a.bih[jc] + a.bhh[jc], bz = a.bih[H + jc] + a.bhh[H + jc];
    const float bin = a.bih[2 * H + jc], bhn = a.bhh[2 * H + jc];
    float hp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = rh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
        const long long row = min(row0 + lr, rows - 1);
        hp[r] = a.hx[row * a.ldh + jc];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = rh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
        const int row = row0 + lr;
        if (!j_ok || row >= rows) continue;
        float* g = a.gi + (long long)row * a.ldg;
        float* hrow = a.gh + (long long)row * a.ldg;
        float hn_out = hp[r];
        if (live_s[lr]) {
            const float rr = gi_sigmoid((acc[0][r] + lds[(0 * 16 + r) * 256 + tg]) + br);
            const float zz = gi_sigmoid((acc[1][r] + lds[(1 * 16 + r) * 256 + tg]) + bz);
            const float ghn = lds[(2 * 16 + r) * 256 + tg] + bhn;
            const float nn = tanhf((acc[2][r] + bin) + rr * ghn);
            hn_out = (1.f - zz) * nn + zz * hp[r];
            g[j] = rr; g[H + j] = zz; g[2 * H + j] = nn;
            hrow[2 * H + j] = ghn;
        }
        a.hx_new[(long long)row * a.ldh + j] = hn_out;
    }
}

}  // namespace

// Can the fused kernel take this update?  (otherwise: the two-projection GEMM launch + gi_gru_gates_fwd)
bool gi_gru_fused_ok(int H, int M, int lda, int ldh, int ldg) {
    const bool on = !(getenv("GI_GRU_FUSED") && atoi(getenv("GI_GRU_FUSED")) == 0);     // (read per call: a test switches it mid-process)
    return on && H >= 4 && M >= 4 && (H & 3) == 0 && (M & 3) == 0 && (lda & 3) == 0 && (ldh & 3) == 0 && ldh >= H &&
           lda >= M && ldg >= 3 * H;
}

int gi_gru_fused_fwd(const float* agg, int lda, const float* hx, int ldh, const float* Wih, const float* Whh,
                     const float* bih, const float* bhh, float* gi, float* gh, int ldg, float* hx_new,
                     const int* seg_off, int rows, const int* rows_dev, int H, int M, void* stream) {
    (void)hipGetLastError();
    if (rows <= 0) return 0;
    if (!agg || !hx || !Wih || !Whh || !bih || !bhh || !gi || !gh || !hx_new || !seg_off) return GI_EINVAL;
    if (!gi_gru_fused_ok(H, M, lda, ldh, ldg)) return GI_EINVAL;
    if ((((uintptr_t)agg | (uintptr_t)hx | (uintptr_t)hx_new) & 15) != 0) return GI_EINVAL;
    GruArgs a{agg, lda, hx, ldh, Wih, Whh, bih, bhh, gi, gh, ldg, hx_new, seg_off, rows, rows_dev, H, M};
    hipStream_t st = (hipStream_t)stream;
    GiProfScope prof(st, GI_PROF_GEMM, 2.0 * (double)rows * 3.0 * H * ((double)M + H));
    // ONE workgroup per CU: the kernel is MFMA-bound with one wave per SIMD (four waves = four SIMDs), and the dispatcher
    // packs two 37-KB workgroups onto one CU while others idle (measured: 35 us per launch as dispatched, see below) —
    // unused dynamic LDS makes a second workgroup not fit.
    static const int pad = getenv("GI_GRU_PAD_LDS") ? atoi(getenv("GI_GRU_PAD_LDS")) : 0;
    static bool attr = false;
    const int dyn = pad ? 48 * 1024 : 0;
    if (pad && !attr) {
        if (hipFuncSetAttribute((const void*)gru_fused_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess)
            return (int)hipGetLastError();
        attr = true;
    }
    hipLaunchKernelGGL(gru_fused_fwd_kernel, dim3(gi_cdiv(rows, GU_TM), gi_cdiv(H, GU_TN)), dim3(512), dyn, st, a);
    return gi_launch_status();
}

// C ABI (include/graphinvent_amd.h): the fused update by itself (tests; gi_ggnn_forward calls the function above)
extern "C" int gi_gru_forward(const float* agg, int lda, const float* hx, int ldh, const float* Wih, const float* Whh,
                              const float* bih, const float* bhh, float* gi, float* gh, int ldg, float* hx_new,
                              const int* seg_off, int rows, int H, int M, void* stream) {
    return gi_gru_fused_fwd(agg, lda, hx, ldh, Wih, Whh, bih, bhh, gi, gh, ldg, hx_new, seg_off, rows, nullptr, H, M, stream);
}
