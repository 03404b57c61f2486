// bf16x3 GEMM, "ping-pong" structure (gfx950): 512 threads = two groups of four waves that alternate between the
// MFMA pipe and the staging work, so that the matrix pipe of every SIMD always has one wave feeding it.
//
// Why (round 4, rocprofv3 PMC of the round-3 kernel and of the first 32-deep-tile kernel, profiles/r04): with three
// independent 4-wave workgroups per CU the MFMA pipe was 35 % busy — the workgroups of a CU start together, run the
// same phases and fall into lockstep: all waves of a SIMD queue on the matrix pipe at once (SQ_WAIT_INST_ANY 48 % of
// the wave cycles), then all of them split fp32 into bf16 planes and write LDS at once (~230 non-MFMA instructions per
// wave and k tile at ~4.8 cycles each) while the matrix pipe idles.  Here the overlap is built in:
//
//   block tile 128 x 256 x 32; group g (waves 4g .. 4g+3, 2 x 2 waves of 64 x 64) owns output columns [128 g, 128 g + 128).
//   k step t, LDS stage t & 1:     phase 1: group 0 MFMAs of tile t        | group 1 stages ITS HALF of tile t + 1
//                                  barrier
//                                  phase 2: group 0 stages its half of t+1 | group 1 MFMAs of tile t
//                                  barrier
//   "its half" = A rows [64 g, 64 g + 64) and B rows [128 g, 128 g + 128): global -> registers (issued two phases
//   earlier, right after the previous tile's registers were drained) -> three bf16 planes -> the other LDS stage.
//   On every SIMD one wave issues 48 MFMAs back to back (1 536 pipe cycles) while its partner issues ~140 VALU / LDS /
//   VMEM instructions (~700 cycles).  Two LDS stages of 72 KB (A 24 KB + B 48 KB) = 144 KB: one workgroup per CU.
//
// LDS image per operand and plane: [k chunk of 8][row][8 bf16], the rows of chunk c rotated by 2 c rows (32 bytes):
// ds_read_b128 fragment reads (32 lanes = 32 consecutive rows of one chunk) and the ds_write_b64 staging writes
// (16 lanes = 2 rows x 8 float4 of a 128-byte line) are bank-conflict free (writes: banks are taken mod 128 bytes).
//
//   C[M, N] = epilogue( A[M, K] . B[N, K]^T ),  A, B fp32 row-major (k contiguous), split while staged.
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <type_traits>

#include "gi_common.h"
#include "gi_mfma.h"

typedef __bf16 gp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gp_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gp_u32x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ float gp_sink[512];                     // where out-of-range lanes of edge tiles store

// tools/gemm_lab.hip (-DGI_B3P_TRACE): shader-clock stamps of one workgroup's phases, [group][stamp]
#ifdef GI_B3P_TRACE
__device__ unsigned long long* gp_trace_buf;
#define GP_STAMP() do { if (trace_on && tr_n < 1000) gp_trace_buf[grp * 1024 + tr_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GP_STAMP() do {} while (0)
#endif
#ifdef GI_B3P_TRACE      // finer: stamp after the global loads have landed / after the LDS writes have drained
#define GP_WAITV() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GP_STAMP(); } while (0)
#define GP_WAITL() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); GP_STAMP(); } while (0)
#else
#define GP_WAITV() do {} while (0)
#define GP_WAITL() do {} while (0)
#endif

constexpr int GP_BM = 128, GP_BN = 256, GP_BK = 32;
constexpr int GP_CHA = GP_BM * 16, GP_CHB = GP_BN * 16;          // bytes of one k chunk (8 bf16) of all rows
constexpr int GP_PLA = 4 * GP_CHA, GP_PLB = 4 * GP_CHB;          // one plane of a 32-deep tile: 8 KB / 16 KB
constexpr int GP_A = 3 * GP_PLA, GP_B = 3 * GP_PLB;              // 24 KB / 48 KB
constexpr int GP_STAGE = GP_A + GP_B;                            // 72 KB

__device__ __forceinline__ unsigned gp_lds_a(int kc, int row) { return kc * GP_CHA + ((row * 16 + kc * 32) & (GP_CHA - 1)); }
__device__ __forceinline__ unsigned gp_lds_b(int kc, int row) { return kc * GP_CHB + ((row * 16 + kc * 32) & (GP_CHB - 1)); }
__device__ __forceinline__ unsigned gp_pk(float lo, float hi) {
    gp_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gp_bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float gp_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float gp_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ void gp_split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = gp_pk(x0, x1);
    const float r0 = x0 - gp_lo(p0), r1 = x1 - gp_hi(p0);
    p1 = gp_pk(r0, r1);
    p2 = gp_pk(r0 - gp_lo(p1), r1 - gp_hi(p1));
}

struct GpBatch {
    gi_gemm_params p[8];
    int start[9];
    int gx[8];
    int n, total, remap;
};

// EPI: 0 = epilogue from the run-time flags, 1 = bias + SELU (forward), 2 = * selu'(act) (dgrad)
template <int EPI>
__global__ __launch_bounds__(512, 2) void gi_b3p_kernel(const GpBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];        // 2 * GP_STAGE
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, w4 = wid & 3, wm = w4 >> 1, wn = w4 & 1, l31 = lane & 31, lhi = lane >> 5;
    const int gt = tid & 255;                        // thread within its group
#ifdef GI_B3P_TRACE
    const bool trace_on = gp_trace_buf && blockIdx.x == 3 && w4 == 0 && lane == 0;
    int tr_n = 0;
#endif

    // ---- tile (block-uniform) ----------------------------------------------------------------------
    int pi = 0;
    while (pi < b.n - 1 && (int)blockIdx.x >= b.start[pi + 1]) ++pi;
    const gi_gemm_params& p = b.p[pi];
    int local = blockIdx.x - b.start[pi];
    if (b.remap) {                                  // XCD-aware tile order (gi_gemm.hip), bijective
        const int tiles = b.start[pi + 1] - b.start[pi];
        const int q = tiles >> 3, r = tiles & 7, xcd = local & 7, j = local >> 3;
        local = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int gx = b.gx[pi];
    const int by = local / gx, bx = local - by * gx;
    const int m_end = p.m_dev ? min(p.M, *p.m_dev) : p.M;
    const int m0 = by * GP_BM, n0 = bx * GP_BN;
    if (m0 >= m_end) return;                        // (bounded launch: beyond the rows on the device)
    const int K = p.K;
    const int nk = (K + GP_BK - 1) / GP_BK, n_full = K / GP_BK;

    // ---- staging coordinates of this thread's share: 8 float4 per 32-deep row -> c8 = gt & 7, row (gt >> 3) + 32 i
    const int c8 = gt & 7, crow = gt >> 3;
    const int a_cmax = (p.lda >= ((K + 3) & ~3)) ? ((K + 3) & ~3) - 4 : K - 4;
    const int b_cmax = (p.ldb >= ((K + 3) & ~3)) ? ((K + 3) & ~3) - 4 : K - 4;
    unsigned a_off[2], b_off[4], a_w[2], b_w[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = 64 * grp + crow + 32 * i;
        const int row = min(m0 + rl, m_end - 1);
        a_off[i] = (unsigned)row * (unsigned)p.lda * 4u;
        a_w[i] = gp_lds_a(c8 >> 1, rl) + 8 * (c8 & 1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = 128 * grp + crow + 32 * i;
        const int row = min(n0 + rl, p.N - 1);
        b_off[i] = (unsigned)row * (unsigned)p.ldb * 4u;
        b_w[i] = GP_A + gp_lds_b(c8 >> 1, rl) + 8 * (c8 & 1);
    }
    v4f ra[2], rb[4];

    auto gload = [&](auto steady_c, int kt) __attribute__((always_inline)) {
        constexpr bool ST = decltype(steady_c)::value;
#if defined(GP_DBG) && (GP_DBG & 8)       // lab: no global loads inside the k loop
        if (kt > 1) return;
#endif
        const int k0 = kt * GP_BK;
        if (ST) {
            const char* abase = (const char*)p.A + (size_t)k0 * 4 + 16 * c8;
            const char* bbase = (const char*)p.B + (size_t)k0 * 4 + 16 * c8;
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = *(const v4f_u*)(abase + a_off[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = *(const v4f_u*)(bbase + b_off[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                ra[i] = gi_load4_raw((const float*)((const char*)p.A + a_off[i]), k0 + 4 * c8, a_cmax);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rb[i] = gi_load4_raw((const float*)((const char*)p.B + b_off[i]), k0 + 4 * c8, b_cmax);
        }
    };
    auto put = [&](unsigned char* S, unsigned w, int plane_bytes, v4f v) __attribute__((always_inline)) {
        gp_u32x2 w0, w1, w2;
        unsigned x0, x1, x2, y0, y1, y2;
#if defined(GP_DBG) && (GP_DBG & 2)       // lab: no split arithmetic
        x0 = __builtin_bit_cast(unsigned, v.x); x1 = __builtin_bit_cast(unsigned, v.y); x2 = x0 ^ x1;
        y0 = __builtin_bit_cast(unsigned, v.z); y1 = __builtin_bit_cast(unsigned, v.w); y2 = y0 ^ y1;
#else
        gp_split2(v.x, v.y, x0, x1, x2);
        gp_split2(v.z, v.w, y0, y1, y2);
#endif
        w0.x = x0; w0.y = y0; w1.x = x1; w1.y = y1; w2.x = x2; w2.y = y2;
#if defined(GP_DBG) && (GP_DBG & 4)       // lab: no LDS writes (keep the values alive)
        if ((x0 ^ x1 ^ x2 ^ y0 ^ y1 ^ y2) == 0x12345678u) *reinterpret_cast<gp_u32x2*>(S + w) = w0;
        return;
#endif
        *reinterpret_cast<gp_u32x2*>(S + w) = w0;
        *reinterpret_cast<gp_u32x2*>(S + plane_bytes + w) = w1;
        *reinterpret_cast<gp_u32x2*>(S + 2 * plane_bytes + w) = w2;
    };
    auto sstore = [&](auto steady_c, int kt) __attribute__((always_inline)) {
        constexpr bool ST = decltype(steady_c)::value;
        unsigned char* S = smem + (kt & 1) * GP_STAGE;
        const int k0 = kt * GP_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v4f v = ra[i];
            if (!ST) v = gi_fix4(v, k0 + 4 * c8, a_cmax, K, true);
            put(S, a_w[i], GP_PLA, v);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v4f v = rb[i];
            if (!ST) v = gi_fix4(v, k0 + 4 * c8, b_cmax, K, true);
            put(S, b_w[i], GP_PLB, v);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // One wave per SIMD computes at a time, so nothing hides an LDS round trip in front of an MFMA: ALL 24 fragment
    // reads of the k tile are issued up front (96 VGPRs), the 24 MFMAs of the first 16-deep half start when its 12
    // fragments have landed (s_waitcnt lgkmcnt(12)) and cover the flight of the second half's.  (Left to itself hipcc
    // reads just in time — ~10 exposed LDS waits per phase, measured: the phase took 2.6 k cycles for 1.5 k of MFMAs.)
    auto compute = [&](int kt) __attribute__((always_inline)) {
#if defined(GP_DBG) && (GP_DBG & 1)       // lab: no fragment reads, no MFMAs
        return;
#endif
        const unsigned char* As = smem + (kt & 1) * GP_STAGE;
        const unsigned char* Bs = As + GP_A;
        gp_bf16x8 af[2][2][3], bf[2][2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned oa = gp_lds_a(2 * s + lhi, wm * 64 + t * 32 + l31);
                const unsigned ob = gp_lds_b(2 * s + lhi, 128 * grp + wn * 64 + t * 32 + l31);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    af[s][t][pl] = *reinterpret_cast<const gp_bf16x8*>(As + pl * GP_PLA + oa);
                    bf[s][t][pl] = *reinterpret_cast<const gp_bf16x8*>(Bs + pl * GP_PLB + ob);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // smallest terms first; four independent accumulators between two MFMAs on the same one
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][t][TA[term]], bf[s][u][TB[term]],
                                                                            acc[t][u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- k loop ------------------------------------------------------------------------------------------
    // registers hold tile t + 1 while tile t is in LDS; "stage(t + 1)" = write it to the other LDS stage and start
    // the loads of tile t + 2.  STEADY iterations touch only tiles that are full in k (straight-line body).
    const std::true_type ST{};
    const std::false_type GEN{};
    if (n_full >= 1) gload(ST, 0); else gload(GEN, 0);
    sstore(GEN, 0);
    if (nk > 1) { if (n_full >= 2) gload(ST, 1); else gload(GEN, 1); }
    __syncthreads();
    int kt = 0;
    if (grp == 0) {
        for (; kt + 2 < n_full; ++kt) {                // tiles kt + 1 and kt + 2 are full in k
            GP_STAMP();
            compute(kt);
            GP_STAMP();
            __syncthreads();
            GP_STAMP();
            GP_WAITV();
            sstore(ST, kt + 1);
            GP_WAITL();
            gload(ST, kt + 2);
            GP_STAMP();
            __syncthreads();
        }
        for (; kt < nk; ++kt) {
            compute(kt);
            __syncthreads();
            if (kt + 1 < nk) sstore(GEN, kt + 1);
            if (kt + 2 < nk) gload(GEN, kt + 2);
            __syncthreads();
        }
    } else {
        for (; kt + 2 < n_full; ++kt) {
            GP_STAMP();
            GP_WAITV();
            sstore(ST, kt + 1);
            GP_WAITL();
            gload(ST, kt + 2);
            GP_STAMP();
            __syncthreads();
            GP_STAMP();
            compute(kt);
            GP_STAMP();
            __syncthreads();
        }
        for (; kt < nk; ++kt) {
            if (kt + 1 < nk) sstore(GEN, kt + 1);
            if (kt + 2 < nk) gload(GEN, kt + 2);
            __syncthreads();
            compute(kt);
            __syncthreads();
        }
    }

    // ---- epilogue (C/D layout of a 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) --
    const int flags = EPI == 1 ? (GI_EPI_BIAS | GI_EPI_SELU) : (EPI == 2 ? GI_EPI_DSELU : p.flags);
    const bool need_act = (flags & (GI_EPI_DSELU | GI_EPI_MULACT)) != 0;
    const bool need_c = (flags & GI_EPI_ACCUM) != 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = n0 + 128 * grp + wn * 64 + u * 32 + l31;
            const bool col_ok = col < p.N;
            const int colc = col_ok ? col : p.N - 1;
            const int row0 = m0 + wm * 64 + t * 32 + 4 * lhi;
            const float bv = (flags & GI_EPI_BIAS) ? p.bias[colc] : 0.f;
            float av[16], cv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {                        // every load of the block before the first store
                const int row = min(row0 + 8 * (r >> 2) + (r & 3), m_end - 1);
                if (need_act) av[r] = p.act[(long long)row * p.ldact + colc];
                if (need_c) cv[r] = p.C[(long long)row * p.ldc + colc];
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[t][u][r] + bv;
                if (flags & GI_EPI_SELU) x = gi_selu(x);
                if (flags & GI_EPI_DSELU) x *= gi_selu_grad(av[r]);
                if (flags & GI_EPI_MULACT) x *= av[r];
                if (flags & GI_EPI_ACCUM) x += cv[r];
                v[r] = x;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 8 * (r >> 2) + (r & 3);
                float* dst = (col_ok & (row < m_end)) ? p.C + (long long)row * p.ldc + col : gp_sink + tid;
                *dst = v[r];
            }
        }
    }
}

int g_b3p_enabled = -1;
bool g_b3p_attr_set[3] = {false, false, false};

}  // namespace

// Process-wide switch (measurement aid): route eligible forward / dgrad GI_GEMM_BF3 launches to the ping-pong
// kernel of this file (1; environment GI_B3P) or not (0).  on < 0 only queries; returns the previous value.
extern "C" int gi_b3p_enable(int on) {
    if (g_b3p_enabled < 0) {
        const char* e = getenv("GI_B3P");
        g_b3p_enabled = e ? (atoi(e) != 0) : 1;
    }
    const int prev = g_b3p_enabled;
    if (on >= 0) g_b3p_enabled = on ? 1 : 0;
    return prev;
}

bool gi_b3p_eligible(const gi_gemm_params* probs, int n) {
    if (!gi_b3p_enable(-1)) return false;
    for (int i = 0; i < n; ++i) {
        const gi_gemm_params& p = probs[i];
        if (!(p.flags & GI_GEMM_BF3) || (p.flags & GI_GEMM_BF3A) || !(p.flags & GI_GEMM_BF3B_F32)) return false;
        if (p.a_major || p.b_major || p.a_idx || p.b_idx || p.k_dev || p.ngroups || p.nsplit != 1) return false;
        if (p.flags & GI_GEMM_SPLITK) return false;
    }
    return true;
}

int gi_b3p_launch(const gi_gemm_params* probs, int n, void* stream) {
    GpBatch b;
    memset(&b, 0, sizeof(b));
    double flops = 0;
    int total = 0, k = 0, epi = -1;
    for (int i = 0; i < n; ++i) {
        const gi_gemm_params& p = probs[i];
        if (!p.A || !p.B || !p.C || p.M < 0 || p.N <= 0 || p.K <= 0) return GI_EINVAL;
        const int f = p.flags & ~(GI_GEMM_BF3 | GI_GEMM_BF3B_F32);
        if (f & ~(GI_EPI_BIAS | GI_EPI_SELU | GI_EPI_DSELU | GI_EPI_ACCUM | GI_EPI_MULACT)) return GI_EINVAL;
        if ((f & GI_EPI_BIAS) && !p.bias) return GI_EINVAL;
        if ((f & (GI_EPI_DSELU | GI_EPI_MULACT)) && !p.act) return GI_EINVAL;
        if (p.lda < p.K || p.ldb < p.K || (p.K < 4 && (p.lda < 4 || p.ldb < 4))) return GI_EINVAL;
        const long long lim = 0xffffffffLL / 4;
        if ((long long)p.M * p.lda > lim || (long long)p.N * p.ldb > lim) return GI_ELIMIT;
        const int e = f == (GI_EPI_BIAS | GI_EPI_SELU) ? 1 : (f == GI_EPI_DSELU ? 2 : 0);
        epi = (epi < 0 || epi == e) ? e : 0;
        if (p.M == 0) continue;
        b.p[k] = p; b.p[k].flags = f;
        b.gx[k] = gi_cdiv(p.N, GP_BN);
        b.start[k] = total;
        total += b.gx[k] * gi_cdiv(p.M, GP_BM);
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
        ++k;
    }
    if (k == 0) return 0;
    b.start[k] = total; b.n = k; b.total = total;
    bool bounded = false;
    for (int i = 0; i < k; ++i) bounded |= b.p[i].m_dev != nullptr;
    b.remap = (total >= 512 && !bounded) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    void (*fn)(const GpBatch) = epi == 1 ? gi_b3p_kernel<1> : (epi == 2 ? gi_b3p_kernel<2> : gi_b3p_kernel<0>);
    if (!g_b3p_attr_set[epi]) {                     // 144 KB of dynamic LDS needs the opt-in
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GP_STAGE) != hipSuccess)
            return (int)hipGetLastError();
        g_b3p_attr_set[epi] = true;
    }
    GiProfScope prof(st, GI_PROF_GEMM, flops);
    gi_gemm_log_launch((b.p[0].flags & GI_EPI_BIAS) ? "p0" : "p1", b.p, k, total, flops);
    hipLaunchKernelGGL(fn, dim3(total), dim3(512), 2 * GP_STAGE, st, b);
    return gi_launch_status();
}
