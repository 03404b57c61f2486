// bf16x3 GEMM, software-pipelined (gfx950): the staging work of k tile t + 1 — global loads, the three-way bf16
// split, the LDS writes, the fragment reads — is INTERLEAVED with the 48 MFMAs of k tile t inside every wave.
//
// Why (round 4; rocprofv3 PMC passes, phase traces and tools/mfma_fill.hip under profiles/r04/):
//   * the round-3 kernel (and the first 32-deep-tile kernel, gi_gemm_b3v.hip) keep the matrix pipe 35 % busy: the
//     three 4-wave workgroups of a CU start together, run the same phases and fall into lockstep — every wave of a
//     SIMD queues on the matrix pipe at once, then all of them split / write LDS at once (~230 non-MFMA
//     instructions per wave and k tile at ~5 cycles each) while the pipe idles;
//   * giving the two jobs to two wave groups that alternate (a "ping-pong" workgroup, the first version of this
//     file) is no better: a wave that issues MFMAs back to back starves its SIMD partner's VALU stream (the staging
//     phase took 2.7 k cycles beside a computing partner, 1.4 k alone; the MFMA phase 2.1 k instead of 1.5 k);
//   * but INSIDE one instruction stream up to 5 VALU instructions per v_mfma_f32_32x32x16_bf16 are free, with one
//     or two waves per SIMD (33-34 matrix-pipe cycles per MFMA; 6 fillers 36, 8 fillers 44) — except v_pk_add_f32,
//     which is not hidden at all (+12 cycles each): this file is built with -fno-slp-vectorize.
//   The split is ~11 VALU per pair of values: a 128 x 256 x 32 tile on 8 waves needs 132 VALU + 24 ds_read_b128 +
//   18 LDS writes + 6..24 loads per wave for its 48 MFMAs = 3.8 per MFMA.
//
// Structure: 512 threads = 2 x 4 waves of 64 x 64; block tile 128 x 256 x 32; two LDS stages of 72 KB (A 24 KB + B
// 48 KB; one workgroup per CU); ONE barrier per k tile.  Iteration t (tile t complete in stage t & 1, the fp32
// values of tile t + 1 in registers, its MFMA stream skewed by 8 so that no MFMA ever waits for an LDS round trip):
//     group  0- 1: last 8 MFMAs of tile t - 1 (second-half fragments)   | ds_read first-half fragments of tile t
//     group  2- 7: 24 MFMAs of tile t, first 16-deep half               | ds_read second-half fragments
//     group  8-11: 16 MFMAs of tile t, second half
//     every group: + 1/12 of the staging of tile t + 1 (one pair-split, LDS writes, then the loads of tile t + 2)
//     s_waitcnt lgkmcnt(0); s_barrier
// The groups are fenced with sched_barrier(0) and ordered inside with sched_group_barrier (1 MFMA : 3 VALU : LDS).
//
// Operand forms: contiguous-k fp32 rows (forward / dgrad A, forward B) or reduction-major fp32 [k][rows] (both
// operands of a weight gradient dZ^T [X | 1], W of a dgrad as stored), transposed in registers on the way into LDS
// (a wave loads 8 reduction rows of one k chunk, a lane one column of each: dword loads, 256 contiguous bytes per
// wave instruction).  LDS image per operand and plane: [k chunk of 8][row][8 bf16], rows of chunk c rotated by 2 c
// rows: fragment reads and all staging writes are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0).
//
//   C[M, N] = epilogue( sum_k A(m, k) B(n, k) ); split-K slabs (plain or grouped) for the weight-gradient layout.
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <type_traits>

#include "gi_common.h"
#include "gi_mfma.h"
#include "gi_x2.h"

typedef __bf16 gp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gp_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gp_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gp_u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ float gp_sink[512];                     // where out-of-range lanes of edge tiles store

// tools/b3p_lab.hip (-DGI_B3P_TRACE): shader-clock stamp at every iteration start of one workgroup's waves 0 and 4
#ifdef GI_B3P_TRACE
__device__ unsigned long long* gp_trace_buf;
#define GP_STAMP() do { if (trace_on && tr_n < 1000) gp_trace_buf[(wid >> 2) * 1024 + tr_n++] = __builtin_amdgcn_s_memtime(); } while (0)
// ... and of every workgroup: [2048 + 4 block + {0: start, 1: k loop start, 2: k loop end, 3: end}]
#define GP_WG_STAMP(i) do { if (gp_trace_buf && threadIdx.x == 0) gp_trace_buf[2048 + 4 * tile_id + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GP_STAMP() do {} while (0)
#define GP_WG_STAMP(i) do {} while (0)
#endif

constexpr int GP_BM = 128, GP_BN = 256, GP_BK = 32;
constexpr int GP_CHA = GP_BM * 16, GP_CHB = GP_BN * 16;          // bytes of one k chunk (8 bf16) of all rows
constexpr int GP_PLA = 4 * GP_CHA, GP_PLB = 4 * GP_CHB;          // one plane of a 32-deep tile: 8 KB / 16 KB
constexpr int GP_A = 3 * GP_PLA, GP_B = 3 * GP_PLB;              // 24 KB / 48 KB (three bf16 planes)
constexpr int GP_STAGE = GP_A + GP_B;                            // 72 KB
constexpr int GP_STAGE_X2 = 2 * (GP_PLA + GP_PLB);               // 48 KB (two fp16 planes, GI_GEMM_X2)

__device__ __forceinline__ unsigned gp_lds_a(int kc, int row) { return kc * GP_CHA + ((row * 16 + kc * 32) & (GP_CHA - 1)); }
__device__ __forceinline__ unsigned gp_lds_b(int kc, int row) { return kc * GP_CHB + ((row * 16 + kc * 32) & (GP_CHB - 1)); }
__device__ __forceinline__ unsigned gp_pk(float lo, float hi) {
    gp_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gp_bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float gp_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float gp_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// two fp32 values -> their three bf16 planes, packed pairwise (low half = x0)
__device__ __forceinline__ void gp_split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = gp_pk(x0, x1);
    const float r0 = x0 - gp_lo(p0), r1 = x1 - gp_hi(p0);
    p1 = gp_pk(r0, r1);
    p2 = gp_pk(r0 - gp_lo(p1), r1 - gp_hi(p1));
}

struct GpBatch {
    gi_gemm_params p[8];
    int start[9];
    int gx[8], gy[8];
    int n, total, remap;
};

#define GP_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// order inside a group of 4 MFMAs + its fillers: MFMA, 3 VALU, up to 3 LDS ops, ... then the loads, then a fence
#define GP_GROUP_ORDER()                                                                                          \
    do {                                                                                                          \
        GP_SGB(0x008, 1); GP_SGB(0x002, 3); GP_SGB(0x080, 3); GP_SGB(0x008, 1); GP_SGB(0x002, 3); GP_SGB(0x080, 3); \
        GP_SGB(0x008, 1); GP_SGB(0x002, 3); GP_SGB(0x080, 3); GP_SGB(0x008, 1); GP_SGB(0x002, 3); GP_SGB(0x080, 3); \
        GP_SGB(0x020, 8);                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

// EPI: 0 = epilogue from the run-time flags, 1 = bias + SELU (forward), 2 = * selu'(act) (dgrad), 3 = plain store
// (weight-gradient slabs).
// X2 (GI_GEMM_X2): operands as two scaled fp16 values each (gi_x2.h) — two LDS planes, three f16 MFMA products per
// fp32 product (24 MFMAs per wave and k tile in 6 groups instead of 48 in 12), scales from a_amax / b_amax.
// BIDX (weight-gradient layout only): the reduction rows of B are gathered through p.b_idx (the first layer of a message
// stack reads h[u_src]) — the index of a reduction row is wave-uniform, i.e. scalar loads ahead of the row loads.
template <bool AM, bool BMJ, int EPI, bool X2, bool BIDX = false>
__device__ __forceinline__ void gp_tile(const GpBatch& b, const int tile_id, unsigned char* const smem) {
    constexpr int NP = X2 ? 2 : 3;
    constexpr int A_BYTES = NP * GP_PLA, STAGE = NP * (GP_PLA + GP_PLB);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3, l31 = lane & 31, lhi = lane >> 5;
#ifdef GI_B3P_TRACE
    const bool trace_on = gp_trace_buf && tile_id == 3 && (wid & 3) == 0 && lane == 0;
    int tr_n = 0;
#endif

    GP_WG_STAMP(0);
    // ---- tile (block-uniform) ----------------------------------------------------------------------
    int pi = 0;
    while (pi < b.n - 1 && tile_id >= b.start[pi + 1]) ++pi;
    const gi_gemm_params& p = b.p[pi];
    int local = tile_id - b.start[pi];
    if (b.remap) {                                  // XCD-aware tile order (gi_gemm.hip), bijective
        const int tiles = b.start[pi + 1] - b.start[pi];
        const int q = tiles >> 3, r = tiles & 7, xcd = local & 7, j = local >> 3;
        local = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int gx = b.gx[pi], gxy = gx * b.gy[pi];
    const int bz = local / gxy;
    const int rem = local - bz * gxy;
    const int by = rem / gx, bx = rem - by * gx;
    const int m_end = p.m_dev ? min(p.M, *p.m_dev) : p.M;
    const int m0 = by * GP_BM, n0 = bx * GP_BN;
    if (m0 >= m_end) return;                        // (bounded launch: beyond the rows on the device)
    int kb = 0, ke = p.K;
    float* Cp = p.C;
    if (p.flags & GI_GEMM_SPLITK) {                 // reduction range of this slab
        int g = 0, s = bz, nsp = p.nsplit;
        if (p.ngroups) {
            while (g < p.ngroups - 1 && s >= p.gsplit[g]) { s -= p.gsplit[g]; ++g; }
            nsp = p.gsplit[g];
            Cp = p.Cg[g];
            kb = p.grp_off[g]; ke = p.grp_off[g + 1];
        }
        const int chunk = (((ke - kb + nsp - 1) / nsp) + 31) & ~31;
        kb += s * chunk;
        ke = min(kb + chunk, ke);
        Cp += (long long)s * p.c_split_stride;
    }
    const int nk = ke > kb ? (ke - kb + GP_BK - 1) / GP_BK : 0;
    const int n_full = ke > kb ? (ke - kb) / GP_BK : 0;
    float sa = 1.f, ia = 1.f, sb = 1.f, ib = 1.f;               // fp16x2: per-tensor power-of-two scales
    if (X2) { gx_scale(gx_amax_read(p.a_amax), sa, ia); gx_scale(gx_amax_read(p.b_amax), sb, ib); }

    // ---- staging coordinates -----------------------------------------------------------------------
    // contig operand: 8 float4 per 32-deep row -> c8 = tid & 7, rows (tid >> 3) + 64 i  (A: 2, B: 4 float4 per thread)
    // major operand : wave = (k chunk w & 3, column half w >> 2), lane = one column (A) / two columns 64 apart (B)
    const int c8 = tid & 7, crow = tid >> 3;
    const int kcw = wid & 3, halfw = wid >> 2;
    const int a_cols = p.M;
    const int b_cols = p.ones_col >= 0 ? p.ones_col : p.N;
    const int a_cmax = (p.lda >= ((p.K + 3) & ~3)) ? ((p.K + 3) & ~3) - 4 : p.K - 4;
    const int b_cmax = (p.ldb >= ((p.K + 3) & ~3)) ? ((p.K + 3) & ~3) - 4 : p.K - 4;
    unsigned a_off[2], b_off[4], a_w[2], b_w[4];
    if (!AM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rl = crow + 64 * i;
            a_off[i] = (unsigned)min(m0 + rl, m_end - 1) * (unsigned)p.lda * 4u;
            a_w[i] = gp_lds_a(c8 >> 1, rl) + 8 * (c8 & 1);
        }
    } else {
        a_off[0] = 4u * (unsigned)min(m0 + 64 * halfw + lane, a_cols - 1);
        a_w[0] = gp_lds_a(kcw, 64 * halfw + lane);
    }
    if (!BMJ) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = crow + 64 * i;
            b_off[i] = (unsigned)min(n0 + rl, p.N - 1) * (unsigned)p.ldb * 4u;
            b_w[i] = A_BYTES + gp_lds_b(c8 >> 1, rl) + 8 * (c8 & 1);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            b_off[i] = 4u * (unsigned)min(n0 + 128 * halfw + lane + 64 * i, b_cols - 1);
            b_w[i] = A_BYTES + gp_lds_b(kcw, 128 * halfw + lane + 64 * i);
        }
    }
    // the ones column of a major B (bias gradient): which of the lane's two columns, if any
    const bool ones0 = BMJ && p.ones_col >= 0 && n0 + 128 * halfw + lane == p.ones_col;
    const bool ones1 = BMJ && p.ones_col >= 0 && n0 + 128 * halfw + lane + 64 == p.ones_col;

    // ---- the register stage: the 24 fp32 values of the next k tile this thread stages --------------------------
    v4f ra[2], rb[4];                              // contig: float4 along k
    float xa[8], xb[2][8];                         // major: 8 consecutive k of one column
    auto load_a_contig = [&](auto st, int kt, int i) __attribute__((always_inline)) {
        const int k0 = kb + kt * GP_BK;
        if (decltype(st)::value) ra[i] = *(const v4f_u*)((const char*)p.A + (size_t)k0 * 4 + 16 * c8 + a_off[i]);
        else ra[i] = gi_load4_raw((const float*)((const char*)p.A + a_off[i]), k0 + 4 * c8, a_cmax);
    };
    auto load_b_contig = [&](auto st, int kt, int i) __attribute__((always_inline)) {
        const int k0 = kb + kt * GP_BK;
        if (decltype(st)::value) rb[i] = *(const v4f_u*)((const char*)p.B + (size_t)k0 * 4 + 16 * c8 + b_off[i]);
        else rb[i] = gi_load4_raw((const float*)((const char*)p.B + b_off[i]), k0 + 4 * c8, b_cmax);
    };
    auto load_a_major = [&](auto st, int kt) __attribute__((always_inline)) {
        const int kr = kb + kt * GP_BK + 8 * kcw;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = decltype(st)::value ? kr + j : min(kr + j, ke - 1);
            xa[j] = *(const float*)((const char*)p.A + (size_t)row * (size_t)p.lda * 4 + a_off[0]);
        }
    };
    auto load_b_major = [&](auto st, int kt, int i) __attribute__((always_inline)) {
        const int kr = kb + kt * GP_BK + 8 * kcw;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int row = decltype(st)::value ? kr + j : min(kr + j, ke - 1);
            if (BIDX) row = __builtin_amdgcn_readfirstlane(p.b_idx[row]);
            xb[i][j] = *(const float*)((const char*)p.B + (size_t)row * (size_t)p.ldb * 4 + b_off[i]);
        }
    };
    // Staging is cut into 12 chunks (one pair-split each); after the chunk that consumes the last values of a
    // register group, that group's loads for the tile after may start:
    //   contig A + contig B: chunk c = half c & 1 of float4 c >> 1 (0, 1: A; 2..5: B)
    //   major A + major B  : chunk c = pair c & 3 of column c >> 2 (0: A; 1, 2: B)
    //   contig A + major B : chunks 0-3 as the first form (A), 4-11 as the second (B columns)
    auto gload_after = [&](auto st, int kt, int c) __attribute__((always_inline)) {
        if (!AM && !BMJ) {
            if (c & 1) { if ((c >> 1) < 2) load_a_contig(st, kt, c >> 1); else load_b_contig(st, kt, (c >> 1) - 2); }
        } else if (AM) {
            if (c == 3) load_a_major(st, kt);
            if (c == 7) load_b_major(st, kt, 0);
            if (c == 11) load_b_major(st, kt, 1);
        } else {
            if (c == 1) load_a_contig(st, kt, 0);
            if (c == 3) load_a_contig(st, kt, 1);
            if (c == 7) load_b_major(st, kt, 0);
            if (c == 11) load_b_major(st, kt, 1);
        }
    };
    auto gload_all = [&](auto st, int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 12; ++c) gload_after(st, kt, c);
    };

    // ---- staging chunk c (0..11) of k tile kt: one pair-split, and the LDS writes it completes ----------------
    unsigned q0[4], q1[4], q2[4];                  // planes of the 8-byte / 16-byte piece being assembled
    float rowmax[2] = {0.f, 0.f};                  // fp16x2 guard (gi_gemm_params.x2_guard): max |a| staged of the thread's two contig A rows
    auto stage_chunk = [&](auto st, int kt, int c) __attribute__((always_inline)) {
        constexpr bool ST = decltype(st)::value;
        unsigned char* S = smem + (kt & 1) * STAGE;
        const int k0 = kb + kt * GP_BK;
        const bool contig_chunk = (!AM && !BMJ) || (!AM && c < 4);
        if (contig_chunk) {
            const int u = c >> 1, h = c & 1;
            const bool isA = u < 2;
            v4f v = isA ? ra[u] : rb[u - 2];
            if (!ST && h == 0) {                   // zero fill / lane shift of a partial k tile, once per float4
                v = gi_fix4(v, k0 + 4 * c8, isA ? a_cmax : b_cmax, p.K, true);
                if (isA) ra[u] = v; else rb[u - 2] = v;
            }
            if (X2 && !AM && isA && h == 0)
                rowmax[u & 1] = fmaxf(fmaxf(rowmax[u & 1], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            if (X2) gx_split2(h ? v.z : v.x, h ? v.w : v.y, isA ? sa : sb, q0[h], q1[h]);
            else gp_split2(h ? v.z : v.x, h ? v.w : v.y, q0[h], q1[h], q2[h]);
            if (h == 1) {
                const unsigned w = isA ? a_w[u] : b_w[u - 2];
                const int pb = isA ? GP_PLA : GP_PLB;
                gp_u32x2 w0 = {q0[0], q0[1]}, w1 = {q1[0], q1[1]};
                *reinterpret_cast<gp_u32x2*>(S + w) = w0;
                *reinterpret_cast<gp_u32x2*>(S + pb + w) = w1;
                if (!X2) { gp_u32x2 w2 = {q2[0], q2[1]}; *reinterpret_cast<gp_u32x2*>(S + 2 * pb + w) = w2; }
            }
        } else {
            const int col = c >> 2, h = c & 3;     // 0: the A column (major A only), 1, 2: the B columns
            const bool isA = AM && col == 0;
            float* x = isA ? xa : xb[col - 1];
            if (h == 0) {
                const int kr = k0 + 8 * kcw;
                const bool one = !isA && (col == 1 ? ones0 : ones1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = one ? (X2 ? ib : 1.f) : x[j];     // fp16x2: 1.0 AFTER the scale (the epilogue leaves that column's 1 / sb out):
                                                                // the tensor's scale may take a plain 1.0 beyond fp16's range
                    if (!ST) v = (kr + j < ke) ? v : 0.f;          // zero fill along the reduction (wave-uniform)
                    x[j] = v;
                }
            }
            if (X2) gx_split2(x[2 * h], x[2 * h + 1], isA ? sa : sb, q0[h], q1[h]);
            else gp_split2(x[2 * h], x[2 * h + 1], q0[h], q1[h], q2[h]);
            if (h == 3) {
                const unsigned w = isA ? a_w[0] : b_w[col - 1];
                const int pb = isA ? GP_PLA : GP_PLB;
                gp_u32x4 w0 = {q0[0], q0[1], q0[2], q0[3]}, w1 = {q1[0], q1[1], q1[2], q1[3]};
                *reinterpret_cast<gp_u32x4*>(S + w) = w0;
                *reinterpret_cast<gp_u32x4*>(S + pb + w) = w1;
                if (!X2) { gp_u32x4 w2 = {q2[0], q2[1], q2[2], q2[3]}; *reinterpret_cast<gp_u32x4*>(S + 2 * pb + w) = w2; }
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    gp_bf16x8 af0[2][3], bf0[2][3], af1[2][3], bf1[2][3];       // fragments of the two 16-deep halves
    auto read_frags = [&](int kt, int s, gp_bf16x8 (&af)[2][3], gp_bf16x8 (&bf)[2][3]) __attribute__((always_inline)) {
        const unsigned char* As = smem + (kt & 1) * STAGE;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned oa = gp_lds_a(2 * s + lhi, wm * 64 + t * 32 + l31);
            const unsigned ob = gp_lds_b(2 * s + lhi, wn * 64 + t * 32 + l31);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                af[t][pl] = *reinterpret_cast<const gp_bf16x8*>(As + pl * GP_PLA + oa);
                bf[t][pl] = *reinterpret_cast<const gp_bf16x8*>(Bs + pl * GP_PLB + ob);
            }
        }
    };
    // one term of the six-product sum for the wave's four accumulators (smallest terms first)
    auto mfma4 = [&](const gp_bf16x8 (&af)[2][3], const gp_bf16x8 (&bf)[2][3], int term) __attribute__((always_inline)) {
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
        constexpr int XA[3] = {1, 0, 0}, XB[3] = {0, 1, 0};      // fp16x2: a2 b1 + a1 b2 + a1 b1
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (X2)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gx_f16x8, af[t][XA[term]]),
                                                                       __builtin_bit_cast(gx_f16x8, bf[u][XB[term]]),
                                                                       acc[t][u], 0, 0, 0);
                else
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][TA[term]], bf[u][TB[term]], acc[t][u], 0, 0, 0);
            }
    };

    // ---- one iteration: the 12 groups of k tile kt ------------------------------------------------------------
    // FIRST: no MFMAs left over from a previous tile.  STAGE: tile kt + 1 exists (its values are in the registers):
    // stage it; LOAD: ... and start the loads of tile kt + 2.  ST: tiles kt + 1 and kt + 2 are full in k.
    auto iteration = [&](auto first_c, auto stage_c, auto load_c, auto st, int kt) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, STAGE = decltype(stage_c)::value, LOAD = decltype(load_c)::value;
        GP_STAMP();
        constexpr int NT = X2 ? 3 : 6;                   // products per 16-deep half; one group of 4 MFMAs per product
        constexpr int SKEW = X2 ? 1 : 2;                 // groups of the previous tile that run first (LDS latency cover)
        constexpr int CPG = 12 / (2 * NT);               // staging chunks per group
#pragma unroll
        for (int g = 0; g < 2 * NT; ++g) {
            if (g == 0) read_frags(kt, 0, af0, bf0);
            if (g == SKEW) read_frags(kt, 1, af1, bf1);
            if (g < SKEW) { if (!FIRST) mfma4(af1, bf1, NT - SKEW + g); }
            else if (g < SKEW + NT) mfma4(af0, bf0, g - SKEW);
            else mfma4(af1, bf1, g - SKEW - NT);
            if (STAGE) {
#pragma unroll
                for (int c = g * CPG; c < (g + 1) * CPG; ++c) {
                    stage_chunk(st, kt + 1, c);
                    if (LOAD) gload_after(st, kt + 2, c);
                }
            }
            GP_GROUP_ORDER();
        }
        __syncthreads();
    };

    // ---- k loop ------------------------------------------------------------------------------------------------
    const std::true_type T{};
    const std::false_type F{};
    if (nk > 0) {
        // prologue: tile 0 -> stage 0 (no MFMAs to hide behind), tile 1 -> registers
        if (n_full >= 1) gload_all(T, 0); else gload_all(F, 0);
#pragma unroll
        for (int c = 0; c < 12; ++c) stage_chunk(F, 0, c);
        if (nk > 1) { if (n_full >= 2) gload_all(T, 1); else gload_all(F, 1); }
        __syncthreads();
        GP_WG_STAMP(1);
        int kt = 0;
        if (kt + 2 < n_full) iteration(T, T, T, T, kt);                       // first iteration
        else if (nk == 1) iteration(T, F, F, F, kt);
        else if (nk == 2) iteration(T, T, F, F, kt);
        else iteration(T, T, T, F, kt);
        ++kt;
        for (; kt + 2 < n_full; ++kt) iteration(F, T, T, T, kt);             // steady: tiles kt + 1, kt + 2 full in k
        for (; kt + 2 < nk; ++kt) iteration(F, T, T, F, kt);
        for (; kt + 1 < nk; ++kt) iteration(F, T, F, F, kt);
        for (; kt < nk; ++kt) iteration(F, F, F, F, kt);
        if (X2) mfma4(af1, bf1, 2);                                            // the skewed tail of the last tile
        else { mfma4(af1, bf1, 4); mfma4(af1, bf1, 5); }
    }
    GP_WG_STAMP(2);

    // ---- fp16x2 dynamic-range guard (forward / dgrad layouts; see gi_gemm_bf3.hip): rows of A whose largest scaled
    // magnitude is below 2^-11 keep fewer than ~14 bits.  Once per launch: the workgroups of the first column tile.
    if (X2 && !AM && p.x2_guard && bx == 0 && !(p.flags & GI_GEMM_SPLITK)) {
        int n_low = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float m = rowmax[i];
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            m = fmaxf(m, __shfl_xor(m, 4));                     // the eight lanes (k chunks) of a row
            const bool real_row = m0 + crow + 64 * i < m_end;
            n_low += (c8 == 0 && real_row && m > 0.f && m * sa < 0x1p-11f) ? 1 : 0;
        }
        if (n_low) {
            atomicAdd(p.x2_guard, n_low);
            if (p.x2_guard_host) *reinterpret_cast<volatile int*>(p.x2_guard_host) = 1;
        }
    }

    // ---- epilogue (C/D layout of a 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) --
    // One workgroup per CU: nothing else runs while a tile's 64 outputs per thread are finished, so this part is
    // written for instruction count — every scalar of the problem in a local (hipcc re-read p.ldc from the kernel
    // arguments per element: 64 scalar round trips), SELU without its branch, addresses as base + constant * ldc.
    const int flags = EPI == 1 ? (GI_EPI_BIAS | GI_EPI_SELU) : (EPI == 2 ? GI_EPI_DSELU : (EPI == 3 ? 0 : (p.flags & ~GI_GEMM_SPLITK)));
    const bool need_act = (flags & (GI_EPI_DSELU | GI_EPI_MULACT)) != 0;
    const bool need_c = (flags & GI_EPI_ACCUM) != 0;
    const int ldc = p.ldc, ldact = p.ldact, Ncols = p.N;
    const float* const actp = p.act;
    const float* const biasp = p.bias;
    float* const sink = gp_sink + tid;
    float amax = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = n0 + wn * 64 + u * 32 + l31;
            const bool col_ok = col < Ncols;
            const int colc = col_ok ? col : Ncols - 1;
            const int row0 = m0 + wm * 64 + t * 32 + 4 * lhi;
            const float bv = (flags & GI_EPI_BIAS) ? biasp[colc] : 0.f;
            const float ibc = (X2 && BMJ && col == p.ones_col) ? 1.f : ib;   // (the ones column was staged as 1 / sb)
            const int rows_left = m_end - row0;                   // rows row0 .. row0 + rows_left - 1 exist
            float* const cbase = Cp + (long long)row0 * ldc + col;
            const float* const abase = need_act ? actp + (long long)min(row0, m_end - 1) * ldact + colc : nullptr;
            const float* const cin = Cp + (long long)min(row0, m_end - 1) * ldc + colc;
            float av[16], cv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {                        // every load of the block before the first store
                const int dr = 8 * (r >> 2) + (r & 3);
                const int drc = min(dr, max(rows_left - 1, 0));   // (clamped: any readable row)
                if (need_act) av[r] = abase[(long long)drc * ldact];
                if (need_c) cv[r] = cin[(long long)drc * ldc];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = 8 * (r >> 2) + (r & 3);
                float x = X2 ? (acc[t][u][r] * ia) * ibc + bv : acc[t][u][r] + bv;
                if (flags & GI_EPI_SELU) {                        // scale * (max(x, 0) + alpha * (exp(min(x, 0)) - 1)): no branch
                    const float e = gi_exp_nonpos(fminf(x, 0.f));
                    x = GI_SELU_SCALE * (fmaxf(x, 0.f) + GI_SELU_ALPHA * (e - 1.f));
                }
                if (flags & GI_EPI_DSELU) x *= gi_selu_grad(av[r]);
                if (flags & GI_EPI_MULACT) x *= av[r];
                if (flags & GI_EPI_ACCUM) x += cv[r];
                const bool ok = col_ok & (dr < rows_left);
                float* dst = ok ? cbase + (long long)dr * ldc : sink;
                *dst = x;
                amax = fmaxf(amax, ok ? fabsf(x) : 0.f);
            }
        }
    }
    if (p.c_amax) gx_amax_publish(amax, p.c_amax);       // for the fp16x2 launches that read this tensor next
    GP_WG_STAMP(3);
}

// One workgroup per CU and 26 k cycles of prologue + epilogue per tile: a launch with more tiles than CUs runs as a
// TILE STREAM — as many workgroups as the device has CUs, each walking tiles id, id + grid, ... (longest reductions
// first: gi_gemm_batch's order) — so no CU waits for a 512-thread / 144 KB workgroup to be torn down and set up
// between two tiles.
template <bool AM, bool BMJ, int EPI, bool X2 = false, bool BIDX = false>
__global__ __launch_bounds__(512, 2) void gi_b3p_kernel(const GpBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];        // 2 stages
    for (int tile = blockIdx.x; tile < b.total; tile += gridDim.x) gp_tile<AM, BMJ, EPI, X2, BIDX>(b, tile, smem);
}

int g_b3p_enabled = -1, g_b3p_stream_cus = -1;
bool g_b3p_attr_set[2][4][4] = {};

}  // namespace

// Process-wide switch (measurement aid): route eligible GI_GEMM_BF3 launches to the software-pipelined kernel of
// this file (1; environment GI_B3P) or not (0).  on < 0 only queries; returns the previous value.
extern "C" int gi_b3p_enable(int on) {
    if (g_b3p_enabled < 0) {
        const char* e = getenv("GI_B3P");
        g_b3p_enabled = e ? (atoi(e) != 0) : 1;
    }
    const int prev = g_b3p_enabled;
    if (on >= 0) g_b3p_enabled = on ? 1 : 0;
    return prev;
}

// fp32 operands only (no pre-split images), no row gathers; layouts: contig/contig, contig/major, major/major
bool gi_b3p_eligible(const gi_gemm_params* probs, int n) {
    if (!gi_b3p_enable(-1)) return false;
    long long tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (long long)gi_cdiv(probs[i].M, GP_BM) * gi_cdiv(probs[i].N, GP_BN);
    // forward / dgrad launches: one 128 x 256 workgroup per CU pays 17 k cycles of prologue + epilogue per tile and
    // quantises badly below ~2.5 tiles per CU (measured at the headline batch, 348 tiles: 75 us against 70 us for the
    // three-workgroups-per-CU kernel of round 3; at 26 000 rows, 1 224 tiles: 225 against 244 us)
    if (!probs[0].a_major && tiles < 640 && !getenv("GI_B3P_ALL")) return false;
    static const bool grouped_to_b3v = getenv("GI_B3V_GROUPED") && atoi(getenv("GI_B3V_GROUPED"));   // (measurement aid)
    if (probs[0].a_major && probs[0].ngroups && grouped_to_b3v) return false;
    for (int i = 0; i < n; ++i) {
        const gi_gemm_params& p = probs[i];
        if (!(p.flags & GI_GEMM_BF3) || (p.flags & GI_GEMM_BF3A)) return false;
        if (p.a_major != probs[0].a_major || p.b_major != probs[0].b_major) return false;
        if (p.a_major && !p.b_major) return false;
        if (!p.b_major && !(p.flags & GI_GEMM_BF3B_F32)) return false;      // contig B must be plain fp32, not an image
        if (p.a_idx || p.k_dev) return false;
        // a gathered B: the weight-gradient layout only, fp16x2 only, every problem of the launch alike (one kernel)
        if (p.b_idx && !(p.a_major && p.b_major && (p.flags & GI_GEMM_X2) && (p.flags & GI_GEMM_SPLITK))) return false;
        if ((p.b_idx != nullptr) != (probs[0].b_idx != nullptr)) return false;
    }
    return true;
}

int gi_b3p_launch(const gi_gemm_params* probs, int n, void* stream) {
    GpBatch b;
    memset(&b, 0, sizeof(b));
    double flops = 0;
    int total = 0, k = 0, epi = -1;
    const bool am = probs[0].a_major != 0, bmj = probs[0].b_major != 0;
    for (int i = 0; i < n; ++i) {
        const gi_gemm_params& p = probs[i];
        const bool splitk = (p.flags & GI_GEMM_SPLITK) != 0;
        if (!p.A || !p.B || p.M < 0 || p.N <= 0 || p.K < 0 || p.nsplit < 1 || p.ngroups < 0 ||
            p.ngroups > GI_MAX_GROUPS)
            return GI_EINVAL;
        if (!splitk && (p.nsplit != 1 || p.ngroups)) return GI_EINVAL;
        if (p.ngroups && !p.grp_off) return GI_EINVAL;
        if (!p.ngroups && !p.C) return GI_EINVAL;
        if (splitk && (!am || !bmj || p.m_dev)) return GI_EINVAL;            // slabs: weight-gradient layout only
        if (p.ones_col >= 0 && (!bmj || p.ones_col != p.N - 1)) return GI_EINVAL;
        const int f = p.flags & ~(GI_GEMM_BF3 | GI_GEMM_BF3B_F32 | GI_GEMM_SPLITK | GI_GEMM_X2 | GI_GEMM_T128);
        if (f & ~(GI_EPI_BIAS | GI_EPI_SELU | GI_EPI_DSELU | GI_EPI_ACCUM | GI_EPI_MULACT)) return GI_EINVAL;
        if (((p.flags & GI_GEMM_X2) != 0) != ((probs[0].flags & GI_GEMM_X2) != 0)) return GI_EINVAL;
        if ((p.flags & GI_GEMM_X2) && (!p.a_amax || !p.b_amax)) return GI_EINVAL;
        if ((f & GI_EPI_BIAS) && !p.bias) return GI_EINVAL;
        if ((f & (GI_EPI_DSELU | GI_EPI_MULACT)) && !p.act) return GI_EINVAL;
        const long long lim = 0xffffffffLL / 4;
        const int bcols = p.ones_col >= 0 ? p.ones_col : p.N;
        if (!am) {
            if (p.lda < p.K || (p.K < 4 && p.lda < 4)) return GI_EINVAL;
            if ((long long)p.M * p.lda > lim) return GI_ELIMIT;
        } else if (p.lda < p.M || p.M < 1) return GI_EINVAL;
        if (!bmj) {
            if (p.ldb < p.K || (p.K < 4 && p.ldb < 4)) return GI_EINVAL;
            if ((long long)p.N * p.ldb > lim) return GI_ELIMIT;
        } else if (p.ldb < bcols || bcols < 1) return GI_EINVAL;
        const int e = splitk ? (f == 0 ? 3 : 0)
                             : (f == (GI_EPI_BIAS | GI_EPI_SELU) ? 1 : (f == GI_EPI_DSELU ? 2 : (f == 0 ? 3 : 0)));
        epi = (epi < 0 || epi == e) ? e : 0;
        if (p.M == 0) continue;
        int zs = 1;
        if (splitk) {
            zs = p.nsplit;
            if (p.ngroups) { zs = 0; for (int g = 0; g < p.ngroups; ++g) { if (p.gsplit[g] < 1) return GI_EINVAL; zs += p.gsplit[g]; } }
        }
        b.p[k] = p; b.p[k].flags = f | (splitk ? GI_GEMM_SPLITK : 0);
        b.gx[k] = gi_cdiv(p.N, GP_BN); b.gy[k] = gi_cdiv(p.M, GP_BM);
        b.start[k] = total;
        if ((long long)b.gx[k] * b.gy[k] * zs + total > 0x3fffffff) return GI_ELIMIT;
        total += b.gx[k] * b.gy[k] * zs;
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
        ++k;
    }
    if (k == 0) return 0;
    b.start[k] = total; b.n = k; b.total = total;
    bool bounded = false;
    for (int i = 0; i < k; ++i) bounded |= b.p[i].m_dev != nullptr;
    b.remap = (total >= 512 && !bounded && !am) ? 1 : 0;
    {   // Weight-gradient launches (split-K slabs): consecutive tile ids are the column / row tiles of ONE slab, i.e. the
        // workgroups that read the SAME rows of both operands — dealt round-robin to the 8 XCDs each of them pulls its
        // 128 + 256 columns of those rows through a different L2 (2.5-3.1 x the operand bytes per launch through the
        // fabric, profiles/r04).  With the bijective remap one XCD walks consecutive ids: a slab's tiles share an L2.
        // GI_B3P_WGRAD_REMAP=0: dispatch order.
        static const bool wremap = !(getenv("GI_B3P_WGRAD_REMAP") && atoi(getenv("GI_B3P_WGRAD_REMAP")) == 0);
        if (am && wremap && !bounded && total >= 16) b.remap = 1;
    }
    hipStream_t st = (hipStream_t)stream;
    typedef void (*kern_t)(const GpBatch);
    kern_t fn;
    int li;
    const bool x2 = (probs[0].flags & GI_GEMM_X2) != 0;
#define GP_PICK(A, B, E) (x2 ? (kern_t)gi_b3p_kernel<A, B, E, true> : (kern_t)gi_b3p_kernel<A, B, E, false>)
    const bool bidx = probs[0].b_idx != nullptr;
    for (int i = 0; i < n; ++i)
        if ((probs[i].b_idx != nullptr) != bidx || (bidx && !(am && x2 && (probs[i].flags & GI_GEMM_SPLITK)))) return GI_EINVAL;
    if (am && bidx) { if (epi != 3) return GI_EINVAL; li = 3; fn = (kern_t)gi_b3p_kernel<true, true, 3, true, true>; }
    else if (am) { li = 2; if (epi != 3) epi = 0; fn = epi == 3 ? GP_PICK(true, true, 3) : GP_PICK(true, true, 0); }
    else if (bmj) { li = 1; if (epi != 2) epi = 0; fn = epi == 2 ? GP_PICK(false, true, 2) : GP_PICK(false, true, 0); }
    else {
        li = 0;
        if (epi != 1 && epi != 2) epi = 0;
        fn = epi == 1 ? GP_PICK(false, false, 1) : (epi == 2 ? GP_PICK(false, false, 2) : GP_PICK(false, false, 0));
    }
#undef GP_PICK
    const int lds_bytes = 2 * (x2 ? GP_STAGE_X2 : GP_STAGE);
    if (!g_b3p_attr_set[x2 ? 1 : 0][li][epi]) {     // 144 KB (96 KB) of dynamic LDS needs the opt-in
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return (int)hipGetLastError();
        g_b3p_attr_set[x2 ? 1 : 0][li][epi] = true;
    }
    GiProfScope prof(st, GI_PROF_GEMM | (x2 ? GI_PROF_PIPE_X2 : GI_PROF_PIPE_BF3), flops);
    gi_gemm_log_launch(x2 ? (am ? "y2" : (bmj ? "y1" : "y0")) : (am ? "p2" : (bmj ? "p1" : "p0")), b.p, k, total, flops);
    int grid = total;
    if (g_b3p_stream_cus < 0) {
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        const char* e = getenv("GI_B3P_STREAM");                  // 0: one workgroup per tile (measurement aid)
        g_b3p_stream_cus = (e && atoi(e) == 0) ? 0 : cus;
    }
    if (g_b3p_stream_cus > 0 && grid > g_b3p_stream_cus && !bounded) grid = g_b3p_stream_cus;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds_bytes, st, b);
    return gi_launch_status();
}
