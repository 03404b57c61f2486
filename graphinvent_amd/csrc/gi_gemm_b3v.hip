// bf16x3 GEMM family on 32-deep k tiles (gfx950): forward, dgrad and WEIGHT-GRADIENT layouts.
//
// Same arithmetic as gi_gemm_bf3.hip (every fp32 operand = the exact sum of three bf16 planes, six bf16 products
// per fp32 product on v_mfma_f32_32x32x16_bf16, fp32 accumulate; 4e-7 of the fp64 product), new data path:
//
//   * k tiles are 32 deep: a contiguous-k operand row contributes one whole 128-byte line per k tile (the 16-deep
//     tiles of round 3 touched half a line per row and k tile, the other half a tile later — after the CU's 32 KB L1
//     had moved on: the "5.4 TB/s from L2 whatever the operand form" of DESIGN.md section 5 was twice the useful bytes);
//   * operands stored [reduction][rows] ("major": both operands of a weight gradient dZ^T [X | 1], and W of a dgrad
//     as stored) are transposed ON THE WAY INTO LDS, in registers: wave w loads the 8 reduction rows of k chunk w,
//     a lane two adjacent columns of each (global_load_dwordx2: one 512-byte row segment per wave instruction), so
//     a lane ends up holding 8 consecutive k of its two columns = two complete 16-byte MFMA k chunks per plane.
//     No transposing LDS read, no transposed copy of W per backward;
//   * LDS image per operand and plane: [k chunk of 8][row][8 bf16], the rows of chunk c rotated by 2 c rows:
//     fragment reads (ds_read_b128, 32 lanes = 32 consecutive rows of one chunk; banks mod 256 bytes) and both kinds
//     of staging writes (banks mod 128 bytes: ds_write_b64 of a contiguous-k operand — 16 lanes = 2 rows x 8 float4 —
//     and ds_write_b128 of a major one — 8 lanes = 8 column pairs, lanes 4-7 of each eight writing their odd column
//     first) are bank-conflict free (SQ_LDS_BANK_CONFLICT of the first version: a third of the LDS cycles);
//   * one LDS stage (48 KB: three workgroups per CU) + one register stage: global -> registers of tile t + 1 is in
//     flight under the MFMAs of tile t, split + LDS write after the barrier that ends tile t's reads.
//
//   C[M, N] = epilogue( sum_k A(m, k) B(n, k) ),  A contig [M][lda] or major [K][lda], B contig [N][ldb] or major
//   [K][ldb] (ones_col: stored column N - 1 of a major B reads as 1.0 -> bias gradient), split-K slabs (plain or
//   grouped by grp_off / gsplit like gi_gemm.hip), block tile 128 x 128 x 32, 4 waves x (64 x 64).
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <type_traits>

#include "gi_common.h"
#include "gi_mfma.h"
#include "gi_x2.h"

typedef __bf16 gv_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gv_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gv_f32x2 __attribute__((ext_vector_type(2)));
typedef gv_f32x2 gv_f32x2_u __attribute__((aligned(4)));
typedef unsigned gv_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gv_u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ float gv_sink[256];                     // where out-of-range lanes of edge tiles store

constexpr int GV_BM = 128, GV_BN = 128, GV_BK = 32;
constexpr int GV_CH = 128 * 16;                    // bytes of one k chunk (8 bf16) of all 128 rows
constexpr int GV_PLANE = 4 * GV_CH;                // 8 KB

__device__ __forceinline__ unsigned gv_lds(int plane, int kc, int row) {
    return plane * GV_PLANE + kc * GV_CH + ((row * 16 + kc * 32) & (GV_CH - 1));
}
__device__ __forceinline__ unsigned gv_pk(float lo, float hi) {
    gv_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gv_bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float gv_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float gv_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ void gv_split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = gv_pk(x0, x1);
    const float r0 = x0 - gv_lo(p0), r1 = x1 - gv_hi(p0);
    p1 = gv_pk(r0, r1);
    p2 = gv_pk(r0 - gv_lo(p1), r1 - gv_hi(p1));
}

struct GvBatch {
    gi_gemm_params p[8];
    int start[9];
    int gx[8], gy[8];
    int n, total, remap;
};

// EPI: 0 = epilogue from the run-time flags, 1 = bias + SELU (forward), 2 = * selu'(act) (dgrad), 3 = plain store
// (weight-gradient slabs).
// X2 (GI_GEMM_X2, round 6): the operands as two scaled fp16 planes (gi_x2.h) — 32 KB of LDS instead of 48, 24 instead of
// 48 MFMAs per wave and k tile; scales from a_amax / b_amax.  BIDX (weight-gradient layout): B's reduction rows gathered
// through p.b_idx (wave-uniform indices).
template <bool AM, bool BMJ, int EPI, bool X2 = false, bool BIDX = false>
__global__ __launch_bounds__(256, X2 ? 4 : 3) void gi_b3v_kernel(const GvBatch b) {
    constexpr int NP = X2 ? 2 : 3;
    constexpr int OPER = NP * GV_PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OPER];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + OPER;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lhi = lane >> 5;

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
    const int gx = b.gx[pi], gxy = gx * b.gy[pi];
    const int bz = local / gxy;
    const int rem = local - bz * gxy;
    const int by = rem / gx, bx = rem - by * gx;
    const int m_end = p.m_dev ? min(p.M, *p.m_dev) : p.M;
    const int m0 = by * GV_BM, n0 = bx * GV_BN;
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
    const int nk = ke > kb ? (ke - kb + GV_BK - 1) / GV_BK : 0;
    const int n_full = ke > kb ? (ke - kb) / GV_BK : 0;
    float sa = 1.f, ia = 1.f, sb = 1.f, ib = 1.f;               // fp16x2: per-tensor power-of-two scales
    if (X2) { gx_scale(gx_amax_read(p.a_amax), sa, ia); gx_scale(gx_amax_read(p.b_amax), sb, ib); }

    // ---- staging coordinates -----------------------------------------------------------------------
    // contig operand: 8 float4 per 32-deep row -> c8 = tid & 7, rows (tid >> 3) + 32 i: 4 float4 per thread
    // major operand : wave = k chunk (8 reduction rows), lane = columns 2 lane, 2 lane + 1: 8 float2 per thread
    const int c8 = tid & 7, crow = tid >> 3;
    const int a_cols = p.M;                                       // stored columns of a major A
    const int b_cols = p.ones_col >= 0 ? p.ones_col : p.N;        // ... of a major B (the ones column is not stored)
    unsigned a_off[4], b_off[4];                                  // contig: row byte offset + 16 c8 / major: column byte offset
    unsigned a_w[4], b_w[4];                                      // contig: LDS byte offset of plane 0
    const int a_cmax = (p.lda >= ((p.K + 3) & ~3)) ? ((p.K + 3) & ~3) - 4 : p.K - 4;
    const int b_cmax = (p.ldb >= ((p.K + 3) & ~3)) ? ((p.K + 3) & ~3) - 4 : p.K - 4;
    if (!AM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = crow + 32 * i;
            int row = min(m0 + rl, m_end - 1);
            a_off[i] = (unsigned)row * (unsigned)p.lda * 4u;
            a_w[i] = gv_lds(0, c8 >> 1, rl) + 8 * (c8 & 1);
        }
    } else {
        a_off[0] = 4u * (unsigned)min(m0 + 2 * lane, (a_cols - 1) & ~1);
    }
    if (!BMJ) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = crow + 32 * i;
            const int row = min(n0 + rl, p.N - 1);
            b_off[i] = (unsigned)row * (unsigned)p.ldb * 4u;
            b_w[i] = gv_lds(0, c8 >> 1, rl) + 8 * (c8 & 1);
        }
    } else {
        b_off[0] = 4u * (unsigned)min(n0 + 2 * lane, (b_cols - 1) & ~1);
    }
    // the ones column of a major B (bias gradient), if this tile holds it: which of the lane's two columns
    const int ones_j = (BMJ && p.ones_col >= 0) ? p.ones_col - (n0 + 2 * lane) : -1;     // 0 / 1 = this lane's
    const bool tile_has_ones = BMJ && p.ones_col >= n0 && p.ones_col < n0 + GV_BN;

    v4f ra[4], rb[4];                    // contig stages
    gv_f32x2 ma[8], mb[8];               // major stages

    auto gload = [&](auto steady_c, int kt) __attribute__((always_inline)) {
        constexpr bool ST = decltype(steady_c)::value;
        const int k0 = kb + kt * GV_BK;
        if (!AM) {
            if (ST) {
                const char* base = (const char*)p.A + (size_t)k0 * 4 + 16 * c8;
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i] = *(const v4f_u*)(base + a_off[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    ra[i] = gi_load4_raw((const float*)((const char*)p.A + a_off[i]), k0 + 4 * c8, a_cmax);
            }
        } else {
            const int kr = k0 + 8 * wid;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = ST ? kr + j : min(kr + j, ke - 1);
                ma[j] = *(const gv_f32x2_u*)((const char*)p.A + (size_t)row * (size_t)p.lda * 4 + a_off[0]);
            }
        }
        if (!BMJ) {
            if (ST) {
                const char* base = (const char*)p.B + (size_t)k0 * 4 + 16 * c8;
#pragma unroll
                for (int i = 0; i < 4; ++i) rb[i] = *(const v4f_u*)(base + b_off[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    rb[i] = gi_load4_raw((const float*)((const char*)p.B + b_off[i]), k0 + 4 * c8, b_cmax);
            }
        } else {
            const int kr = k0 + 8 * wid;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int row = ST ? kr + j : min(kr + j, ke - 1);
                if (BIDX) row = __builtin_amdgcn_readfirstlane(p.b_idx[row]);
                mb[j] = *(const gv_f32x2_u*)((const char*)p.B + (size_t)row * (size_t)p.ldb * 4 + b_off[0]);
            }
        }
    };
    // 8 consecutive k of one row -> the three planes' 16-byte chunks
    auto split8 = [&](const float (&x)[8], float s, gv_u32x4& q0, gv_u32x4& q1, gv_u32x4& q2) __attribute__((always_inline)) {
        unsigned a0, a1, a2 = 0, b0, b1, b2 = 0, c0, c1, c2 = 0, d0, d1, d2 = 0;
        if (X2) {
            gx_split2(x[0], x[1], s, a0, a1); gx_split2(x[2], x[3], s, b0, b1);
            gx_split2(x[4], x[5], s, c0, c1); gx_split2(x[6], x[7], s, d0, d1);
        } else {
        gv_split2(x[0], x[1], a0, a1, a2);
        gv_split2(x[2], x[3], b0, b1, b2);
        gv_split2(x[4], x[5], c0, c1, c2);
        gv_split2(x[6], x[7], d0, d1, d2);
        }
        q0.x = a0; q0.y = b0; q0.z = c0; q0.w = d0;
        q1.x = a1; q1.y = b1; q1.z = c1; q1.w = d1;
        q2.x = a2; q2.y = b2; q2.z = c2; q2.w = d2;
    };
    float rowmax[4] = {0.f, 0.f, 0.f, 0.f};        // fp16x2 guard (gi_gemm_params.x2_guard): max |a| staged of the thread's four contig A rows
    auto store_contig = [&](auto steady_c, int kt, unsigned char* S, v4f (&r)[4], const unsigned (&w)[4], int cmax,
                            int kend, float scale, bool track) __attribute__((always_inline)) {
        constexpr bool ST = decltype(steady_c)::value;
        const int k0 = kb + kt * GV_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v4f v = r[i];
            if (!ST) v = gi_fix4(v, k0 + 4 * c8, cmax, kend, true);
            if (X2 && track) rowmax[i] = fmaxf(fmaxf(rowmax[i], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            gv_u32x2 w0, w1, w2;
            unsigned x0, x1, x2 = 0, y0, y1, y2 = 0;
            if (X2) { gx_split2(v.x, v.y, scale, x0, x1); gx_split2(v.z, v.w, scale, y0, y1); }
            else { gv_split2(v.x, v.y, x0, x1, x2); gv_split2(v.z, v.w, y0, y1, y2); }
            w0.x = x0; w0.y = y0; w1.x = x1; w1.y = y1; w2.x = x2; w2.y = y2;
            *reinterpret_cast<gv_u32x2*>(S + w[i]) = w0;
            *reinterpret_cast<gv_u32x2*>(S + GV_PLANE + w[i]) = w1;
            if (!X2) *reinterpret_cast<gv_u32x2*>(S + 2 * GV_PLANE + w[i]) = w2;
        }
    };
    auto store_major = [&](auto steady_c, int kt, unsigned char* S, gv_f32x2 (&m)[8], bool ones, float scale) __attribute__((always_inline)) {
        // fp16x2: the ones column is staged as 1 / scale, i.e. exactly 1.0 AFTER the scale whatever the tensor's amax (the
        // epilogue leaves that column's 1 / sb out)
        const float one = X2 ? ib : 1.f;
        constexpr bool ST = decltype(steady_c)::value;
        const int kr = kb + kt * GV_BK + 8 * wid;
        float x[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = m[j].x; y[j] = m[j].y;
            if (ones) {                                          // (block-uniform branch; lane-wise select)
                x[j] = ones_j == 0 ? one : x[j];
                y[j] = ones_j == 1 ? one : y[j];
            }
            if (!ST) {                                           // zero fill along the reduction (wave-uniform)
                const bool ok = kr + j < ke;
                x[j] = ok ? x[j] : 0.f; y[j] = ok ? y[j] : 0.f;
            }
        }
        // eight consecutive lanes write eight 16-byte rows per instruction: rows 2 L (32 bytes apart) would collide
        // pairwise mod 128 bytes, so lanes 4-7 of every eight write their odd column first
        const bool swp = (lane & 4) != 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float a = x[j], c = y[j]; x[j] = swp ? c : a; y[j] = swp ? a : c; }
        const int sw = swp ? 1 : 0;
        gv_u32x4 q0, q1, q2;
        split8(x, scale, q0, q1, q2);
        const unsigned o0 = gv_lds(0, wid, 2 * lane + sw);
        *reinterpret_cast<gv_u32x4*>(S + o0) = q0;
        *reinterpret_cast<gv_u32x4*>(S + GV_PLANE + o0) = q1;
        if (!X2) *reinterpret_cast<gv_u32x4*>(S + 2 * GV_PLANE + o0) = q2;
        split8(y, scale, q0, q1, q2);
        const unsigned o1 = gv_lds(0, wid, 2 * lane + 1 - sw);
        *reinterpret_cast<gv_u32x4*>(S + o1) = q0;
        *reinterpret_cast<gv_u32x4*>(S + GV_PLANE + o1) = q1;
        if (!X2) *reinterpret_cast<gv_u32x4*>(S + 2 * GV_PLANE + o1) = q2;
    };
    auto sstore = [&](auto steady_c, int kt) __attribute__((always_inline)) {
        if (!AM) store_contig(steady_c, kt, As, ra, a_w, a_cmax, p.K, sa, true);
        else store_major(steady_c, kt, As, ma, false, sa);
        if (!BMJ) store_contig(steady_c, kt, Bs, rb, b_w, b_cmax, p.K, sb, false);
        else store_major(steady_c, kt, Bs, mb, tile_has_ones, sb);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            gv_bf16x8 af[2][3], bf[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned oa = gv_lds(0, 2 * s + lhi, wm * 64 + t * 32 + l31);
                const unsigned ob = gv_lds(0, 2 * s + lhi, wn * 64 + t * 32 + l31);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    af[t][pl] = *reinterpret_cast<const gv_bf16x8*>(As + pl * GV_PLANE + oa);
                    bf[t][pl] = *reinterpret_cast<const gv_bf16x8*>(Bs + pl * GV_PLANE + ob);
                }
            }
            // smallest terms first; four independent accumulators between two MFMAs on the same one
            constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
            constexpr int XA[3] = {1, 0, 0}, XB[3] = {0, 1, 0};      // fp16x2: a2 b1 + a1 b2 + a1 b1
#pragma unroll
            for (int term = 0; term < (X2 ? 3 : 6); ++term)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (X2)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gx_f16x8, af[t][XA[term]]),
                                                                               __builtin_bit_cast(gx_f16x8, bf[u][XB[term]]),
                                                                               acc[t][u], 0, 0, 0);
                        else
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][TA[term]], bf[u][TB[term]],
                                                                                acc[t][u], 0, 0, 0);
                    }
        }
    };

    // ---- k loop: one LDS stage, one register stage ---------------------------------------------------------
    const std::true_type ST{};
    const std::false_type GEN{};
    if (nk > 0) {
        if (n_full >= 1) gload(ST, 0); else gload(GEN, 0);
        int kt = 0;
        for (; kt + 1 < n_full; ++kt) {              // this tile and the next are full in k
            sstore(ST, kt);
            __syncthreads();
            gload(ST, kt + 1);
            compute();
            __syncthreads();
        }
        for (; kt < nk; ++kt) {                      // the last full tile and / or the partial one
            sstore(GEN, kt);
            __syncthreads();
            if (kt + 1 < nk) gload(GEN, kt + 1);
            compute();
            __syncthreads();
        }
    }

    // ---- fp16x2 dynamic-range guard (forward / dgrad layouts; see gi_gemm_bf3.hip): rows of A whose largest scaled
    // magnitude is below 2^-11 keep fewer than ~14 bits.  Once per launch: the workgroups of the first column tile.
    if (X2 && !AM && p.x2_guard && bx == 0 && !(p.flags & GI_GEMM_SPLITK)) {
        int n_low = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float m = rowmax[i];
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            m = fmaxf(m, __shfl_xor(m, 4));                     // the eight lanes (k chunks) of a row
            const bool real_row = m0 + crow + 32 * i < m_end;
            n_low += (c8 == 0 && real_row && m > 0.f && m * sa < 0x1p-11f) ? 1 : 0;
        }
        if (n_low) {
            atomicAdd(p.x2_guard, n_low);
            if (p.x2_guard_host) *reinterpret_cast<volatile int*>(p.x2_guard_host) = 1;
        }
    }

    // ---- epilogue (C/D layout of a 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) --
    const int flags = EPI == 1 ? (GI_EPI_BIAS | GI_EPI_SELU) : (EPI == 2 ? GI_EPI_DSELU : (EPI == 3 ? 0 : p.flags));
    const bool need_act = (flags & (GI_EPI_DSELU | GI_EPI_MULACT)) != 0;
    const bool need_c = (flags & GI_EPI_ACCUM) != 0;
    float amax = 0.f;                                             // largest |value| stored (gi_gemm_params.c_amax)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = n0 + wn * 64 + u * 32 + l31;
            const bool col_ok = col < p.N;
            const int colc = col_ok ? col : p.N - 1;
            const int row0 = m0 + wm * 64 + t * 32 + 4 * lhi;
            const float bv = (flags & GI_EPI_BIAS) ? p.bias[colc] : 0.f;
            const float ibc = (X2 && BMJ && col == p.ones_col) ? 1.f : ib;             // (the ones column was staged as 1 / sb)
            float av[16], cv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {                        // every load of the block before the first store
                const int row = min(row0 + 8 * (r >> 2) + (r & 3), m_end - 1);
                if (need_act) av[r] = p.act[(long long)row * p.ldact + colc];
                if (need_c) cv[r] = Cp[(long long)row * p.ldc + colc];
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = X2 ? (acc[t][u][r] * ia) * ibc + bv : acc[t][u][r] + bv;    // (two steps: ia * ib alone may leave fp32's range)
                if (flags & GI_EPI_SELU) x = gi_selu(x);
                if (flags & GI_EPI_DSELU) x *= gi_selu_grad(av[r]);
                if (flags & GI_EPI_MULACT) x *= av[r];
                if (flags & GI_EPI_ACCUM) x += cv[r];
                v[r] = x;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 8 * (r >> 2) + (r & 3);
                const bool ok = col_ok & (row < m_end);
                float* dst = ok ? Cp + (long long)row * p.ldc + col : gv_sink + tid;
                *dst = v[r];
                amax = fmaxf(amax, ok ? fabsf(v[r]) : 0.f);
            }
        }
    }
    // A bf16x3 launch can be the PRODUCER of a tensor an fp16x2 launch reads next (a stack's last layer 192+ wide has no
    // amax cell for its own dZ and stays bf16x3, while the dZ it hands to the layer below is scaled from this cell).
    // Round 6: this kernel left the cell at zero — scale 1, fp16 planes of raw 1e-4-sized dZ values, every gradient
    // upstream of such a layer ~5e-4 off (tests/test_dims_gpu.py, A = 432).
    if (p.c_amax && !(p.flags & GI_GEMM_SPLITK)) gx_amax_publish(amax, p.c_amax);
}

int g_b3v_enabled = -1;

}  // namespace

// Process-wide switch (measurement aid): route eligible GI_GEMM_BF3 launches to the 32-deep-tile kernels of this
// file (1, default; environment GI_B3V) or to the round-3 kernel (0).  on < 0 only queries; returns the previous value.
extern "C" int gi_b3v_enable(int on) {
    if (g_b3v_enabled < 0) {
        const char* e = getenv("GI_B3V");
        g_b3v_enabled = e ? (atoi(e) != 0) : 1;
    }
    const int prev = g_b3v_enabled;
    if (on >= 0) g_b3v_enabled = on ? 1 : 0;
    return prev;
}

// Can this launch run here?  fp32 operands only (no pre-split images), no row gathers, even leading dimensions
// for major operands (8-byte column pairs), 32-bit offsets for contig ones.
bool gi_b3v_eligible(const gi_gemm_params* probs, int n) {
    if (!gi_b3v_enable(-1)) return false;
    for (int i = 0; i < n; ++i) {
        const gi_gemm_params& p = probs[i];
        if (!(p.flags & GI_GEMM_BF3) || (p.flags & GI_GEMM_BF3A)) return false;
        // fp16x2 (round 6): the weight-gradient layout, every problem of the launch alike
        if (((p.flags & GI_GEMM_X2) != 0) != ((probs[0].flags & GI_GEMM_X2) != 0)) return false;
        if ((p.flags & GI_GEMM_X2) && !(p.a_amax && p.b_amax)) return false;
        if ((p.flags & GI_GEMM_X2) && !(p.a_major && p.b_major)) {       // forward / dgrad layouts as fp16x2: measurement switch
            static const bool fwd_x2 = getenv("GI_B3V_X2_FWD") && atoi(getenv("GI_B3V_X2_FWD")) != 0;
            if (!fwd_x2) return false;
        }
        if (p.a_major != probs[0].a_major || p.b_major != probs[0].b_major) return false;
        if (p.a_major && !p.b_major) return false;
        if (!p.b_major && !(p.flags & GI_GEMM_BF3B_F32)) return false;      // contig B must be plain fp32, not an image
        if (p.a_idx || p.k_dev) return false;
        if (p.b_idx && !((p.flags & GI_GEMM_X2) && (p.flags & GI_GEMM_SPLITK))) return false;    // gathered B: fp16x2 slabs only
        if ((p.b_idx != nullptr) != (probs[0].b_idx != nullptr)) return false;
    }
    return true;
}
// fp16x2 weight-gradient launches go to the pipelined 128 x 256 kernel (gi_gemm_b3p.hip) unless they ask for this one
bool gi_b3v_wants(const gi_gemm_params* probs, int n) {
    for (int i = 0; i < n; ++i)
        if (!(probs[i].flags & GI_GEMM_T128)) return false;
    return n > 0 && gi_b3v_eligible(probs, n);
}

int gi_b3v_launch(const gi_gemm_params* probs, int n, void* stream) {
    GvBatch b;
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
        if ((f & GI_EPI_BIAS) && !p.bias) return GI_EINVAL;
        if ((f & (GI_EPI_DSELU | GI_EPI_MULACT)) && !p.act) return GI_EINVAL;
        if (((p.flags & GI_GEMM_X2) != 0) != ((probs[0].flags & GI_GEMM_X2) != 0)) return GI_EINVAL;
        if ((p.flags & GI_GEMM_X2) && (!p.a_amax || !p.b_amax || (splitk && f != 0))) return GI_EINVAL;
        if ((p.b_idx != nullptr) != (probs[0].b_idx != nullptr) || (p.b_idx && !(p.flags & GI_GEMM_X2))) return GI_EINVAL;
        const long long lim = 0xffffffffLL / 4;
        const int bcols = p.ones_col >= 0 ? p.ones_col : p.N;
        if (!am) {
            if (p.lda < p.K || (p.K < 4 && p.lda < 4) || (long long)p.M * p.lda > lim) return p.lda < p.K ? GI_EINVAL : GI_ELIMIT;
        } else if (p.lda < p.M + (p.M & 1) || p.M < 1) return GI_EINVAL;
        if (!bmj) {
            if (p.ldb < p.K || (p.K < 4 && p.ldb < 4) || (long long)p.N * p.ldb > lim) return p.ldb < p.K ? GI_EINVAL : GI_ELIMIT;
        } else if (p.ldb < bcols + (bcols & 1) || bcols < 1) return GI_EINVAL;
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
        b.gx[k] = gi_cdiv(p.N, GV_BN); b.gy[k] = gi_cdiv(p.M, GV_BM);
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
    hipStream_t st = (hipStream_t)stream;
    const bool x2 = (probs[0].flags & GI_GEMM_X2) != 0, bidx = probs[0].b_idx != nullptr;
    GiProfScope prof(st, GI_PROF_GEMM | (x2 ? GI_PROF_PIPE_X2 : GI_PROF_PIPE_BF3), flops);
    gi_gemm_log_launch(x2 ? "v2" : (am ? "b2" : (bmj ? "b1" : "b0")), b.p, k, total, flops);
    if (x2 && !am) {                                     // forward / dgrad layouts (GI_B3V_X2_FWD)
        if (bidx) return GI_EINVAL;
#define GV_LAUNCH_X2(A, B, E) hipLaunchKernelGGL((gi_b3v_kernel<A, B, E, true, false>), dim3(total), dim3(256), 0, st, b)
        if (bmj) { if (epi == 2) GV_LAUNCH_X2(false, true, 2); else GV_LAUNCH_X2(false, true, 0); }
        else { if (epi == 1) GV_LAUNCH_X2(false, false, 1); else if (epi == 2) GV_LAUNCH_X2(false, false, 2); else GV_LAUNCH_X2(false, false, 0); }
#undef GV_LAUNCH_X2
        return gi_launch_status();
    }
    if (x2) {                                            // (weight-gradient slabs: plain stores)
        if (epi != 3) return GI_EINVAL;
        // weight-gradient launches: consecutive tile ids = the tiles of ONE slab, which read the same rows of both
        // operands — one XCD (one L2) per slab instead of eight (gi_gemm_b3p.hip)
        if (total >= 16 && !bounded) b.remap = 1;
        if (bidx) hipLaunchKernelGGL((gi_b3v_kernel<true, true, 3, true, true>), dim3(total), dim3(256), 0, st, b);
        else hipLaunchKernelGGL((gi_b3v_kernel<true, true, 3, true, false>), dim3(total), dim3(256), 0, st, b);
        return gi_launch_status();
    }
#define GV_LAUNCH(A, B, E) hipLaunchKernelGGL((gi_b3v_kernel<A, B, E>), dim3(total), dim3(256), 0, st, b)
    if (am) { if (epi == 3) GV_LAUNCH(true, true, 3); else GV_LAUNCH(true, true, 0); }
    else if (bmj) { if (epi == 2) GV_LAUNCH(false, true, 2); else GV_LAUNCH(false, true, 0); }
    else { if (epi == 1) GV_LAUNCH(false, false, 1); else if (epi == 2) GV_LAUNCH(false, false, 2); else GV_LAUNCH(false, false, 0); }
#undef GV_LAUNCH
    return gi_launch_status();
}
