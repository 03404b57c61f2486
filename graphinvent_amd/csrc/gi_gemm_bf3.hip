// fp32 GEMM on the bf16 MFMA pipe by three-way operand splits (gfx950).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate and, on this path's big launches, at 0.60-0.64 of
// its own peak with the chip power-limited.  Every fp32 value is the exact sum of
// three bf16 values up to 2^-25 of its magnitude,
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (round to nearest even;
//                                                                    both residuals are exact in fp32)
// so   a b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + O(2^-24 |a b|):
// six bf16 products per fp32 product, exact in the MFMA's fp32 accumulate, 12 v_mfma_f32_32x32x16_bf16 (32 cycles
// each) per 32x32x32 block product against 16 v_mfma_f32_32x32x2_f32 (64 cycles each): 2.7x on the MFMA pipe
// at an error of the size of fp32's own rounding (measured against the fp64 product: 4.3e-7 relative, the fp32 MFMA chain 5.1e-7; the path's
// parity bar is 1e-4).
//
//   C[M, N] = epilogue( A[M, K] . B[N, K]^T )
// Operand forms: fp32 row-major, split while it is staged (A always in the model; B with GI_GEMM_BF3B_F32: the
// forward's weights as stored, dgrad's W^T as a plain fp32 transposed copy from gi_bf3_pack(as_f32)), or a
// PRE-SPLIT image of three bf16 planes [rows][Kp], Kp = K rounded up to 32, zero padded (gi_bf3_pack; B by
// default, A with GI_GEMM_BF3A).  The launch tracks the bytes its workgroups pull from L2 (measured: 16 / 20 / 24 KB
// per workgroup and k tile -> 75.6 / 82.7 / 95.7 us), so the model uses the fp32 form for both operands.
// Block = 256 threads = 4 waves (2 x 2), block tile 128 x 128 x 16, wave tile 64 x 64 = 2 x 2 accumulators of
// 32 x 32.  LDS per k tile and operand: three planes of 128 rows x 16 bf16 (32 B per row, its two 16-byte
// chunks swapped when (row >> 3) & 1: the 16 lanes of every ds_read_b128 lane group — rows {0-3, 12-15, 20-27} /
// {4-11, 16-19, 28-31} of a fragment — then hit the 16 distinct 16-byte slots of the 256-byte bank row), double
// buffered: 48 KB and <= 168 VGPRs -> THREE workgroups per CU, whose staging / conversion phases run under each
// other's MFMAs (one barrier per k tile of 24 MFMAs per wave); k tiles travel global -> registers two tiles ahead;
// launches of >= 512 tiles walk them in the XCD-aware order of gi_gemm.hip.
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <type_traits>

#include "gi_common.h"
#include "gi_mfma.h"
#include "gi_x2.h"

typedef __bf16 gi_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gi_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gi_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gi_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gi_u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ float b3_sink[256];
#ifdef B3_PIN            // lab: keep the steady loop's global loads where the source has them (two k tiles ahead)
#define B3_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define B3_SB()
#endif                    // where out-of-range lanes of edge tiles store

constexpr int B3_BM = 128, B3_BN = 128, B3_BK = 16;
constexpr int B3_ROWB = B3_BK * 2;                 // bytes of one row of one plane of a k tile (16 bf16)
constexpr int B3_PLANE = 128 * B3_ROWB;            // 4 KB

__host__ __device__ inline int b3_r32(int k) { return (k + 31) & ~31; }

__device__ __forceinline__ unsigned b3_pk(float lo, float hi) {
    gi_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gi_bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float b3_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float b3_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// two fp32 values -> their three bf16 planes, packed pairwise (low half = x0)
__device__ __forceinline__ void b3_split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = b3_pk(x0, x1);
    const float r0 = x0 - b3_lo(p0), r1 = x1 - b3_hi(p0);
    p1 = b3_pk(r0, r1);
    const float s0 = r0 - b3_lo(p1), s1 = r1 - b3_hi(p1);
    p2 = b3_pk(s0, s1);
}

// ---- operand image ------------------------------------------------------------------------------------
struct B3PackArgs {
    gi_bf3_pack_desc d[GI_BF3_PACK_MAX];
    long long start[GI_BF3_PACK_MAX + 1];      // prefix of k PAIRS over the descriptors
    int n;
};

__global__ __launch_bounds__(256) void gi_bf3_pack_kernel(const B3PackArgs a) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= a.start[a.n]) return;
    int i = 0;
    while (i < a.n - 1 && id >= a.start[i + 1]) ++i;
    const gi_bf3_pack_desc& d = a.d[i];
    const int Kp = b3_r32(d.cols), half = Kp >> 1;
    const long long local = id - a.start[i];
    const int n = (int)(local / half), k = 2 * (int)(local - (long long)n * half);
    auto src = [&](int kk) -> float {
        if (kk >= d.cols) return 0.f;
        return d.transpose ? d.W[(long long)kk * d.ld + n] : d.W[(long long)n * d.ld + kk];
    };
    if (d.as_f32) {                                            // plain fp32 copy [rows][r4(cols)] (transposed or not)
        const int ldo = (d.cols + 3) & ~3;
        float* out = reinterpret_cast<float*>(d.image) + (long long)n * ldo;
        if (k < ldo) { gi_f32x2 v = {src(k), src(k + 1)}; *reinterpret_cast<gi_f32x2*>(out + k) = v; }
        return;
    }
    unsigned p0, p1, p2;
    b3_split2(src(k), src(k + 1), p0, p1, p2);
    unsigned* img = reinterpret_cast<unsigned*>(d.image);
    const long long plane = (long long)d.rows * half, at = (long long)n * half + (k >> 1);
    img[at] = p0; img[plane + at] = p1; img[2 * plane + at] = p2;
}

// ---- GEMM ---------------------------------------------------------------------------------------------
struct B3Batch {
    gi_gemm_params p[8];
    int start[9];
    int gx[8];
    int n, total;
    int remap;                                 // XCD-aware tile order (launches of many workgroups)
};

// EPI: 0 = epilogue from the run-time flags, 1 = bias + SELU (forward), 2 = * selu'(act) (dgrad).
// APL: the A operand is a pre-split image too (GI_GEMM_BF3A: three bf16 planes [M][Kp], what an epilogue with
// gi_gemm_params.planes wrote): its staging is then a plain copy like B's.
// BFP: B is plain fp32 [N][ldb] (GI_GEMM_BF3B_F32) and split while staging like A — 4 bytes per element through
// L2 instead of the image's 6.
// X2 (GI_GEMM_X2, with BFP and not APL): the operands as two scaled fp16 values each instead of three bf16 (gi_x2.h):
// two LDS planes per operand, three f16 MFMA products per fp32 product, scales from a_amax / b_amax, 1 / (sa sb) in
// the epilogue.
template <int EPI, bool APL, bool BFP, bool X2 = false>
__global__ __launch_bounds__(256, 3) void gi_gemm_bf3_kernel(const B3Batch b) {
    constexpr int NP = X2 ? 2 : 3;
    constexpr int OPER = NP * B3_PLANE, BUF = 2 * OPER;           // one operand of a k tile (NP planes), A + B
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lhi = lane >> 5;

    // ---- tile (block-uniform) ----------------------------------------------------------------------
    int pi = 0;
    while (pi < b.n - 1 && (int)blockIdx.x >= b.start[pi + 1]) ++pi;
    const gi_gemm_params& p = b.p[pi];
    int local = blockIdx.x - b.start[pi];
    {   // XCD-aware tile order (as in gi_gemm.hip): consecutive workgroup ids go round-robin to the 8 XCDs; the
        // remap lets one XCD walk consecutive tiles, so the column tiles that share an A row panel hit one L2
        const int tiles = b.start[pi + 1] - b.start[pi];
        if (b.remap) {
            const int q = tiles >> 3, r = tiles & 7, xcd = local & 7, j = local >> 3;
            local = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        }
    }
    const int by = local / b.gx[pi], bx = local - by * b.gx[pi];
    const int m_end = p.m_dev ? min(p.M, *p.m_dev) : p.M;
    const int m0 = by * B3_BM, n0 = bx * B3_BN;
    if (m0 >= m_end) return;                                   // (bounded launch: beyond the rows on the device)
#ifdef B3_LAB_EMPTY                        // lab, TIMING ONLY: the launch itself
    if (p.K != 123456) return;
#endif
    const int K = p.K, Kp = b3_r32(K), nk = Kp / B3_BK;
    float sa = 1.f, ia = 1.f, sb = 1.f, ib = 1.f;               // fp16x2: per-tensor power-of-two scales
#ifdef B3_LAB_NO_AMAX                       // lab, TIMING ONLY
    if (X2) { sa = sb = 0x1p10f; ia = ib = 0x1p-10f; }
#else
    if (X2) { gx_scale(gx_amax_read(p.a_amax), sa, ia); gx_scale(gx_amax_read(p.b_amax), sb, ib); }
#endif
    const unsigned char* const Bimg = reinterpret_cast<const unsigned char*>(p.B);
    const long long bplane = (long long)p.N * Kp * 2;          // bytes per plane of the image

    // ---- staging coordinates -----------------------------------------------------------------------
    // A: 128 rows x 4 float4 per k tile -> 2 float4 per thread: k chunk c4 = tid & 3, rows (tid >> 2) + 64 i
    const int c4 = tid & 3;
    unsigned a_off[2], a_lds[2];
    const int a_cmax = (p.lda >= ((K + 3) & ~3)) ? ((K + 3) & ~3) - 4 : K - 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = (tid >> 2) + 64 * i;
        int row = min(m0 + rl, m_end - 1);
        if (p.a_idx) row = p.a_idx[row];
        a_off[i] = (unsigned)row * (unsigned)p.lda * 4u;
        a_lds[i] = rl * B3_ROWB + 16 * ((c4 >> 1) ^ ((rl >> 3) & 1)) + 8 * (c4 & 1);
    }
    // B as fp32 (BFP): the same staging as A on rows n0 ..
    unsigned bf_off[2], bf_lds[2];
    const int bf_cmax = (p.ldb >= ((K + 3) & ~3)) ? ((K + 3) & ~3) - 4 : K - 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = (tid >> 2) + 64 * i;
        const int row = min(n0 + rl, p.N - 1);
        bf_off[i] = (unsigned)row * (unsigned)p.ldb * 4u;
        bf_lds[i] = rl * B3_ROWB + 16 * ((c4 >> 1) ^ ((rl >> 3) & 1)) + 8 * (c4 & 1);
    }
    // B: per plane 128 rows x 2 chunks of 16 B -> one chunk per thread and plane
    unsigned b_off, b_lds;
    {
        const int rl = tid >> 1, c = tid & 1;
        const int row = min(n0 + rl, p.N - 1);
        b_off = (unsigned)row * (unsigned)Kp * 2u + 16u * c;
        b_lds = rl * B3_ROWB + 16 * (c ^ ((rl >> 3) & 1));
    }

    // two register stages: k tile kt + 2 travels global -> registers during the MFMAs of tiles kt and kt + 1
    // (16-deep tiles are 24 MFMAs = 0.3 us of work per wave, far less than a trip to L2 / HBM).
    // STEADY tiles are full in k: uniform base (advanced by the SALU) + a constant 32-bit lane offset per load, no
    // clamping, no fix-up — the generic form (clamped column, zero fill beyond K) only runs a tile's last steps.
    // fp16x2 dynamic-range guard (gi_gemm_params.x2_guard): the largest magnitude this thread staged of its two A rows
    float rowmax[2] = {0.f, 0.f};
    v4f ra0[2], ra1[2], rf0[2], rf1[2];            // (rf*: B as fp32 when BFP, else unused)
    gi_u32x4 rb0[3], rb1[3], rp0[3], rp1[3];       // (rp*: A planes when APL, else unused)
    unsigned bf_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bf_voff[i] = bf_off[i] + 16u * c4;
    const unsigned char* const Aimg = reinterpret_cast<const unsigned char*>(p.A);
    const long long aplane = (long long)p.M * Kp * 2;
    unsigned ap_off, ap_lds;
    {
        const int rl = tid >> 1, c = tid & 1;
        const int row = min(m0 + rl, m_end - 1);
        ap_off = (unsigned)row * (unsigned)Kp * 2u + 16u * c;
        ap_lds = rl * B3_ROWB + 16 * (c ^ ((rl >> 3) & 1));
    }
    unsigned a_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_voff[i] = a_off[i] + 16u * c4;
    auto gload = [&](auto steady_c, int kt, v4f (&ra)[2], gi_u32x4 (&rb)[3], gi_u32x4 (&rp)[3], v4f (&rf)[2]) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;
        const int k0 = kt * B3_BK;
        if (APL) {
            const unsigned char* pbase = Aimg + (size_t)k0 * 2;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) rp[pl] = *reinterpret_cast<const gi_u32x4*>(pbase + pl * aplane + ap_off);
        } else if (STEADY) {
#ifdef B3_LAB_NO_GLOAD                     // lab, TIMING ONLY: the steady loop re-uses what the prologue loaded
            return;
#endif
            const char* abase = (const char*)p.A + (size_t)k0 * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = *(const v4f_u*)(abase + a_voff[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                ra[i] = gi_load4_raw((const float*)((const char*)p.A + a_off[i]), k0 + 4 * c4, a_cmax);
        }
        if (BFP) {
            if (STEADY) {
                const char* fbase = (const char*)p.B + (size_t)k0 * 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) rf[i] = *(const v4f_u*)(fbase + bf_voff[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    rf[i] = gi_load4_raw((const float*)((const char*)p.B + bf_off[i]), k0 + 4 * c4, bf_cmax);
            }
        } else {
            const unsigned char* bbase = Bimg + (size_t)k0 * 2;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) rb[pl] = *reinterpret_cast<const gi_u32x4*>(bbase + pl * bplane + b_off);
        }
    };
    auto sstore = [&](auto steady_c, int kt, int buf, v4f (&ra)[2], gi_u32x4 (&rb)[3], gi_u32x4 (&rp)[3], v4f (&rf)[2]) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;
        unsigned char* As = smem + buf * BUF;
        unsigned char* Bs = As + OPER;
        const int k0 = kt * B3_BK;
        if (APL) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<gi_u32x4*>(As + pl * B3_PLANE + ap_lds) = rp[pl];
        } else
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v4f v = ra[i];
            if (!STEADY) v = gi_fix4(v, k0 + 4 * c4, a_cmax, K, true);
            gi_u32x2 w0, w1, w2;
            unsigned x0, x1, x2, y0, y1, y2;
            if (X2) {
#ifdef B3_LAB_NO_SPLIT                     // lab, TIMING ONLY: the raw bits instead of the two planes
                x0 = __builtin_bit_cast(unsigned, v.x); x1 = __builtin_bit_cast(unsigned, v.y);
                y0 = __builtin_bit_cast(unsigned, v.z); y1 = __builtin_bit_cast(unsigned, v.w);
#else
                rowmax[i] = fmaxf(fmaxf(rowmax[i], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                gx_split2(v.x, v.y, sa, x0, x1);
                gx_split2(v.z, v.w, sa, y0, y1);
#endif
            } else {
                b3_split2(v.x, v.y, x0, x1, x2);
                b3_split2(v.z, v.w, y0, y1, y2);
                w2.x = x2; w2.y = y2;
                *reinterpret_cast<gi_u32x2*>(As + 2 * B3_PLANE + a_lds[i]) = w2;
            }
            w0.x = x0; w0.y = y0; w1.x = x1; w1.y = y1;
            *reinterpret_cast<gi_u32x2*>(As + a_lds[i]) = w0;
            *reinterpret_cast<gi_u32x2*>(As + B3_PLANE + a_lds[i]) = w1;
        }
        if (BFP) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                v4f v = rf[i];
                if (!STEADY) v = gi_fix4(v, k0 + 4 * c4, bf_cmax, K, true);
                gi_u32x2 w0, w1, w2;
                unsigned x0, x1, x2, y0, y1, y2;
                if (X2) {
#ifdef B3_LAB_NO_SPLIT
                    x0 = __builtin_bit_cast(unsigned, v.x); x1 = __builtin_bit_cast(unsigned, v.y);
                    y0 = __builtin_bit_cast(unsigned, v.z); y1 = __builtin_bit_cast(unsigned, v.w);
#else
                    gx_split2(v.x, v.y, sb, x0, x1);
                    gx_split2(v.z, v.w, sb, y0, y1);
#endif
                } else {
                    b3_split2(v.x, v.y, x0, x1, x2);
                    b3_split2(v.z, v.w, y0, y1, y2);
                    w2.x = x2; w2.y = y2;
                    *reinterpret_cast<gi_u32x2*>(Bs + 2 * B3_PLANE + bf_lds[i]) = w2;
                }
                w0.x = x0; w0.y = y0; w1.x = x1; w1.y = y1;
                *reinterpret_cast<gi_u32x2*>(Bs + bf_lds[i]) = w0;
                *reinterpret_cast<gi_u32x2*>(Bs + B3_PLANE + bf_lds[i]) = w1;
            }
        } else {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<gi_u32x4*>(Bs + pl * B3_PLANE + b_lds) = rb[pl];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * BUF;
        const unsigned char* Bs = As + OPER;
        gi_bf16x8 af[2][3], bf[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ra_ = wm * 64 + t * 32 + l31, rb_ = wn * 64 + t * 32 + l31;
            const int oa = ra_ * B3_ROWB + 16 * (lhi ^ ((ra_ >> 3) & 1)), ob = rb_ * B3_ROWB + 16 * (lhi ^ ((rb_ >> 3) & 1));
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                af[t][pl] = *reinterpret_cast<const gi_bf16x8*>(As + pl * B3_PLANE + oa);
                bf[t][pl] = *reinterpret_cast<const gi_bf16x8*>(Bs + pl * B3_PLANE + ob);
            }
        }
        if (X2) {                                    // a2 b1 + a1 b2 + a1 b1 on the f16 MFMA (smallest terms first)
            constexpr int XA[3] = {1, 0, 0}, XB[3] = {0, 1, 0};
#ifdef B3_LAB_NO_MFMA                      // lab, TIMING ONLY: one MFMA per wave and k tile instead of twelve
#define B3_U(x) __builtin_bit_cast(gi_u32x4, x)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                __builtin_bit_cast(gx_f16x8, B3_U(af[0][0]) ^ B3_U(af[1][0]) ^ B3_U(af[0][1]) ^ B3_U(af[1][1])),
                __builtin_bit_cast(gx_f16x8, B3_U(bf[0][0]) ^ B3_U(bf[1][0]) ^ B3_U(bf[0][1]) ^ B3_U(bf[1][1])), acc[0][0], 0, 0, 0);
#undef B3_U
            return;
#endif
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(gx_f16x8, af[t][XA[term]]), __builtin_bit_cast(gx_f16x8, bf[u][XB[term]]),
                            acc[t][u], 0, 0, 0);
            return;
        }
        // smallest terms first; four independent accumulators between two MFMAs on the same one
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#ifdef B3_X2_TIMING                       // lab, TIMING ONLY: three products instead of six
        constexpr int T0 = 3;
#else
        constexpr int T0 = 0;
#endif
#pragma unroll
        for (int term = T0; term < 6; ++term)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][TA[term]], bf[u][TB[term]],
                                                                        acc[t][u], 0, 0, 0);
    };

    // ---- k loop (nk is even: the image is padded to 32): loads two tiles ahead, one barrier per tile ------------
    const std::true_type ST{};
    const std::false_type GEN{};
    const int n_full = K / B3_BK;                                // k tiles that are full in k
#ifndef B3_LAB_NO_PROLOGUE                  // lab, TIMING ONLY (with B3_LAB_NO_LOOP)
    gload(GEN, 0, ra0, rb0, rp0, rf0);
    gload(GEN, min(1, nk - 1), ra1, rb1, rp1, rf1);
    sstore(GEN, 0, 0, ra0, rb0, rp0, rf0);
    __syncthreads();
#endif
    int kt = 0;
#ifdef B3_LAB_NO_LOOP                      // lab, TIMING ONLY: prologue + epilogue
    kt = nk;
#endif
    for (; kt + 3 < n_full; kt += 2) {                           // every tile touched is full: kt + 1 .. kt + 3
        gload(ST, kt + 2, ra0, rb0, rp0, rf0);
        B3_SB();
        compute(0);
        sstore(ST, kt + 1, 1, ra1, rb1, rp1, rf1);
        __syncthreads();
        gload(ST, kt + 3, ra1, rb1, rp1, rf1);
        B3_SB();
        compute(1);
        sstore(ST, kt + 2, 0, ra0, rb0, rp0, rf0);
        __syncthreads();
    }
    for (; kt < nk; kt += 2) {                                   // the last one or two pairs: generic
        gload(GEN, min(kt + 2, nk - 1), ra0, rb0, rp0, rf0);               // (past the end: re-load the last tile, unused)
        compute(0);
        sstore(GEN, kt + 1, 1, ra1, rb1, rp1, rf1);
        __syncthreads();
        gload(GEN, min(kt + 3, nk - 1), ra1, rb1, rp1, rf1);
        compute(1);
        sstore(GEN, min(kt + 2, nk - 1), 0, ra0, rb0, rp0, rf0);
        __syncthreads();
    }

    // ---- fp16x2 dynamic-range guard: rows of A that the per-tensor scale leaves with too few bits -----------------
    // A row whose largest SCALED magnitude is below 2^-11 (more than 2^24 below the tensor's maximum, which scales into
    // [2^13, 2^14)) keeps fewer than ~14 significant bits: h1 is still a normal fp16, the residual h2 already rounds to
    // fp16's subnormal quantum 2^-24.  Counted once per launch (by the workgroups of the first column tile); zero rows
    // are exact and not counted.
    if (X2 && p.x2_guard && bx == 0) {
        int n_low = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float m = rowmax[i];
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));                     // the four lanes (k chunks) of a row
            const bool real_row = m0 + (tid >> 2) + 64 * i < m_end;
            n_low += (c4 == 0 && real_row && m > 0.f && m * sa < 0x1p-11f) ? 1 : 0;
        }
        if (n_low) {
            atomicAdd(p.x2_guard, n_low);
            if (p.x2_guard_host) *reinterpret_cast<volatile int*>(p.x2_guard_host) = 1;
        }
    }

#ifdef B3_LAB_NO_EPI                       // lab, TIMING ONLY: one store per thread instead of the epilogue
    {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
        if (t == 123.456f) p.C[tid] = t;
        return;
    }
#endif
    // ---- epilogue (C/D layout of a 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) --
    const int flags = EPI == 1 ? (GI_EPI_BIAS | GI_EPI_SELU) : (EPI == 2 ? GI_EPI_DSELU : p.flags);
    const bool need_act = (flags & (GI_EPI_DSELU | GI_EPI_MULACT)) != 0;
    const bool need_c = (flags & GI_EPI_ACCUM) != 0;
    float amax = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = n0 + wn * 64 + u * 32 + l31;
            const bool col_ok = col < p.N;
            const int colc = col_ok ? col : p.N - 1;
            const int row0 = m0 + wm * 64 + t * 32 + 4 * lhi;
#ifdef B3_LAB_NO_BIAS                       // lab, TIMING ONLY
            const float bv = 0.25f;
#else
            const float bv = (flags & GI_EPI_BIAS) ? p.bias[colc] : 0.f;
#endif
            float av[16], cv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {                        // every load of the block before the first store
                const int row = min(row0 + 8 * (r >> 2) + (r & 3), m_end - 1);
#ifdef B3_LAB_NO_ACT                        // lab, TIMING ONLY
                if (need_act) av[r] = 0.5f + r;
#else
                if (need_act) av[r] = p.act[(long long)row * p.ldact + colc];
#endif
                if (need_c) cv[r] = p.C[(long long)row * p.ldc + colc];
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = X2 ? (acc[t][u][r] * ia) * ib + bv : acc[t][u][r] + bv;
#ifndef B3_LAB_NO_SELU                      // lab, TIMING ONLY
                if (flags & GI_EPI_SELU) x = gi_selu(x);
#endif
                if (flags & GI_EPI_DSELU) x *= gi_selu_grad(av[r]);
                if (flags & GI_EPI_MULACT) x *= av[r];
                if (flags & GI_EPI_ACCUM) x += cv[r];
                v[r] = x;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 8 * (r >> 2) + (r & 3);
                const bool ok = col_ok & (row < m_end);
#ifdef B3_LAB_NO_STORE                     // lab, TIMING ONLY
                float* dst = (ok && v[r] == 123.456f) ? p.C + (long long)row * p.ldc + col : b3_sink + tid;
#else
                float* dst = ok ? p.C + (long long)row * p.ldc + col : b3_sink + tid;
#endif
                *dst = v[r];
                amax = fmaxf(amax, ok ? fabsf(v[r]) : 0.f);
            }
        }
    }
    if (p.c_amax) gx_amax_publish(amax, p.c_amax);       // for the fp16x2 launches that read this tensor next
}

}  // namespace

extern "C" long long gi_bf3_image_elems(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return GI_EINVAL;
    return 3LL * rows * b3_r32(cols);
}

extern "C" int gi_bf3_pack(const gi_bf3_pack_desc* descs, int n, void* stream) {
    (void)hipGetLastError();
    if (!descs || n < 1 || n > GI_BF3_PACK_MAX) return GI_EINVAL;
    B3PackArgs a;
    memset(&a, 0, sizeof(a));
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        const gi_bf3_pack_desc& d = descs[i];
        if (!d.W || !d.image || d.rows <= 0 || d.cols <= 0 || d.ld < (d.transpose ? d.rows : d.cols))
            return GI_EINVAL;
        if (((uintptr_t)d.image & 15) != 0) return GI_EINVAL;
        a.d[i] = d;
        a.start[i] = total;
        total += (long long)d.rows * (b3_r32(d.cols) / 2);
    }
    a.start[n] = total; a.n = n;
    if (total > 0x7fffffffLL * 256) return GI_ELIMIT;
    hipLaunchKernelGGL(gi_bf3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return gi_launch_status();
}

// gi_gemm_batch hands launches whose problems all carry GI_GEMM_BF3 to this launcher
int gi_gemm_bf3_launch(const gi_gemm_params* probs, int n, void* stream) {
    if (gi_b3v_wants(probs, n)) return gi_b3v_launch(probs, n, stream);        // GI_GEMM_T128: fp16x2 weight gradients, 128 x 128 tiles
    if (gi_b3p_eligible(probs, n)) return gi_b3p_launch(probs, n, stream);     // forward / dgrad: ping-pong kernel
    if (gi_b3v_eligible(probs, n)) return gi_b3v_launch(probs, n, stream);     // 32-deep tiles, all three layouts
    B3Batch b;
    memset(&b, 0, sizeof(b));
    double flops = 0;
    int total = 0, k = 0;
    int epi = -1, napl = 0;
    for (int i = 0; i < n; ++i) napl += (probs[i].flags & GI_GEMM_BF3A) != 0;
    if (napl && napl != n) return GI_EINVAL;
    const bool apl = napl != 0;
    int nbfp = 0;
    for (int i = 0; i < n; ++i) nbfp += (probs[i].flags & GI_GEMM_BF3B_F32) != 0;
    if (nbfp && nbfp != n) return GI_EINVAL;
    const bool bfp = nbfp != 0;
    int nx2 = 0;
    for (int i = 0; i < n; ++i) nx2 += (probs[i].flags & GI_GEMM_X2) != 0;
    if (nx2 && (nx2 != n || !bfp || apl)) return GI_EINVAL;       // fp16x2: every problem, both operands plain fp32
    const bool x2 = nx2 != 0;
    for (int i = 0; i < n; ++i) if (x2 && (!probs[i].a_amax || !probs[i].b_amax)) return GI_EINVAL;
    for (int i = 0; i < n; ++i) {
        const gi_gemm_params& p = probs[i];
        if (p.a_major || p.b_major || p.ngroups || p.b_idx || p.nsplit != 1 || p.k_dev || !p.A || !p.B || !p.C)
            return GI_EINVAL;
        if (p.M < 0 || p.N <= 0 || p.K <= 0 || (!apl && p.lda < p.K)) return GI_EINVAL;
        if (apl && (p.a_idx || ((uintptr_t)p.A & 15) != 0)) return GI_EINVAL;
        // operands are addressed by 32-bit byte offsets: a gathered row index is not bounded by M, so its offset
        // cannot be checked here (round-3 advisor finding) — the bf16x3 path takes no row gather (the model's
        // launches have none); and the 16-byte k chunks clamp into [0, 4): rows shorter than 4 floats need padding
        if (p.a_idx) return GI_EINVAL;
        if (p.K < 4 && ((!apl && p.lda < 4) || (bfp && p.ldb < 4))) return GI_EINVAL;
        if (!bfp && ((uintptr_t)p.B & 15) != 0) return GI_EINVAL;
        if (bfp && p.ldb < p.K) return GI_EINVAL;
        const int f = p.flags & ~(GI_GEMM_BF3 | GI_GEMM_BF3A | GI_GEMM_BF3B_F32 | GI_GEMM_X2);
        if (f & ~(GI_EPI_BIAS | GI_EPI_SELU | GI_EPI_DSELU | GI_EPI_ACCUM | GI_EPI_MULACT)) return GI_EINVAL;
        if ((f & GI_EPI_BIAS) && !p.bias) return GI_EINVAL;
        if ((f & (GI_EPI_DSELU | GI_EPI_MULACT)) && !p.act) return GI_EINVAL;
        const long long lim = 0xffffffffLL / 4;
        if (!p.a_idx && (long long)p.M * p.lda > lim) return GI_ELIMIT;
        if ((long long)p.N * b3_r32(p.K) * 2 > 0xffffffffLL || (bfp && (long long)p.N * p.ldb > lim)) return GI_ELIMIT;
        const int e = f == (GI_EPI_BIAS | GI_EPI_SELU) ? 1 : (f == GI_EPI_DSELU ? 2 : 0);
        epi = (epi < 0 || epi == e) ? e : 0;
        if (p.M == 0) continue;
        b.p[k] = p; b.p[k].flags = f;
        b.gx[k] = gi_cdiv(p.N, B3_BN);
        b.start[k] = total;
        total += b.gx[k] * gi_cdiv(p.M, B3_BM);
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
        ++k;
    }
    if (k == 0) return 0;
    b.start[k] = total; b.n = k; b.total = total;
    {
        bool bounded = false;
        for (int i = 0; i < k; ++i) bounded |= b.p[i].m_dev != nullptr;
        // measured on the node-level hidden-layer launch (684 tiles): 76.7 -> 73.0 us forward, 84.0 -> 82.1 dgrad
        b.remap = (total >= 512 && !bounded) ? 1 : 0;       // (bounded: the order would be over the bound's tiles)
    }
    hipStream_t st = (hipStream_t)stream;
    GiProfScope prof(st, GI_PROF_GEMM | (x2 ? GI_PROF_PIPE_X2 : GI_PROF_PIPE_BF3), flops);
    gi_gemm_log_launch(x2 ? ((b.p[0].flags & GI_EPI_BIAS) ? "x0" : "x1") : ((b.p[0].flags & GI_EPI_BIAS) ? "b0" : "b1"), b.p, k,
                       total, flops);
#define GI_B3_LAUNCH(E, A, F) hipLaunchKernelGGL((gi_gemm_bf3_kernel<E, A, F>), dim3(total), dim3(256), 0, st, b)
#define GI_X2_LAUNCH(E) hipLaunchKernelGGL((gi_gemm_bf3_kernel<E, false, true, true>), dim3(total), dim3(256), 0, st, b)
    if (x2) {
        if (epi == 1) GI_X2_LAUNCH(1); else if (epi == 2) GI_X2_LAUNCH(2); else GI_X2_LAUNCH(0);
    } else if (bfp) {                                     // (forward weights as stored: own epilogue or run-time flags)
        if (apl) return GI_EINVAL;
        if (epi == 1) GI_B3_LAUNCH(1, false, true); else GI_B3_LAUNCH(0, false, true);
    } else if (apl) {
        if (epi == 1) GI_B3_LAUNCH(1, true, false); else if (epi == 2) GI_B3_LAUNCH(2, true, false); else GI_B3_LAUNCH(0, true, false);
    } else {
        if (epi == 1) GI_B3_LAUNCH(1, false, false); else if (epi == 2) GI_B3_LAUNCH(2, false, false); else GI_B3_LAUNCH(0, false, false);
    }
#undef GI_B3_LAUNCH
#undef GI_X2_LAUNCH
    return gi_launch_status();
}
