// fp32 MFMA GEMM family for the GGNN MLP stacks (gfx950).
//
// C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ), fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32
// (exact fp32: bitwise a k-ordered fmaf chain — the reference's 1e-4 fp32 parity bar rules out
// bf16 and gfx950 has no xf32).  Block = 256 threads = 4 waves in a 2x2 grid; each wave owns
// TM x TN accumulator tiles of 32x32; block tile (64 TM) x (64 TN) x 32.
//
// Operand storage, per operand:
//   "contig": stored [row][reduction], reduction contiguous  -> LDS [rows][32+4], fragments by
//             ds_read_b128 (conflict free: row stride 36 floats puts 16 rows on 16 distinct
//             16-byte slots of the 256-byte bank row)
//   "major" : stored [reduction][row]                          -> LDS [32][rows+4], fragments by
//             ds_read_b32 (a half-wave reads 32 consecutive floats)
// Inside one 8-deep reduction group, MFMA j (0..3) consumes reduction index  j + 4*(lane>>5);
// both operands use that same map, so any storage combination multiplies matching k.
//
//   forward : A = X   contig (optionally row-gathered by a_idx),  B = W [out,in] contig
//   dgrad   : A = dZ  contig,                                     B = W [out,in] major
//   wgrad   : A = dZ  major,  B = X major (+ ones column -> bias gradient), reduction over rows,
//             split into nsplit slabs (deterministic; summed later by gi_reduce_slabs)
//
// Pipeline: global -> registers for tile t+1 is issued before the MFMAs of tile t, written to the
// other LDS buffer after them; one __syncthreads per 32-deep tile.
#include "gi_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4f_u __attribute__((aligned(4)));     // global rows are only guaranteed 4-byte aligned

template <int TM, int TN, bool A_MAJOR, bool B_MAJOR>
__global__ __launch_bounds__(256) void gi_gemm_kernel(const gi_gemm_params p) {
    constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
    constexpr int A_LD = A_MAJOR ? BM + 4 : BK + 4;
    constexpr int A_ROWS = A_MAJOR ? BK : BM;
    constexpr int B_LD = B_MAJOR ? BN + 4 : BK + 4;
    constexpr int B_ROWS = B_MAJOR ? BK : BN;
    constexpr int A_SZ = A_ROWS * A_LD, B_SZ = B_ROWS * B_LD;
    constexpr int NA = 2 * TM, NB = 2 * TN;          // float4 staged per thread per tile
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];
    float* const As = smem;
    float* const Bs = smem + 2 * A_SZ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- group / split resolution (block-uniform) -------------------------------------------
    const bool splitk = (p.flags & GI_GEMM_SPLITK) != 0;
    int g = blockIdx.z, s = 0;
    if (splitk) { g = blockIdx.z / p.nsplit; s = blockIdx.z - g * p.nsplit; }
    const float* __restrict__ Ap = p.A;
    const float* __restrict__ Bp = (p.ngroups && !splitk) ? p.Bg[g] : p.B;
    const float* __restrict__ biasp = (p.ngroups && !splitk) ? p.biasg[g] : p.bias;
    float* __restrict__ Cp = (p.ngroups && splitk) ? p.Cg[g] : p.C;

    int m_begin = 0, m_end = p.M, k_begin = 0, k_end = p.K;
    if (p.grp_off) {
        const int lo = p.grp_off[g], hi = p.grp_off[g + 1];
        if (splitk) { k_begin = lo; k_end = hi; } else { m_begin = lo; m_end = hi; }
    }
    if (splitk) {
        const int len = k_end - k_begin;
        const int chunk = (((len + p.nsplit - 1) / p.nsplit) + 31) & ~31;
        const int kb = k_begin + s * chunk;
        k_end = min(kb + chunk, k_end);
        k_begin = kb;
        Cp += (long long)s * p.c_split_stride;
    }
    const int m0 = m_begin + blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    if (m0 >= m_end) return;
    const int bcols = (p.ones_col >= 0) ? p.ones_col : p.N;     // real stored columns of a major B

    // ---- per-thread staging coordinates -----------------------------------------------------
    // contig operand: 8 float4 per 32-wide row -> c4 = tid&7, row = (tid>>3) + 32*i
    // major operand : BX/4 float4 per stored row -> c4 = tid % (BX/4), red = tid/(BX/4) + i*RP
    const int cc4 = tid & 7, crow = tid >> 3;
    constexpr int A_C4 = BM / 4, A_RP = 256 / A_C4;
    constexpr int B_C4 = BN / 4, B_RP = 256 / B_C4;
    const int a_mc4 = tid % A_C4, a_mr = tid / A_C4;
    const int b_mc4 = tid % B_C4, b_mr = tid / B_C4;

    long long a_off[NA], b_off[NB];                    // element offsets of fixed (contig) rows, -1 = out of range
    if (!A_MAJOR) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = m0 + crow + 32 * i;
            a_off[i] = -1;
            if (row < m_end) a_off[i] = (long long)(p.a_idx ? p.a_idx[row] : row) * p.lda;
        }
    }
    if (!B_MAJOR) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = n0 + crow + 32 * i;
            b_off[i] = (row < p.N) ? (long long)row * p.ldb : -1;
        }
    }

    v4f ra[NA], rb[NB];

    auto gload = [&](int k0) {
        if (!A_MAJOR) {
            const int kk = k0 + 4 * cc4;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                v4f v = {0.f, 0.f, 0.f, 0.f};
                if (a_off[i] >= 0) {
                    const float* src = Ap + a_off[i] + kk;
                    if (kk + 4 <= k_end) v = *(const v4f_u*)src;
                    else {
                        if (kk < k_end) v.x = src[0];
                        if (kk + 1 < k_end) v.y = src[1];
                        if (kk + 2 < k_end) v.z = src[2];
                    }
                }
                ra[i] = v;
            }
        } else {
            const int col = m0 + 4 * a_mc4;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int red = k0 + a_mr + i * A_RP;
                v4f v = {0.f, 0.f, 0.f, 0.f};
                if (red < k_end) {
                    const float* src = Ap + (long long)red * p.lda + col;
                    if (col + 4 <= p.M) v = *(const v4f_u*)src;
                    else {
                        if (col < p.M) v.x = src[0];
                        if (col + 1 < p.M) v.y = src[1];
                        if (col + 2 < p.M) v.z = src[2];
                    }
                }
                ra[i] = v;
            }
        }
        if (!B_MAJOR) {
            const int kk = k0 + 4 * cc4;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                v4f v = {0.f, 0.f, 0.f, 0.f};
                if (b_off[i] >= 0) {
                    const float* src = Bp + b_off[i] + kk;
                    if (kk + 4 <= k_end) v = *(const v4f_u*)src;
                    else {
                        if (kk < k_end) v.x = src[0];
                        if (kk + 1 < k_end) v.y = src[1];
                        if (kk + 2 < k_end) v.z = src[2];
                    }
                }
                rb[i] = v;
            }
        } else {
            const int col = n0 + 4 * b_mc4;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int red = k0 + b_mr + i * B_RP;
                v4f v = {0.f, 0.f, 0.f, 0.f};
                if (red < k_end) {
                    const long long srow = p.b_idx ? p.b_idx[red] : red;
                    const float* src = Bp + srow * p.ldb + col;
                    if (col + 4 <= bcols) v = *(const v4f_u*)src;
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = col + j;
                            v[j] = (c < bcols) ? src[j] : ((c == p.ones_col) ? 1.f : 0.f);
                        }
                    }
                }
                rb[i] = v;
            }
        }
    };

    auto sstore = [&](int buf) {
        float* a = As + buf * A_SZ;
        float* b = Bs + buf * B_SZ;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!A_MAJOR) *(v4f*)&a[(crow + 32 * i) * A_LD + 4 * cc4] = ra[i];
            else *(v4f*)&a[(a_mr + i * A_RP) * A_LD + 4 * a_mc4] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (!B_MAJOR) *(v4f*)&b[(crow + 32 * i) * B_LD + 4 * cc4] = rb[i];
            else *(v4f*)&b[(b_mr + i * B_RP) * B_LD + 4 * b_mc4] = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const float* a = As + buf * A_SZ;
        const float* b = Bs + buf * B_SZ;
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            float af[TM][4], bf[TN][4];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int row = wm * 32 * TM + t * 32 + l31;
                if (!A_MAJOR) {
                    const v4f v = *(const v4f*)&a[row * A_LD + k8 * 8 + 4 * lhi];
                    af[t][0] = v.x; af[t][1] = v.y; af[t][2] = v.z; af[t][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[t][j] = a[(k8 * 8 + j + 4 * lhi) * A_LD + row];
                }
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int row = wn * 32 * TN + t * 32 + l31;
                if (!B_MAJOR) {
                    const v4f v = *(const v4f*)&b[row * B_LD + k8 * 8 + 4 * lhi];
                    bf[t][0] = v.x; bf[t][1] = v.y; bf[t][2] = v.z; bf[t][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bf[t][j] = b[(k8 * 8 + j + 4 * lhi) * B_LD + row];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][j], bf[tn][j],
                                                                           acc[tm][tn], 0, 0, 0);
        }
    };

    // ---- main loop ----------------------------------------------------------------------------
    const int nk = (k_end > k_begin) ? (k_end - k_begin + BK - 1) / BK : 0;
    if (nk > 0) {
        gload(k_begin);
        sstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) gload(k_begin + (kt + 1) * BK);
        compute(kt & 1);
        if (more) sstore((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int flags = p.flags;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * 32 * TN + tn * 32 + l31;
            if (col >= p.N) continue;
            const float bv = (flags & GI_EPI_BIAS) ? biasp[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row >= m_end) continue;
                float v = acc[tm][tn][r] + bv;
                if (flags & GI_EPI_SELU) v = gi_selu(v);
                if (flags & GI_EPI_DSELU) v *= gi_selu_grad(p.act[(long long)row * p.ldact + col]);
                float* dst = Cp + (long long)row * p.ldc + col;
                if (flags & GI_EPI_ACCUM) v += *dst;
                *dst = v;
            }
        }
    }
}

template <int TM, int TN>
static int launch_tile(const gi_gemm_params& p, dim3 grid, hipStream_t st) {
    if (!p.a_major && !p.b_major)
        hipLaunchKernelGGL((gi_gemm_kernel<TM, TN, false, false>), grid, dim3(256), 0, st, p);
    else if (!p.a_major && p.b_major)
        hipLaunchKernelGGL((gi_gemm_kernel<TM, TN, false, true>), grid, dim3(256), 0, st, p);
    else if (p.a_major && p.b_major)
        hipLaunchKernelGGL((gi_gemm_kernel<TM, TN, true, true>), grid, dim3(256), 0, st, p);
    else
        return GI_EINVAL;
    return gi_launch_status();
}

extern "C" int gi_gemm(const gi_gemm_params* pp, void* stream) {
    if (!pp) return GI_EINVAL;
    const gi_gemm_params& p = *pp;
    if (p.M < 0 || p.N <= 0 || p.K < 0 || p.nsplit < 1 || p.ngroups < 0 ||
        p.ngroups > GI_MAX_GROUPS)
        return GI_EINVAL;
    if ((p.a_idx && p.a_major) || (p.b_idx && !p.b_major)) return GI_EINVAL;
    if (p.ones_col >= 0 && (!p.b_major || p.ones_col != p.N - 1)) return GI_EINVAL;
    const bool splitk = (p.flags & GI_GEMM_SPLITK) != 0;
    if (p.ngroups && !p.grp_off) return GI_EINVAL;
    if (!splitk && p.nsplit != 1) return GI_EINVAL;
    const int BM = 64 * p.tm, BN = 64 * p.tn;
    const int rows = splitk ? p.M : (p.ngroups ? p.max_group_rows : p.M);
    if (rows <= 0) return 0;
    const int groups = p.ngroups ? p.ngroups : 1;
    dim3 grid(gi_cdiv(p.N, BN), gi_cdiv(rows, BM), splitk ? groups * p.nsplit : groups);
    if (grid.y > 65535u || grid.z > 65535u) return GI_ELIMIT;
    hipStream_t st = (hipStream_t)stream;
    // useful flops of this launch (real dims; for grouped / split launches M resp. K is the total)
    GiProfScope prof(st, GI_PROF_GEMM, 2.0 * (double)p.M * (double)p.N * (double)p.K);
    if (p.tm == 1 && p.tn == 1) return launch_tile<1, 1>(p, grid, st);
    if (p.tm == 1 && p.tn == 2) return launch_tile<1, 2>(p, grid, st);
    if (p.tm == 2 && p.tn == 2) return launch_tile<2, 2>(p, grid, st);
    return GI_EINVAL;
}
