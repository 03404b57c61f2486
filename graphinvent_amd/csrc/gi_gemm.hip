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
#include <stdlib.h>
#include <string.h>

#include <stdio.h>
#include <stdlib.h>

#include "gi_common.h"

#include "gi_mfma.h"
#include <type_traits>

__device__ float gi_store_sink[256];               // where out-of-range lanes of edge tiles store

template <int TM, int TN, bool A_MAJOR, bool B_MAJOR>
__device__ __forceinline__ void gi_gemm_body(const gi_gemm_params& p, const int bx_in,
                                             const int by_in, const int bz, const int gx,
                                             const int gy) {
    // XCD-aware tile order (flag bit 5, set by the host for launches of many workgroups, see
    // remap_min_blocks): the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs
    // (private L2 each); the remap makes one XCD walk consecutive tiles so the column tiles sharing
    // an A row panel hit the same L2 (bijective for any grid size).
    int bx = bx_in, by = by_in;
    if (p.flags & 32) {
        const int T = gx * gy, id = bx_in + gx * by_in;
        const int q = T >> 3, r = T & 7, xcd = id & 7, i = id >> 3;
        const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        by = nid / gx;
        bx = nid - by * gx;
    }
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
    int g = bz, s = 0, nsp = p.nsplit;
    if (splitk) {
        g = 0; s = bz;
        if (p.ngroups) {                                  // per-group slab counts (work-proportional)
            while (g < p.ngroups - 1 && s >= p.gsplit[g]) { s -= p.gsplit[g]; ++g; }
            nsp = p.gsplit[g];
        }
    }
    const float* __restrict__ Ap = p.A;
    const float* __restrict__ Bp = (p.ngroups && !splitk) ? p.Bg[g] : p.B;
    const float* __restrict__ biasp = (p.ngroups && !splitk) ? p.biasg[g] : p.bias;
    float* Cp = (p.ngroups && splitk) ? p.Cg[g] : p.C;

    int m_begin = 0, m_end = p.M, k_begin = 0, k_end = p.K;
    if (p.grp_off) {
        const int lo = p.grp_off[g], hi = p.grp_off[g + 1];
        if (splitk) { k_begin = lo; k_end = hi; } else { m_begin = lo; m_end = hi; }
    }
    if (splitk) {
        const int len = k_end - k_begin;
        const int chunk = (((len + nsp - 1) / nsp) + 31) & ~31;
        const int kb = k_begin + s * chunk;
        k_end = min(kb + chunk, k_end);
        k_begin = kb;
        Cp += (long long)s * p.c_split_stride;
    }
    const int split_idx = s, n_splits = nsp;
    const int m0 = m_begin + by * BM;
    const int n0 = bx * BN;
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

    long long a_off[NA], b_off[NB];        // element offsets of the fixed (contig) rows; invalid rows -> row 0
    bool a_ok[NA], b_ok[NB];
    if (!A_MAJOR) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = m0 + crow + 32 * i;
            a_ok[i] = row < m_end;
            const int rr = a_ok[i] ? row : m0;
            a_off[i] = (long long)(p.a_idx ? p.a_idx[rr] : rr) * p.lda;
        }
    }
    if (!B_MAJOR) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = n0 + crow + 32 * i;
            b_ok[i] = row < p.N;
            b_off[i] = (long long)(b_ok[i] ? row : 0) * p.ldb;
        }
    }

    v4f ra0[NA], rb0[NB], ra1[NA], rb1[NB];    // two register stages: k tiles t+1 and t+2 in flight

    // Only the REDUCTION dimension needs zero fill (garbage there would reach valid outputs).  Along
    // the output dimensions out-of-range rows/columns are merely clamped to readable addresses:
    // whatever they hold only feeds output elements the epilogue discards.  So every tile that is
    // full in k stores the raw vectors ("fast"); the last, partial k tile takes the fix-up path.
    const int a_cmax = A_MAJOR ? ((p.lda >= ((p.M + 3) & ~3)) ? ((p.M + 3) & ~3) - 4 : p.M - 4) : k_end - 4;
    const int b_cmax = B_MAJOR ? ((p.ldb >= ((bcols + 3) & ~3)) ? ((bcols + 3) & ~3) - 4 : bcols - 4)
                               : k_end - 4;
    const bool a_fast = A_MAJOR ? (p.lda >= ((p.M + 3) & ~3)) : true;
    const bool b_fast = B_MAJOR ? (p.ldb >= ((bcols + 3) & ~3)) : true;

    // raw loads only (nothing consumes the data here); A half and B half are issued separately so
    // they can be spread between MFMA groups
    auto gload_a = [&](v4f (&ra)[NA], int k0) {
        if (!A_MAJOR) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = gi_load4_raw(Ap + a_off[i], k0 + 4 * cc4, a_cmax);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int red = min(k0 + a_mr + i * A_RP, k_end - 1);
                ra[i] = gi_load4_raw(Ap + (long long)red * p.lda, m0 + 4 * a_mc4, a_cmax);
            }
        }
    };
    // BIDX (compile time): rows of a major B gathered through b_idx.  The index load is a dependent
    // load in front of the data load (it drains the load queue once per tile), so the body exists
    // twice and only the gathered problems (first layer of the message stacks' weight gradients)
    // run the BIDX version.
    auto gload_b = [&](v4f (&rb)[NB], int k0, auto bidx) {
        constexpr bool BIDX = decltype(bidx)::value;
        if (!B_MAJOR) {
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = gi_load4_raw(Bp + b_off[i], k0 + 4 * cc4, b_cmax);
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int red = min(k0 + b_mr + i * B_RP, k_end - 1);
                const long long srow = BIDX ? p.b_idx[red] : red;
                rb[i] = gi_load4_raw(Bp + srow * p.ldb, n0 + 4 * b_mc4, b_cmax);
            }
        }
    };

    // (fix-up of the last / padding k tile) + LDS write, A half and B half
    auto sstore_a = [&](v4f (&ra)[NA], int buf, int k0, const bool steady) {
        float* a = As + buf * A_SZ;
        const bool full_k = k0 + BK <= k_end;        // block-uniform
        if (steady || (full_k && a_fast)) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (!A_MAJOR) *(v4f*)&a[(crow + 32 * i) * A_LD + 4 * cc4] = ra[i];
                else *(v4f*)&a[(a_mr + i * A_RP) * A_LD + 4 * a_mc4] = ra[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (!A_MAJOR) {
                    *(v4f*)&a[(crow + 32 * i) * A_LD + 4 * cc4] =
                        gi_fix4(ra[i], k0 + 4 * cc4, a_cmax, k_end, a_ok[i]);
                } else {
                    const bool ok = k0 + a_mr + i * A_RP < k_end;
                    *(v4f*)&a[(a_mr + i * A_RP) * A_LD + 4 * a_mc4] =
                        gi_fix4(ra[i], m0 + 4 * a_mc4, a_cmax, p.M, ok);
                }
            }
        }
    };
    auto sstore_b = [&](v4f (&rb)[NB], int buf, int k0, const bool steady) {
        float* b = Bs + buf * B_SZ;
        const bool full_k = k0 + BK <= k_end;
        if (steady || (full_k && b_fast)) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (!B_MAJOR) {
                    *(v4f*)&b[(crow + 32 * i) * B_LD + 4 * cc4] = rb[i];
                } else {
                    const int col = n0 + 4 * b_mc4;
                    v4f v = rb[i];                   // bias-gradient column of wgrad (-1 never matches)
                    v.x = (col == p.ones_col) ? 1.f : v.x;
                    v.y = (col + 1 == p.ones_col) ? 1.f : v.y;
                    v.z = (col + 2 == p.ones_col) ? 1.f : v.z;
                    v.w = (col + 3 == p.ones_col) ? 1.f : v.w;
                    *(v4f*)&b[(b_mr + i * B_RP) * B_LD + 4 * b_mc4] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (!B_MAJOR) {
                    *(v4f*)&b[(crow + 32 * i) * B_LD + 4 * cc4] =
                        gi_fix4(rb[i], k0 + 4 * cc4, b_cmax, k_end, b_ok[i]);
                } else {
                    const bool ok = k0 + b_mr + i * B_RP < k_end;
                    const int col = n0 + 4 * b_mc4;
                    v4f v = gi_fix4(rb[i], col, b_cmax, bcols, ok);
                    v.x = (ok & (col == p.ones_col)) ? 1.f : v.x;
                    v.y = (ok & (col + 1 == p.ones_col)) ? 1.f : v.y;
                    v.z = (ok & (col + 2 == p.ones_col)) ? 1.f : v.z;
                    v.w = (ok & (col + 3 == p.ones_col)) ? 1.f : v.w;
                    *(v4f*)&b[(b_mr + i * B_RP) * B_LD + 4 * b_mc4] = v;
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments of one 8-deep reduction group
    auto read_frags = [&](int buf, int k8, float (&af)[TM][4], float (&bf)[TN][4]) {
        const float* a = As + buf * A_SZ;
        const float* b = Bs + buf * B_SZ;
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
    };
    auto mma = [&](const float (&af)[TM][4], const float (&bf)[TN][4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][j], bf[tn][j],
                                                                       acc[tm][tn], 0, 0, 0);
    };

    // ---- main loop ----------------------------------------------------------------------------
    // Software pipeline, pinned with sched_barrier(0) (hipcc otherwise parks every non-MFMA
    // instruction after the tile's MFMA block, where nothing hides it — a lone wave then reaches only
    // 54 % MFMA duty, measured).  While the MFMAs of tile t run, the wave also issues
    //   - the LDS fragment reads of the NEXT 8-deep group,
    //   - the global loads of tile t+2 into the register stage that has just been drained,
    //   - the LDS writes of tile t+1 (loaded during tile t-1) into the other LDS buffer,
    // i.e. every memory instruction sits in the shadow of a 64-cycle MFMA.  The tile count is
    // rounded up to even (a tile past k_end stages zeros) so the two-stage body has no mid exit.
    //
    // STEADY = true is the straight-line body for the tiles whose successors are full in k: loads
    // (clamped, always readable) and LDS writes are unconditional, so the body is ONE basic block
    // and hipcc's wait-count pass can keep tile t+2's four loads in flight across the LDS write of
    // tile t+1 (s_waitcnt vmcnt(6) / vmcnt(4)).  With the run-time `more` / partial-tile branches of
    // the generic body it merges the states of both arms and emits vmcnt(0) before every LDS write
    // — each tile then waits out the L2 latency of loads issued two MFMA groups earlier (0.70 us per
    // tile for a lone workgroup against 0.43 us of MFMA work).  The generic body only runs the
    // last one or two tile pairs (partial / padding tiles).
    const int nk = (k_end > k_begin) ? (((k_end - k_begin + BK - 1) / BK + 1) & ~1) : 0;
    float af0[TM][4], bf0[TN][4], af1[TM][4], bf1[TN][4];
#define GI_TILE(BUF, SA, SB_, RA, RB, KSTORE, KLOAD, DO_STORE, DO_LOAD, STEADY)                 \
    {                                                                                             \
        read_frags(BUF, 0, af0, bf0);                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        mma(af0, bf0); read_frags(BUF, 1, af1, bf1); if (DO_LOAD) gload_a(RA, KLOAD);             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        mma(af1, bf1); read_frags(BUF, 2, af0, bf0); if (DO_LOAD) gload_b(RB, KLOAD, bidx);           \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        mma(af0, bf0); read_frags(BUF, 3, af1, bf1);                                              \
        if (DO_STORE) sstore_a(SA, (BUF) ^ 1, KSTORE, STEADY);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        mma(af1, bf1); if (DO_STORE) sstore_b(SB_, (BUF) ^ 1, KSTORE, STEADY);                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        __syncthreads();                                                                          \
    }
    auto run = [&](auto bidx) __attribute__((always_inline)) {
        if (nk > 0) {
            gload_a(ra0, k_begin); gload_b(rb0, k_begin, bidx);
            gload_a(ra1, k_begin + BK); gload_b(rb1, k_begin + BK, bidx);
            sstore_a(ra0, 0, k_begin, false); sstore_b(rb0, 0, k_begin, false);
        }
        __syncthreads();
        // pairs (kt, kt+1) that write tiles kt+1 and kt+2 to LDS: steady while kt+2 is a full tile
        const int n_full = (k_end - k_begin) / BK;
        const int kt_steady = (a_fast && b_fast && n_full >= 3) ? (((n_full - 3) & ~1) + 2) : 0;
        int kt = 0;
        for (; kt < kt_steady; kt += 2) {
            const int k1 = k_begin + (kt + 1) * BK, k2 = k1 + BK, k3 = k2 + BK;
            GI_TILE(0, ra1, rb1, ra0, rb0, k1, k2, true, true, true)
            GI_TILE(1, ra0, rb0, ra1, rb1, k2, k3, true, true, true)
        }
        for (; kt < nk; kt += 2) {
            const bool more = kt + 2 < nk;
            const int k1 = k_begin + (kt + 1) * BK, k2 = k1 + BK, k3 = k2 + BK;
            // tile kt from LDS buffer 0: store tile kt+1 (stage 1) -> buffer 1, fetch tile kt+2 -> stage 0
            GI_TILE(0, ra1, rb1, ra0, rb0, k1, k2, true, more, false)
            // tile kt+1 from LDS buffer 1: store tile kt+2 (stage 0) -> buffer 0, fetch tile kt+3 -> stage 1
            GI_TILE(1, ra0, rb0, ra1, rb1, k2, k3, more, more, false)
        }
    };
    if (B_MAJOR && p.b_idx) run(std::true_type{});
    else run(std::false_type{});
#undef GI_TILE

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Per 32x32 tile: ALL loads (activation for selu', old C for accumulate) are issued first from
    // clamped addresses, then the arithmetic, then ALL stores, with out-of-range lanes redirected to
    // a sink instead of branched around.  A predicated store is a basic-block boundary at which
    // hipcc drains vmcnt(0) (stores count in vmcnt on gfx950): that was one serial ~450-cycle memory
    // round trip per output element, a quarter of a K=500 wave's lifetime (tools/gemm_timing.hip).
    const int flags = p.flags;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * 32 * TN + tn * 32 + l31;
            const bool col_ok = col < p.N;
            const int colc = col_ok ? col : p.N - 1;
            const int row0 = m0 + wm * 32 * TM + tm * 32 + 4 * lhi;
            float av[16], cv[16];
            if (flags & (GI_EPI_DSELU | GI_EPI_MULACT)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(row0 + (r & 3) + 8 * (r >> 2), m_end - 1);
                    av[r] = p.act[(long long)row * p.ldact + colc];
                }
            }
            if (flags & GI_EPI_ACCUM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(row0 + (r & 3) + 8 * (r >> 2), m_end - 1);
                    cv[r] = Cp[(long long)row * p.ldc + colc];
                }
            }
            const float bv = (flags & GI_EPI_BIAS) ? biasp[colc] : 0.f;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[tm][tn][r] + bv;
                if (flags & GI_EPI_SELU) x = gi_selu(x);
                if (flags & GI_EPI_DSELU) x *= gi_selu_grad(av[r]);
                if (flags & GI_EPI_MULACT) x *= av[r];
                if (flags & GI_EPI_ACCUM) x += cv[r];
                v[r] = x;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                float* dst = (col_ok & (row < m_end)) ? Cp + (long long)row * p.ldc + col
                                                      : gi_store_sink + tid;
                *dst = v[r];
            }
        }
    }
    // ---- GI_GEMM_REDUCE: the last workgroup of this output tile sums its slabs ---------------------
    // The workgroup's slab stores are released at agent scope (L2 write-back: the 8 XCD L2s are not
    // coherent with each other), it takes a ticket on the tile's counter, and the holder of the last
    // ticket invalidates its own L2 view before it reads the other workgroups' slabs.  The sum runs
    // over the splits in index order whoever computes it.
    if constexpr (A_MAJOR && B_MAJOR) {
        if (flags & GI_GEMM_REDUCE) {
            __shared__ int last_s;
            // every wave's slab stores are in this XCD's L2 before the barrier (workgroup-scope release);
            // ONE agent-scope release (L2 write-back) by the ticket taker then covers them all
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                int* cnt = p.red_count + ((long long)g * gy + by) * gx + bx;
                const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last_s = (old == n_splits - 1) ? 1 : 0;
            }
            __syncthreads();
            if (last_s) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const float* base = Cp - (long long)split_idx * p.c_split_stride;     // slab 0
                float* dW = p.ngroups ? const_cast<float*>(p.Bg[g]) : p.red_dW;
                float* db = p.ngroups ? const_cast<float*>(p.biasg[g]) : p.red_db;
                const int wcols = (p.ones_col >= 0) ? p.ones_col : p.N;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const int col = n0 + wn * 32 * TN + tn * 32 + l31;
                        const int row0 = m0 + wm * 32 * TM + tm * 32 + 4 * lhi;
                        if (col >= p.N) continue;
#pragma unroll 4
                        for (int r = 0; r < 16; ++r) {
                            const int row = row0 + (r & 3) + 8 * (r >> 2);
                            if (row >= m_end) continue;
                            const float* src = base + (long long)row * p.ldc + col;
                            float s0 = 0.f, s1 = 0.f;
                            int j = 0;
                            for (; j + 1 < n_splits; j += 2) {
                                s0 += src[(long long)j * p.c_split_stride];
                                s1 += src[(long long)(j + 1) * p.c_split_stride];
                            }
                            if (j < n_splits) s0 += src[(long long)j * p.c_split_stride];
                            float* dst = (col < wcols) ? dW + (long long)row * p.red_ldw + col
                                                       : (db ? db + row : nullptr);
                            if (dst) *dst = p.red_accum ? *dst + (s0 + s1) : (s0 + s1);
                        }
                    }
                }
            }
        }
    }
}

template <int TM, int TN, bool A_MAJOR, bool B_MAJOR>
__global__ __launch_bounds__(256) void gi_gemm_kernel(const gi_gemm_params p) {
    gi_gemm_body<TM, TN, A_MAJOR, B_MAJOR>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x,
                                           gridDim.y);
}

// Several independent GEMMs of the same tile/layout class in ONE launch ("horizontal fusion"):
// the sibling MLPs of the readout (4 node-level stacks, 3 graph-level stacks) and the two GRU
// projections have 16..450 workgroups each — together they fill the 256 CUs and halve the number of
// launch ramps on the critical path.  Workgroup id -> (problem, x, y, z) through a prefix table.
#define GI_GEMM_BATCH_MAX 8
struct GemmBatch {
    gi_gemm_params p[GI_GEMM_BATCH_MAX];
    int start[GI_GEMM_BATCH_MAX + 1];          // first workgroup id of every problem
    int gx[GI_GEMM_BATCH_MAX], gy[GI_GEMM_BATCH_MAX];
    int n;
};

template <int TM, int TN, bool A_MAJOR, bool B_MAJOR>
__global__ __launch_bounds__(256) void gi_gemm_batch_kernel(const GemmBatch b) {
    int i = 0;
    const int id = blockIdx.x;
    while (i < b.n - 1 && id >= b.start[i + 1]) ++i;
    const int local = id - b.start[i];
    const int gx = b.gx[i], gxy = gx * b.gy[i];
    const int bz = local / gxy;
    const int rem = local - bz * gxy;
    const int by = rem / gx;
    gi_gemm_body<TM, TN, A_MAJOR, B_MAJOR>(b.p[i], rem - by * gx, by, bz, gx, b.gy[i]);
}

// XCD-aware tile order for launches of at least this many workgroups (0 = never): more than ~1.3
// full rounds of 256 CUs x 4 resident workgroups.  Measured on the training step (3 A/B pairs):
// never 2.74 ms, always 2.76-2.77 ms, >= 1300..2100 workgroups 2.70-2.71 ms.  The small launches lose
// with the remap (an XCD's share of a 1-round launch is not balanced), the big ones win (column tiles
// sharing an A row panel hit one L2).  GI_GEMM_XCD_REMAP=<min workgroups> overrides (measurements).
static int remap_min_blocks() {
    static const int v = getenv("GI_GEMM_XCD_REMAP") ? atoi(getenv("GI_GEMM_XCD_REMAP")) : 1350;
    return v;
}
static bool want_remap(long long blocks) {
    const int m = remap_min_blocks();
    return m > 0 && blocks >= m;
}

// Measurement aid (tools/gemm_launch_report.py): GI_GEMM_LOG=<file> appends one line per launch
// — layout class, workgroups, useful flops, then M,N,K,groups per problem — to correlate with a
// rocprofv3 kernel trace.
static FILE* launch_log() {
    static FILE* f = [] {
        const char* path = getenv("GI_GEMM_LOG");
        return path ? fopen(path, "a") : nullptr;
    }();
    return f;
}

static void log_launch(const gi_gemm_params* probs, int n, int blocks, double flops) {
    FILE* f = launch_log();
    if (!f) return;
    fprintf(f, "%d%d %d %d %.0f", probs[0].a_major, probs[0].b_major, n, blocks, flops);
    for (int i = 0; i < n; ++i)
        fprintf(f, " %dx%dx%d:g%d:s%d", probs[i].M, probs[i].N, probs[i].K, probs[i].ngroups,
                probs[i].nsplit);
    fputc('\n', f);
    fflush(f);
}

static int validate(const gi_gemm_params& p) {
    if (p.M < 0 || p.N <= 0 || p.K < 0 || p.nsplit < 1 || p.ngroups < 0 ||
        p.ngroups > GI_MAX_GROUPS)
        return GI_EINVAL;
    if ((p.a_idx && p.a_major) || (p.b_idx && !p.b_major)) return GI_EINVAL;
    if (p.ones_col >= 0 && (!p.b_major || p.ones_col != p.N - 1)) return GI_EINVAL;
    const bool splitk = (p.flags & GI_GEMM_SPLITK) != 0;
    if (p.ngroups && !p.grp_off) return GI_EINVAL;
    if (!splitk && p.nsplit != 1) return GI_EINVAL;
    if (p.a_major && !p.b_major) return GI_EINVAL;
    if (!((p.tm == 1 && p.tn == 1) || (p.tm == 1 && p.tn == 2) || (p.tm == 2 && p.tn == 2)))
        return GI_EINVAL;
    // rows narrower than one 16-byte vector must be stored padded to 4 floats
    if (!p.a_major && p.K < 4 && p.lda < 4) return GI_EINVAL;
    if (!p.b_major && p.K < 4 && p.ldb < 4) return GI_EINVAL;
    if (p.a_major && p.M < 4 && p.lda < 4) return GI_EINVAL;
    if (p.b_major && p.N < 4 && p.ldb < 4) return GI_EINVAL;
    if (splitk && p.ngroups)
        for (int g = 0; g < p.ngroups; ++g)
            if (p.gsplit[g] < 1) return GI_EINVAL;
    if (p.flags & GI_GEMM_REDUCE) {
        if (!splitk || !p.a_major || !p.b_major || !p.red_count || p.red_ldw < 1) return GI_EINVAL;
        if (p.ngroups) {
            for (int g = 0; g < p.ngroups; ++g)
                if (!p.Bg[g]) return GI_EINVAL;
        } else if (!p.red_dW) {
            return GI_EINVAL;
        }
    }
    return 0;
}

// grid of one problem; x*y*z == 0 means "nothing to launch"
static dim3 problem_grid(const gi_gemm_params& p) {
    const bool splitk = (p.flags & GI_GEMM_SPLITK) != 0;
    const int BM = 64 * p.tm, BN = 64 * p.tn;
    const int rows = splitk ? p.M : (p.ngroups ? p.max_group_rows : p.M);
    if (rows <= 0) return dim3(0, 0, 0);
    const int groups = p.ngroups ? p.ngroups : 1;
    int zsplit = p.nsplit;
    if (splitk && p.ngroups) {
        zsplit = 0;
        for (int g = 0; g < p.ngroups; ++g) zsplit += p.gsplit[g];
    }
    return dim3(gi_cdiv(p.N, BN), gi_cdiv(rows, BM), splitk ? zsplit : groups);
}

#define GI_DISPATCH(KERNEL, TMV, TNV, GRID, ARG)                                                   \
    do {                                                                                           \
        if (!am && !bm) hipLaunchKernelGGL((KERNEL<TMV, TNV, false, false>), GRID, dim3(256), 0, st, ARG); \
        else if (!am && bm) hipLaunchKernelGGL((KERNEL<TMV, TNV, false, true>), GRID, dim3(256), 0, st, ARG); \
        else hipLaunchKernelGGL((KERNEL<TMV, TNV, true, true>), GRID, dim3(256), 0, st, ARG);      \
    } while (0)

extern "C" int gi_gemm(const gi_gemm_params* pp, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (!pp) return GI_EINVAL;
    gi_gemm_params p = *pp;
    const int rc = validate(p);
    if (rc) return rc;
    const dim3 grid = problem_grid(p);
    if (grid.x == 0) return 0;
    if (want_remap((long long)grid.x * grid.y * grid.z)) p.flags |= 32;
    if (grid.y > 65535u || grid.z > 65535u) return GI_ELIMIT;
    hipStream_t st = (hipStream_t)stream;
    // useful flops of this launch (real dims; for grouped / split launches M resp. K is the total)
    GiProfScope prof(st, GI_PROF_GEMM, 2.0 * (double)p.M * (double)p.N * (double)p.K);
    log_launch(&p, 1, grid.x * grid.y * grid.z, 2.0 * (double)p.M * (double)p.N * (double)p.K);
    const bool am = p.a_major, bm = p.b_major;
    if (p.tm == 1 && p.tn == 1) GI_DISPATCH(gi_gemm_kernel, 1, 1, grid, p);
    else if (p.tm == 1 && p.tn == 2) GI_DISPATCH(gi_gemm_kernel, 1, 2, grid, p);
    else GI_DISPATCH(gi_gemm_kernel, 2, 2, grid, p);
    return gi_launch_status();
}

// k tiles one workgroup of the problem walks through (its run time, to first order)
static int wg_k_tiles(const gi_gemm_params& p) {
    if (!(p.flags & GI_GEMM_SPLITK)) return gi_cdiv(p.K, 32);
    const int len = p.ngroups ? p.max_group_rows : p.K;
    const int nsp = p.ngroups ? p.gsplit[0] : p.nsplit;
    return gi_cdiv(gi_cdiv(len, nsp > 0 ? nsp : 1), 32);
}

extern "C" int gi_gemm_batch(const gi_gemm_params* probs, int n, void* stream) {
    (void)hipGetLastError();
    if (!probs || n < 1 || n > GI_GEMM_BATCH_MAX) return GI_EINVAL;
    if (n == 1) return gi_gemm(probs, stream);
    GemmBatch b;
    memset(&b, 0, sizeof(b));
    double flops = 0;
    int total = 0, k = 0;
    // longest reductions first: workgroup ids are dispatched in order, so the short workgroups are
    // the ones that fill the last, partly empty round of the launch
    int order[GI_GEMM_BATCH_MAX];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i)                       // stable insertion sort by reduction length
        for (int j = i; j > 0 && wg_k_tiles(probs[order[j]]) > wg_k_tiles(probs[order[j - 1]]); --j) {
            const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
    for (int ii = 0; ii < n; ++ii) {
        const gi_gemm_params& p = probs[order[ii]];
        const int rc = validate(p);
        if (rc) return rc;
        if (p.tm != probs[0].tm || p.tn != probs[0].tn || p.a_major != probs[0].a_major ||
            p.b_major != probs[0].b_major)
            return GI_EINVAL;
        const dim3 g = problem_grid(p);
        if (g.x == 0) continue;
        b.p[k] = p; b.gx[k] = g.x; b.gy[k] = g.y; b.start[k] = total;
        total += g.x * g.y * g.z;
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
        ++k;
    }
    if (k == 0) return 0;
    b.start[k] = total;
    b.n = k;
    if (want_remap(total))
        for (int i = 0; i < k; ++i) b.p[i].flags |= 32;
    hipStream_t st = (hipStream_t)stream;
    GiProfScope prof(st, GI_PROF_GEMM, flops);
    log_launch(b.p, k, total, flops);
    const bool am = probs[0].a_major, bm = probs[0].b_major;
    const dim3 grid(total);
    if (probs[0].tm == 1 && probs[0].tn == 1) GI_DISPATCH(gi_gemm_batch_kernel, 1, 1, grid, b);
    else if (probs[0].tm == 1 && probs[0].tn == 2) GI_DISPATCH(gi_gemm_batch_kernel, 1, 2, grid, b);
    else GI_DISPATCH(gi_gemm_batch_kernel, 2, 2, grid, b);
    return gi_launch_status();
}
