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
//
// ONE kernel runs every launch: a workgroup walks the output tiles id = blockIdx.x, + gridDim.x, ...
// of a (batched) launch.  With gridDim.x = number of tiles that is the ordinary one-tile-per-
// workgroup launch; big launches get a PERSISTENT grid instead (as many workgroups as the device
// holds at once), and a workgroup issues the first global loads of its NEXT tile in front of the
// epilogue of the current one.  Measured on the hot launches (tools/gemm_lab.hip per-workgroup trace,
// round 3): a workgroup spent 12 % of its life in the prologue waiting for those loads and 17 % in
// the epilogue, holding one of its CU's four slots while feeding nothing to the MFMA pipe.
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "gi_common.h"

#include "gi_mfma.h"
#include "gi_x2.h"
#include <type_traits>


__device__ float gi_store_sink[256];               // where out-of-range lanes of edge tiles store

// Operand addressing: block-uniform base pointer + 32-bit BYTE offset per lane (`global_load ... v_off,
// s[base]`): one address VGPR and 32-bit integer arithmetic per access instead of a 64-bit pointer per
// lane.  Every matrix handed to gi_gemm must therefore span less than 4 GB (checked by validate()).
__device__ __forceinline__ v4f gi_load4_at(const float* base, unsigned row_bytes, int col, int cmax) {
    const unsigned off = row_bytes + 4u * (unsigned)max(min(col, cmax), 0);
    return *(const v4f_u*)((const char*)base + off);
}

// tools/gemm_lab.hip (-DGI_GEMM_TRACE): per-tile shader-clock stamps + hardware placement
#ifdef GI_GEMM_TRACE
__device__ unsigned long long* gi_trace_buf;       // [tiles][8]
#define GI_TRACE(id, slot)                                                                        \
    do { if (gi_trace_buf && threadIdx.x == 0)                                                     \
        gi_trace_buf[(size_t)(id) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define GI_TRACE_HW(id)                                                                            \
    do { if (gi_trace_buf && threadIdx.x == 0) {                                                   \
        gi_trace_buf[(size_t)(id) * 8 + 4] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   \
        gi_trace_buf[(size_t)(id) * 8 + 5] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   \
        gi_trace_buf[(size_t)(id) * 8 + 6] = blockIdx.x; } } while (0)
#else
#define GI_TRACE(id, slot) do {} while (0)
#define GI_TRACE_HW(id) do {} while (0)
#endif

// Several independent GEMMs of the same tile/layout class in ONE launch ("horizontal fusion"):
// the sibling MLPs of the readout (4 node-level stacks, 3 graph-level stacks) and the two GRU
// projections have 16..450 workgroups each — together they fill the 256 CUs and halve the number of
// launch ramps on the critical path.  Tile id -> (problem, x, y, z) through a prefix table.
#define GI_GEMM_BATCH_MAX 8
struct GemmBatch {
    gi_gemm_params p[GI_GEMM_BATCH_MAX];
    int start[GI_GEMM_BATCH_MAX + 1];          // first tile id of every problem
    int gx[GI_GEMM_BATCH_MAX], gy[GI_GEMM_BATCH_MAX];
    int n, total;
};

// Block-uniform description of one output tile (SGPRs) + the per-thread stored rows of a contig A
// (the only per-thread state a tile carries: everything else is recomputed from the scalars).
template <int NA>
struct GemmTile {
    int pi;                                // problem of the batch (kernel argument, indexed: never its address)
    const float* Ap; const float* Bp; const float* biasp; float* Cp;
    int id, m_end, k_begin, k_end, m0, n0;
    int lda, ldb, a_cmax, b_cmax;
    bool a_fast, b_fast, valid, empty;
    int a_row[NA];                         // contig A: stored row of staging slot i (gathered / clamped)
};
// What the epilogue of a finished tile needs while the NEXT tile is being computed.
struct GemmEpi {
    int pi; const float* biasp; float* Cp;
    int m_end, m0, n0;
    bool valid;
};

// EPI: 0 = epilogue from the run-time flags; 1 = the epilogue of the layout's own launch class, known at
// compile time (forward: bias + SELU; dgrad: * selu'(act); weight-gradient slabs: plain store) — the
// launcher picks it when every problem of the launch has exactly those flags.
template <int TM, int TN, bool A_MAJOR, bool B_MAJOR, int EPI, bool PERSIST>
__global__ __launch_bounds__(256) void gi_gemm_tiles_kernel(const GemmBatch b) {
    constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
    constexpr int A_LD = A_MAJOR ? BM + 4 : BK + 4;
    constexpr int A_ROWS = A_MAJOR ? BK : BM;
    constexpr int B_LD = B_MAJOR ? BN + 4 : BK + 4;
    constexpr int B_ROWS = B_MAJOR ? BK : BN;
    constexpr int A_SZ = A_ROWS * A_LD, B_SZ = B_ROWS * B_LD;
    constexpr int NA = 2 * TM, NB = 2 * TN;          // float4 staged per thread per tile
    typedef GemmTile<NA> Tile;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];
    float* const As = smem;
    float* const Bs = smem + 2 * A_SZ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- per-thread staging coordinates -----------------------------------------------------
    // contig operand: 8 float4 per 32-wide row -> c4 = tid&7, row = (tid>>3) + 32*i
    // major operand : BX/4 float4 per stored row -> c4 = tid % (BX/4), red = tid/(BX/4) + i*RP
    const int cc4 = tid & 7, crow = tid >> 3;
    constexpr int A_C4 = BM / 4, A_RP = 256 / A_C4;
    constexpr int B_C4 = BN / 4, B_RP = 256 / B_C4;
    const int a_mc4 = tid % A_C4, a_mr = tid / A_C4;
    const int b_mc4 = tid % B_C4, b_mr = tid / B_C4;

    // ---- tile id -> problem, tile coordinates, group / split resolution (block-uniform) -------
    auto setup = [&](Tile& t, int id) __attribute__((always_inline)) {
        t.id = id;
        t.valid = id < b.total;
        if (!t.valid) return;
        int i = 0;
        while (i < b.n - 1 && id >= b.start[i + 1]) ++i;
        const gi_gemm_params& p = b.p[i];
        t.pi = i;
        const int gx = b.gx[i], gy = b.gy[i], gxy = gx * gy;
        int local = id - b.start[i];
        // split-K launches (weight gradients), flag bit 11: the XCD-aware order over the problem's WHOLE tile list, so
        // that one XCD walks consecutive slabs — the tiles of a slab read the SAME rows of both operands and share
        // them in one L2 (round 5; the order inside a slab, below, spreads a slab over all eight L2s)
        const bool slab_order = (p.flags & 2048) != 0;
        if (slab_order) {
            const int T = b.start[i + 1] - b.start[i];
            const int q = T >> 3, r = T & 7, xcd = local & 7, j = local >> 3;
            local = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        }
        const int bz = local / gxy;
        int rem = local - bz * gxy;
        // XCD-aware tile order (flag bit 5, set by the host for launches of many workgroups, see
        // remap_min_blocks): the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs
        // (private L2 each) and a persistent workgroup's stride is a multiple of 8; the remap makes one
        // XCD walk consecutive tiles so the column tiles sharing an A row panel hit the same L2
        // (bijective for any grid size).
        // (bounded launch: gy covers the row BOUND; the order is taken over the tiles that have rows, or
        // the XCDs that own the surplus tiles would idle — measured: 5 of 8 XCDs did all the work)
        int gxy_real = gxy;
        if (p.m_dev) gxy_real = gx * ((min(p.M, *p.m_dev) + BM - 1) / BM);
        const bool surplus = rem >= gxy_real;
        if ((p.flags & 32) && !surplus && !slab_order) {
            const int q = gxy_real >> 3, r = gxy_real & 7, xcd = rem & 7, j = rem >> 3;
            rem = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        }
        const int by = rem / gx, bx = rem - by * gx;
        const bool splitk = (p.flags & GI_GEMM_SPLITK) != 0;
        int g = bz, s = 0, nsp = p.nsplit;
        if (splitk) {
            g = 0; s = bz;
            if (p.ngroups) {                                  // per-group slab counts (work-proportional)
                while (g < p.ngroups - 1 && s >= p.gsplit[g]) { s -= p.gsplit[g]; ++g; }
                nsp = p.gsplit[g];
            }
        }
        t.Ap = p.A;
        t.Bp = (p.ngroups && !splitk) ? p.Bg[g] : p.B;
        t.biasp = (p.ngroups && !splitk) ? p.biasg[g] : p.bias;
        t.Cp = (p.ngroups && splitk) ? p.Cg[g] : p.C;
        int m_begin = 0;
        t.m_end = p.M; t.k_begin = 0; t.k_end = p.K;
        if (p.m_dev) t.m_end = min(p.M, *p.m_dev);          // bounded launch: real extents on the device
        if (p.k_dev) t.k_end = min(p.K, *p.k_dev);
        if (p.grp_off) {
            const int lo = p.grp_off[g], hi = p.grp_off[g + 1];
            if (splitk) { t.k_begin = lo; t.k_end = hi; } else { m_begin = lo; t.m_end = hi; }
        }
        if (splitk) {
            const int len = t.k_end - t.k_begin;
            const int chunk = (((len + nsp - 1) / nsp) + 31) & ~31;
            const int kb = t.k_begin + s * chunk;
            t.k_end = min(kb + chunk, t.k_end);
            t.k_begin = kb;
            t.Cp += (long long)s * p.c_split_stride;
        }
        t.m0 = m_begin + by * BM;
        t.n0 = bx * BN;
        t.empty = t.m0 >= t.m_end;                          // tile beyond the rows of a short group
        if (t.empty) { t.k_end = t.k_begin; t.m0 = 0; t.m_end = 1; }
        t.lda = p.lda; t.ldb = p.ldb;
        if (!A_MAJOR) {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int row = min(t.m0 + crow + 32 * q, t.m_end - 1);     // rows past the end: any readable row
                t.a_row[q] = p.a_idx ? p.a_idx[row] : row;
            }
        }
        // Only the REDUCTION dimension needs zero fill (garbage there would reach valid outputs).  Along
        // the output dimensions out-of-range rows/columns are merely clamped to readable addresses:
        // whatever they hold only feeds output elements the epilogue discards.  So every tile that is
        // full in k stores the raw vectors ("fast"); the last, partial k tile takes the fix-up path.
        const int bcols = (p.ones_col >= 0) ? p.ones_col : p.N;     // real stored columns of a major B
        const int bc4 = (bcols + 3) & ~3, m4 = (p.M + 3) & ~3;
        t.a_cmax = A_MAJOR ? ((p.lda >= m4) ? m4 - 4 : p.M - 4) : t.k_end - 4;
        t.b_cmax = B_MAJOR ? ((p.ldb >= bc4) ? bc4 - 4 : bcols - 4) : t.k_end - 4;
        t.a_fast = A_MAJOR ? (p.lda >= m4) : true;
        t.b_fast = B_MAJOR ? (p.ldb >= bc4) : true;
    };

    v4f ra0[NA], rb0[NB], ra1[NA], rb1[NB];    // two register stages: k tiles t+1 and t+2 in flight

    // raw loads only (nothing consumes the data here); A half and B half are issued separately so
    // they can be spread between MFMA groups
    auto gload_a = [&](const Tile& t, v4f (&ra)[NA], int k0) {
        if (!A_MAJOR) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                ra[i] = gi_load4_at(t.Ap, (unsigned)t.a_row[i] * (unsigned)t.lda * 4u, k0 + 4 * cc4, t.a_cmax);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int red = min(k0 + a_mr + i * A_RP, t.k_end - 1);
                ra[i] = gi_load4_at(t.Ap, (unsigned)red * (unsigned)t.lda * 4u, t.m0 + 4 * a_mc4, t.a_cmax);
            }
        }
    };
    // BIDX (compile time): rows of a major B gathered through b_idx.  The index load is a dependent
    // load in front of the data load (it drains the load queue once per tile), so the body exists
    // twice and only launches with a gathered problem (first layer of the message stacks' weight
    // gradients) run the BIDX version.
    auto gload_b = [&](const Tile& t, v4f (&rb)[NB], int k0, auto bidx) {
        constexpr bool BIDX = decltype(bidx)::value;
        if (!B_MAJOR) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int row = min(t.n0 + crow + 32 * i, b.p[t.pi].N - 1);
                rb[i] = gi_load4_at(t.Bp, (unsigned)row * (unsigned)t.ldb * 4u, k0 + 4 * cc4, t.b_cmax);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int red = min(k0 + b_mr + i * B_RP, t.k_end - 1);
                int srow = red;
                if (BIDX) { if (b.p[t.pi].b_idx) srow = b.p[t.pi].b_idx[red]; }
                rb[i] = gi_load4_at(t.Bp, (unsigned)srow * (unsigned)t.ldb * 4u, t.n0 + 4 * b_mc4, t.b_cmax);
            }
        }
    };

    // (fix-up of the last / padding k tile) + LDS write, A half and B half
    auto sstore_a = [&](const Tile& t, v4f (&ra)[NA], int buf, int k0, const bool steady) {
        float* a = As + buf * A_SZ;
        const bool full_k = k0 + BK <= t.k_end;        // block-uniform
        if (steady || (full_k && t.a_fast)) {
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
                        gi_fix4(ra[i], k0 + 4 * cc4, t.a_cmax, t.k_end, t.m0 + crow + 32 * i < t.m_end);
                } else {
                    const bool ok = k0 + a_mr + i * A_RP < t.k_end;
                    *(v4f*)&a[(a_mr + i * A_RP) * A_LD + 4 * a_mc4] =
                        gi_fix4(ra[i], t.m0 + 4 * a_mc4, t.a_cmax, b.p[t.pi].M, ok);
                }
            }
        }
    };
    auto sstore_b = [&](const Tile& t, v4f (&rb)[NB], int buf, int k0, const bool steady) {
        float* bb = Bs + buf * B_SZ;
        const bool full_k = k0 + BK <= t.k_end;
        const int ones_col = b.p[t.pi].ones_col;
        if (steady || (full_k && t.b_fast)) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (!B_MAJOR) {
                    *(v4f*)&bb[(crow + 32 * i) * B_LD + 4 * cc4] = rb[i];
                } else {
                    const int col = t.n0 + 4 * b_mc4;
                    v4f v = rb[i];                   // bias-gradient column of wgrad (-1 never matches)
                    v.x = (col == ones_col) ? 1.f : v.x;
                    v.y = (col + 1 == ones_col) ? 1.f : v.y;
                    v.z = (col + 2 == ones_col) ? 1.f : v.z;
                    v.w = (col + 3 == ones_col) ? 1.f : v.w;
                    *(v4f*)&bb[(b_mr + i * B_RP) * B_LD + 4 * b_mc4] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (!B_MAJOR) {
                    *(v4f*)&bb[(crow + 32 * i) * B_LD + 4 * cc4] =
                        gi_fix4(rb[i], k0 + 4 * cc4, t.b_cmax, t.k_end, t.n0 + crow + 32 * i < b.p[t.pi].N);
                } else {
                    const bool ok = k0 + b_mr + i * B_RP < t.k_end;
                    const int col = t.n0 + 4 * b_mc4;
                    const int bcols = ones_col >= 0 ? ones_col : b.p[t.pi].N;
                    v4f v = gi_fix4(rb[i], col, t.b_cmax, bcols, ok);
                    v.x = (ok & (col == ones_col)) ? 1.f : v.x;
                    v.y = (ok & (col + 1 == ones_col)) ? 1.f : v.y;
                    v.z = (ok & (col + 2 == ones_col)) ? 1.f : v.z;
                    v.w = (ok & (col + 3 == ones_col)) ? 1.f : v.w;
                    *(v4f*)&bb[(b_mr + i * B_RP) * B_LD + 4 * b_mc4] = v;
                }
            }
        }
    };

    f32x16 acc[TM][TN];

    // fragments of one 8-deep reduction group
    auto read_frags = [&](int buf, int k8, float (&af)[TM][4], float (&bf)[TN][4]) {
        const float* a = As + buf * A_SZ;
        const float* bb = Bs + buf * B_SZ;
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
                const v4f v = *(const v4f*)&bb[row * B_LD + k8 * 8 + 4 * lhi];
                bf[t][0] = v.x; bf[t][1] = v.y; bf[t][2] = v.z; bf[t][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[t][j] = bb[(k8 * 8 + j + 4 * lhi) * B_LD + row];
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

    // ---- the k-tile stream of a workgroup --------------------------------------------------------
    // Software pipeline, pinned with sched_barrier(0) (hipcc otherwise parks every non-MFMA
    // instruction after the tile's MFMA block, where nothing hides it — a lone wave then reaches only
    // 54 % MFMA duty, measured).  While the MFMAs of k tile t run, the wave also issues
    //   - the LDS fragment reads of the NEXT 8-deep group,
    //   - the global loads of k tile t+2 into the register stage that has just been drained,
    //   - the LDS writes of k tile t+1 (loaded during k tile t-1) into the other LDS buffer,
    // i.e. every memory instruction sits in the shadow of a 64-cycle MFMA.  A tile's k-tile count is
    // rounded up to even (a k tile past k_end stages zeros), the unit is a PAIR of k tiles (LDS buffer
    // 0 then 1, register stage 0 then 1), and the stream does not stop at output-tile boundaries: the
    // last pair of a tile loads the first two k tiles of the workgroup's NEXT tile and stages its first
    // one, so a new tile starts with its operands in LDS (no prologue; measured before: 12 % of a
    // workgroup's life).  Loads and LDS writes are unconditional (clamped addresses; what nobody needs is
    // garbage nobody reads), so a pair is branch free apart from the partial-k-tile fix-up:
    //
    // STEADY pairs (every k tile they stage is full in k) are ONE basic block: hipcc's wait-count pass
    // keeps the next k tile's loads in flight across the LDS write of the current one (s_waitcnt
    // vmcnt(6) / vmcnt(4)).  With run-time branches in the body it merges the states of both arms and
    // emits vmcnt(0) before every LDS write — each k tile then waits out the L2 latency of loads issued
    // two MFMA groups earlier.  The generic pair (fix-up decided at run time) only runs a tile's last
    // one or two pairs.
    //
    constexpr int own_flags = (!A_MAJOR && !B_MAJOR) ? (GI_EPI_BIAS | GI_EPI_SELU) : (!A_MAJOR ? GI_EPI_DSELU : 0);
    float af0[TM][4], bf0[TN][4], af1[TM][4], bf1[TN][4];

    // one pair of k tiles.  TS1 @ ks1: the k tile staged from register stage 1 into LDS buffer 1, TS2 @ ks2: the
    // one staged from stage 0 into buffer 0 afterwards, TL @ kl0 / kl1: the two k tiles loaded (stage 0, then 1).
    // Inside a tile all three are the tile itself; in its last pair TS2 and TL are the workgroup's next tile.
    // `follow` (block-uniform; always true in a STEADY pair): false in the last pair of a workgroup's last
    // tile — nothing follows, so it loads and stages nothing more.
    auto pair = [&](auto steady_c, const bool follow, const Tile& TS1, int ks1, const Tile& TS2, int ks2,
                    const Tile& TL, int kl0, int kl1, auto bidx) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;
        const bool FINAL = !STEADY && !follow;
        read_frags(0, 0, af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        mma(af0, bf0); read_frags(0, 1, af1, bf1); if (!FINAL) gload_a(TL, ra0, kl0);
        __builtin_amdgcn_sched_barrier(0);
        mma(af1, bf1); read_frags(0, 2, af0, bf0); if (!FINAL) gload_b(TL, rb0, kl0, bidx);
        __builtin_amdgcn_sched_barrier(0);
        mma(af0, bf0); read_frags(0, 3, af1, bf1); sstore_a(TS1, ra1, 1, ks1, STEADY);
        __builtin_amdgcn_sched_barrier(0);
        mma(af1, bf1); sstore_b(TS1, rb1, 1, ks1, STEADY);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_frags(1, 0, af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        mma(af0, bf0); read_frags(1, 1, af1, bf1); if (!FINAL) gload_a(TL, ra1, kl1);
        __builtin_amdgcn_sched_barrier(0);
        mma(af1, bf1); read_frags(1, 2, af0, bf0); if (!FINAL) gload_b(TL, rb1, kl1, bidx);
        __builtin_amdgcn_sched_barrier(0);
        mma(af0, bf0); read_frags(1, 3, af1, bf1); if (!FINAL) sstore_a(TS2, ra0, 0, ks2, STEADY);
        __builtin_amdgcn_sched_barrier(0);
        mma(af1, bf1); if (!FINAL) sstore_b(TS2, rb0, 0, ks2, STEADY);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };

    // ---- epilogue in one piece: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Four rows per step (r = 4c .. 4c+3 are rows row0 + 8c + 0..3), the loads of step c+1 in flight under the
    // arithmetic and stores of step c; out-of-range lanes store to a sink instead of being branched around (a
    // predicated store is a basic-block boundary at which hipcc drains vmcnt(0) — stores count in vmcnt on
    // gfx950 — i.e. one serial memory round trip per output element).
    auto epilogue = [&](const GemmEpi& t) __attribute__((always_inline)) {
        const gi_gemm_params& p = b.p[t.pi];
        const int flags = EPI ? own_flags : p.flags;
        const int m_end = t.m_end;
        float* const Cp = t.Cp;
        const bool need_act = (flags & (GI_EPI_DSELU | GI_EPI_MULACT)) != 0;
        const bool need_c = (flags & GI_EPI_ACCUM) != 0;
        float amax = 0.f;                                // max |stored value| (gi_gemm_params.c_amax)
        auto load_rows = [&](int row0, int colc, int c, float (&av)[4], float (&cv)[4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(row0 + 8 * c + r, m_end - 1);
                if (need_act) av[r] = *(const float*)((const char*)p.act + ((unsigned)row * (unsigned)p.ldact + (unsigned)colc) * 4u);
                if (need_c) cv[r] = *(const float*)((const char*)Cp + ((unsigned)row * (unsigned)p.ldc + (unsigned)colc) * 4u);
            }
        };
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int col = t.n0 + wn * 32 * TN + tn * 32 + l31;
                const bool col_ok = col < p.N;
                const int colc = col_ok ? col : p.N - 1;
                const int row0 = t.m0 + wm * 32 * TM + tm * 32 + 4 * lhi;
                const float bv = (flags & GI_EPI_BIAS) ? t.biasp[colc] : 0.f;
                float av[2][4], cv[2][4];
                if (need_act | need_c) load_rows(row0, colc, 0, av[0], cv[0]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < 3 && (need_act | need_c)) load_rows(row0, colc, c + 1, av[(c + 1) & 1], cv[(c + 1) & 1]);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc[tm][tn][4 * c + r] + bv;
                        if (flags & GI_EPI_SELU) x = gi_selu(x);
                        if (flags & GI_EPI_DSELU) x *= gi_selu_grad(av[c & 1][r]);
                        if (flags & GI_EPI_MULACT) x *= av[c & 1][r];
                        if (flags & GI_EPI_ACCUM) x += cv[c & 1][r];
                        v[r] = x;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = row0 + 8 * c + r;
                        const bool ok = col_ok & (row < m_end);
                        float* dst = ok ? (float*)((char*)Cp + ((unsigned)row * (unsigned)p.ldc + (unsigned)col) * 4u)
                                        : gi_store_sink + tid;
                        *dst = v[r];
                        amax = fmaxf(amax, ok ? fabsf(v[r]) : 0.f);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // the fp16x2 launches that read this tensor next scale it by its largest magnitude (gi_x2.h)
        if (p.c_amax) gx_amax_publish(amax, p.c_amax);
    };
    auto epi_of = [&](const Tile& t) __attribute__((always_inline)) {
        GemmEpi e;
        e.pi = t.pi; e.biasp = t.biasp; e.Cp = t.Cp; e.m_end = t.m_end; e.m0 = t.m0; e.n0 = t.n0;
        e.valid = true;
        return e;
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
        // from ONE register the optimiser cannot see through: as a constant the zero tile is hoisted out of
        // the tile loop and lives in 16 VGPRs for the whole kernel
        float zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = zero;
    };
    // the workgroup's next tile WITH a reduction range from `id` on; tiles without one (an empty group, a
    // trailing split-K slice) are finished on the way: their epilogue of zeros (needs acc == 0)
    auto advance = [&](Tile& t, int id, int stride) __attribute__((always_inline)) {
        while (true) {
            setup(t, id);
            if (!t.valid || t.k_end > t.k_begin) break;
            if (!t.empty) epilogue(epi_of(t));
            id += stride;
        }
    };

    // ---- tile loop ------------------------------------------------------------------------------
    auto run = [&](auto bidx) __attribute__((always_inline)) {
        Tile cur, nxt;
        const int stride = gridDim.x;
        zero_acc();
        advance(cur, blockIdx.x, stride);
        if (!cur.valid) return;
        GI_TRACE_HW(cur.id);
        GI_TRACE(cur.id, 0);
        // the only prologue of the workgroup: first two k tiles -> register stages, the first one -> LDS
        gload_a(cur, ra0, cur.k_begin); gload_b(cur, rb0, cur.k_begin, bidx);
        gload_a(cur, ra1, cur.k_begin + BK); gload_b(cur, rb1, cur.k_begin + BK, bidx);
        sstore_a(cur, ra0, 0, cur.k_begin, false); sstore_b(cur, rb0, 0, cur.k_begin, false);
        __syncthreads();
        while (true) {
            GI_TRACE(cur.id, 1);
            if (PERSIST) advance(nxt, cur.id + stride, stride);            // acc == 0 here
            else nxt.valid = false;
            const bool more = PERSIST && nxt.valid;
            const int k_begin = cur.k_begin, k_end = cur.k_end;
            const int npairs = ((k_end - k_begin + BK - 1) / BK + 1) >> 1;
            // pairs q < n_steady stage only k tiles that are full in k (k tiles 2q+1 and 2q+2)
            const int n_full = (k_end - k_begin) / BK;
            const int n_steady = (cur.a_fast && cur.b_fast) ? min(max((n_full - 1) >> 1, 0), npairs - 1) : 0;
            int q = 0;
            for (; q < n_steady; ++q) {
                const int k1 = k_begin + (2 * q + 1) * BK;
                pair(std::true_type{}, true, cur, k1, cur, k1 + BK, cur, k1 + BK, k1 + 2 * BK, bidx);
            }
            for (; q < npairs; ++q) {                          // the last pair hands over to the next tile, if any
                const int k1 = k_begin + (2 * q + 1) * BK;
                const bool last = q == npairs - 1;
                const Tile& nx = (PERSIST && last && more) ? nxt : cur;
                const int k2 = (PERSIST && last && more) ? nxt.k_begin : k1 + BK;
                pair(std::false_type{}, !last || more, cur, k1, nx, k2, nx, k2, k2 + BK, bidx);
            }
            GI_TRACE(cur.id, 2);
            epilogue(epi_of(cur));
            GI_TRACE(cur.id, 3);
            if (!more) break;
            zero_acc();
            cur = nxt;
            GI_TRACE_HW(cur.id);
            GI_TRACE(cur.id, 0);
        }
    };
    bool any_bidx = false;
    if (B_MAJOR) {
        for (int i = 0; i < b.n; ++i) any_bidx |= b.p[i].b_idx != nullptr;
    }
    if (B_MAJOR && any_bidx) run(std::true_type{});
    else run(std::false_type{});

}

// XCD-aware tile order for launches of at least this many workgroups (0 = never): more than ~1.3
// full rounds of 256 CUs x 4 resident workgroups.  Measured on the training step (3 A/B pairs):
// never 2.74 ms, always 2.76-2.77 ms, >= 1300..2100 workgroups 2.70-2.71 ms.  The small launches lose
// with the remap (an XCD's share of a 1-round launch is not balanced), the big ones win (column tiles
// sharing an A row panel hit one L2).
static int remap_min_blocks() { return 1350; }
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

void gi_gemm_log_launch(const char* cls, const gi_gemm_params* probs, int n, int blocks, double flops) {
    FILE* f = launch_log();
    if (!f) return;
    fprintf(f, "%s %d %d %.0f", cls, n, blocks, flops);
    for (int i = 0; i < n; ++i)
        fprintf(f, " %dx%dx%d:g%d:s%d", probs[i].M, probs[i].N, probs[i].K, probs[i].ngroups,
                probs[i].nsplit);
    fputc('\n', f);
    fflush(f);
}
static void log_launch(const gi_gemm_params* probs, int n, int blocks, double flops) {
    const char cls[3] = {(char)('0' + (probs[0].a_major ? 1 : 0)), (char)('0' + (probs[0].b_major ? 1 : 0)), 0};
    gi_gemm_log_launch(cls, probs, n, blocks, flops);
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
    if ((p.m_dev || p.k_dev) && (p.ngroups || (p.flags & GI_GEMM_SPLITK))) return GI_EINVAL;
    // 32-bit byte offsets inside every matrix (gi_load4_at)
    const long long lim = 0xffffffffLL / 4;
    const bool splitk2 = (p.flags & GI_GEMM_SPLITK) != 0;
    const long long a_rows = p.a_major ? p.K : p.M, b_rows = p.b_major ? p.K : p.N;
    if (!p.a_idx && a_rows * (long long)p.lda > lim) return GI_ELIMIT;
    if (!p.b_idx && b_rows * (long long)p.ldb > lim) return GI_ELIMIT;
    if ((long long)p.M * p.ldc > lim || (p.act && (long long)p.M * p.ldact > lim)) return GI_ELIMIT;
    (void)splitk2;
    if (p.flags & ~(GI_EPI_BIAS | GI_EPI_SELU | GI_EPI_DSELU | GI_EPI_ACCUM | GI_GEMM_SPLITK | GI_EPI_MULACT))
        return GI_EINVAL;                               // bit 5 (tile order) is the launcher's
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

// k tiles one workgroup of the problem walks through (its run time, to first order)
static int wg_k_tiles(const gi_gemm_params& p) {
    if (!(p.flags & GI_GEMM_SPLITK)) return gi_cdiv(p.K, 32);
    const int len = p.ngroups ? p.max_group_rows : p.K;
    const int nsp = p.ngroups ? p.gsplit[0] : p.nsplit;
    return gi_cdiv(gi_cdiv(len, nsp > 0 ? nsp : 1), 32);
}

typedef void (*gi_tiles_fn)(const GemmBatch);
// variants: tile (3) x layout (3) x {run-time epilogue | the layout's own epilogue x {one tile per workgroup, tile stream}}
static gi_tiles_fn tiles_kernel(int tm, int tn, bool am, bool bm, int epi, bool persist) {
#define GI_PICK3(TMV, TNV, E, P)                                                                   \
    return (!am && !bm) ? gi_gemm_tiles_kernel<TMV, TNV, false, false, E, P>                      \
         : (!am && bm)  ? gi_gemm_tiles_kernel<TMV, TNV, false, true, E, P>                       \
                        : gi_gemm_tiles_kernel<TMV, TNV, true, true, E, P>
#define GI_PICK2(TMV, TNV, E) do { if (persist) { GI_PICK3(TMV, TNV, E, true); } else { GI_PICK3(TMV, TNV, E, false); } } while (0)
// (run-time epilogue x tile stream: forward layout only — the bounded forward's GRU projections)
#define GI_PICK(TMV, TNV) do { if (epi) GI_PICK2(TMV, TNV, 1);                                                     \
        else if (persist && !am && !bm) return gi_gemm_tiles_kernel<TMV, TNV, false, false, 0, true>;               \
        else { GI_PICK3(TMV, TNV, 0, false); } } while (0)
    if (tm == 1 && tn == 1) GI_PICK(1, 1);
    if (tm == 1 && tn == 2) GI_PICK(1, 2);
    GI_PICK(2, 2);
#undef GI_PICK
#undef GI_PICK2
#undef GI_PICK3
}
// the epilogue flags of a layout's own launch class (kernel template EPI = 1)
static int own_epilogue(bool am, bool bm) {
    return (!am && !bm) ? (GI_EPI_BIAS | GI_EPI_SELU) : (!am ? GI_EPI_DSELU : 0);
}

// Persistent grid: the number of workgroups of this kernel variant the device holds at once
// (occupancy x CUs, from the runtime, cached per variant).  A launch with at least
// persist_tenths / 10 times that many tiles runs as that many workgroups walking the tile list
// (stride = grid); smaller launches keep one workgroup per tile.  0 disables
// (default 0, see DESIGN.md: measured a tie on the device alone and a loss beside the weight-gradient stream).
static int g_persist_tenths = -1, g_grid_cap = 0;
static int persist_tenths() {
    if (g_persist_tenths < 0) g_persist_tenths = 0;
    return g_persist_tenths;
}
// Measurement / test hook: persist_tenths >= 0 sets that threshold; grid_cap > 0 runs
// every launch with more tiles than that as grid_cap workgroups walking the tile list (tests use a tiny
// cap to push small problems through the tile loop), 0 removes the cap.
extern "C" int gi_gemm_config(int persist_tenths_, int grid_cap) {
    if (persist_tenths_ >= 0) g_persist_tenths = persist_tenths_;
    g_grid_cap = grid_cap > 0 ? grid_cap : 0;
    return 0;
}
static int resident_blocks(gi_tiles_fn fn, int tm, int tn, bool am, bool bm, int epi) {
    static int cache[2][2][2][2][2];                    // [tm-1][tn-1][am][bm][epi]
    int& c = cache[tm - 1][tn - 1][am ? 1 : 0][bm ? 1 : 0][epi];
    if (c == 0) {
        int dev = 0, cus = 0, per_cu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, 256, 0) != hipSuccess ||
            per_cu < 1)
            per_cu = 1;
        c = (cus > 0 ? cus : 256) * per_cu;
        (void)hipGetLastError();
    }
    return c;
}

static int launch_tiles(GemmBatch& b, double flops, hipStream_t st) {
    const gi_gemm_params& p0 = b.p[0];
    const bool am = p0.a_major, bm = p0.b_major;
    int epi = 1;                                        // every problem has the layout's own epilogue?
    for (int i = 0; i < b.n; ++i)
        if ((b.p[i].flags & ~GI_GEMM_SPLITK) != own_epilogue(am, bm)) epi = 0;
    if (want_remap(b.total))
        for (int i = 0; i < b.n; ++i) b.p[i].flags |= 32;
    {   // GI_WGRAD_SLAB_ORDER=0: split-K launches keep the per-slab order (measurement aid)
        static const bool slab = !(getenv("GI_WGRAD_SLAB_ORDER") && atoi(getenv("GI_WGRAD_SLAB_ORDER")) == 0);
        if (slab && b.total >= 64)
            for (int i = 0; i < b.n; ++i)
                if ((b.p[i].flags & GI_GEMM_SPLITK) && !b.p[i].m_dev) b.p[i].flags |= 2048;
    }
    int grid = b.total;
    // Bounded launches (rows counted on the device, m_dev): the grid would be sized for the BOUND and its surplus
    // workgroups, though they exit at once, are dispatched at the tail of the launch at the dispatcher's rate
    // (measured: 13 000-row bound on 7 259 real rows = 44 % empty tiles, +46 % launch time).  As a tile stream
    // over the device's resident workgroups the empty tiles cost a few scalar loads each.
    bool bounded = false;
    for (int i = 0; i < b.n; ++i) bounded |= b.p[i].m_dev != nullptr;
    const bool can_stream = epi || (!am && !bm);        // the tile-stream variants that are compiled
    const int pt = (bounded && can_stream) ? 10 : (epi ? persist_tenths() : 0);
    if (pt > 0) {
        const int res = resident_blocks(tiles_kernel(p0.tm, p0.tn, am, bm, epi, true), p0.tm, p0.tn, am, bm, epi);
        if ((long long)b.total * 10 >= (long long)res * pt) grid = res;
    }
    if (can_stream && g_grid_cap > 0 && grid > g_grid_cap) grid = g_grid_cap;
    GiProfScope prof(st, GI_PROF_GEMM, flops);
    log_launch(b.p, b.n, b.total, flops);
    hipLaunchKernelGGL(tiles_kernel(p0.tm, p0.tn, am, bm, epi, grid < b.total), dim3(grid), dim3(256), 0, st, b);
    return gi_launch_status();
}

extern "C" int gi_gemm_batch(const gi_gemm_params* probs, int n, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (!probs || n < 1 || n > GI_GEMM_BATCH_MAX) return GI_EINVAL;
    int nbf3 = 0;
    for (int i = 0; i < n; ++i) nbf3 += (probs[i].flags & GI_GEMM_BF3) != 0;
    if (nbf3) return nbf3 == n ? gi_gemm_bf3_launch(probs, n, stream) : GI_EINVAL;
    GemmBatch b;
    memset(&b, 0, sizeof(b));
    double flops = 0;
    int total = 0, k = 0;
    // longest reductions first: tile ids are handed out in order, so the short tiles are the ones
    // that fill the last, partly empty round of the launch
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
        if ((long long)g.x * g.y * g.z + total > 0x3fffffff) return GI_ELIMIT;
        b.p[k] = p; b.gx[k] = g.x; b.gy[k] = g.y; b.start[k] = total;
        total += g.x * g.y * g.z;
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
        ++k;
    }
    if (k == 0) return 0;
    b.start[k] = total;
    b.n = k; b.total = total;
    return launch_tiles(b, flops, (hipStream_t)stream);
}

extern "C" int gi_gemm(const gi_gemm_params* pp, void* stream) {
    if (!pp) return GI_EINVAL;
    return gi_gemm_batch(pp, 1, stream);
}
