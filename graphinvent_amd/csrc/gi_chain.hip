// Resident-activation MLP chain (gfx950): a whole per-bond-type message MLP — every Linear + SELU of
// `MLP.forward` (gnn/modules.py:166-170) as called from `GGNN.message_terms` (gnn/mpnn.py:284-294) and
// `AttentionGGNN.aggregate_message` (gnn/mpnn.py:370-389) — or its whole dZ chain, in ONE launch.
//
// Why: layer by layer these are K <= 250 GEMMs over ~8 k rows, five dependent launches per message
// pass (+ five for the dZ chain) that each leave the MFMA pipes 86 % idle (launch ramp, prologue,
// epilogue; profiles/r01).  A row block's chain depends on nothing but its own rows, so one workgroup
// can carry 32 rows through all layers:
//   * the 32 x (<= 256) activation tile lives in LDS (33 KB) and is rewritten in place by each
//     layer's epilogue — it is the MFMA A operand of the next layer, never re-read from HBM;
//   * the layer's whole output row (<= 256 columns) is held in accumulators: 8 waves x one 32x32
//     v_mfma_f32_32x32x2_f32 tile (exact fp32, the parity bar);
//   * weights stream from L2 through a double-buffered LDS k-tile (32 deep) with two register stages,
//     and the stream runs AHEAD across layer boundaries (the weights do not depend on activations),
//     so a layer change costs one barrier pair, not a launch + pipeline refill;
//   * every layer's output is also written to HBM once (activations for the backward / dZ for the
//     deferred weight-gradient GEMMs).
// 512 threads = 8 waves = 2 per SIMD: while one wave of a SIMD waits for its LDS fragments the other
// issues MFMAs.  LDS 107 KB -> one workgroup per CU; U/32 row blocks ~ one per CU at B = 1000.
//
//   forward  (BWD = false): Y_l = selu(A W_l^T + b_l),  W_l stored [N][K]   (reduction contiguous)
//   backward (BWD = true) : dZ_{l-1} = (dZ_l W_l) * selu'(act_{l-1}),  W_l stored [K][N]
#include <stdlib.h>
#include <string.h>

#include "gi_mfma.h"

namespace {

constexpr int CH_ROWS = 32;                 // rows per workgroup
constexpr int CH_W = GI_CHAIN_MAXW;         // widest layer (8 waves x 32 columns)
constexpr int CH_ALD = CH_W + 4;            // activation tile row stride (conflict-free ds_read_b128)
constexpr int CH_BLD_M = CH_W + 4;          // weight tile row stride, reduction-major storage
// KT = reduction depth of one staged weight tile.  32: 107 KB of LDS, one workgroup per CU, 16 MFMAs
// per wave between barriers.  16: 74 KB, TWO workgroups per CU — when the row blocks do not divide
// evenly over the 256 CUs (U / 32 = 264 blocks at the headline batch) the overflow blocks run beside
// the others instead of as a second round on an otherwise idle chip.
template <int KT> struct ChainGeom {
    static constexpr int BLD_C = KT + 4;                    // weight tile row stride, reduction-contiguous
    static constexpr int BSZ = (CH_W * BLD_C > KT * CH_BLD_M) ? CH_W * BLD_C : KT * CH_BLD_M;
    static constexpr int NS = KT / 8;                       // float4 staged per thread per tile
    static constexpr int NG = KT / 8;                       // 8-deep MFMA groups per tile
};

struct ChainArgs {
    gi_chain_params c[2];
    int nchains;
    int tile_off[2][GI_MAX_GROUPS + 1];     // prefix of 32-row tiles over the groups of a chain
    int chain_off[3];                       // prefix of tiles over the chains
    long long* trace;                       // measurement aid (GI_CHAIN_TRACE): 16 words per workgroup
};

template <bool BWD, int CH_KT>
__global__ __launch_bounds__(512) void gi_chain_kernel(const ChainArgs args) {
    using Geo = ChainGeom<CH_KT>;
    constexpr int CH_BLD_C = Geo::BLD_C, CH_BSZ = Geo::BSZ, NS = Geo::NS, NG = Geo::NG;
    constexpr int CPR = CH_KT / 4;                          // float4 per weight row of a [N][K] tile
    constexpr int RPP = 512 / CPR;                          // weight rows staged per pass ([N][K] tiles)
    __shared__ __attribute__((aligned(16))) float As[CH_ROWS * CH_ALD];
    __shared__ __attribute__((aligned(16))) float Bs[2 * CH_BSZ];

    // ---- which (chain, group, row block) -------------------------------------------------------
    const int id = blockIdx.x;
    const int ci = (args.nchains > 1 && id >= args.chain_off[1]) ? 1 : 0;
    const gi_chain_params& P = args.c[ci];
    const int local = id - args.chain_off[ci];
    int g = 0;
    while (g < P.ngroups - 1 && local >= args.tile_off[ci][g + 1]) ++g;
    // (device loads land in VGPRs; readfirstlane tells the compiler the row range is wave-uniform,
    // so everything derived from it — buffer descriptors included — stays scalar)
    const int lo = P.grp_off ? __builtin_amdgcn_readfirstlane(P.grp_off[g]) : 0;
    const int hi = P.grp_off ? __builtin_amdgcn_readfirstlane(P.grp_off[g + 1]) : P.rows;
    const int r0 = lo + CH_ROWS * (local - args.tile_off[ci][g]);
    if (r0 >= hi) return;                                   // block-uniform, before any barrier
    const int L = P.nlayers;
    const long long t_start = args.trace ? (long long)wall_clock64() : 0;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // weight staging coordinates: [N][K] storage: CPR float4 per KT-deep row; [K][N]: 64 float4 per row
    const int cc4 = tid % CPR, crow = tid / CPR;            // rows crow + RPP i
    const int mc4 = tid & 63, mrow = tid >> 6;              // reduction rows mrow + 8 i

    // Iterator over the (layer, k tile) steps of the chain; the layer's scalars are fetched once per
    // layer change, so the step body itself is free of scalar loads and branches.  Past the last
    // tile an iterator stays on it: the weight stream then re-stages that tile, which nobody reads.
    struct It { int l, kt, nk, K, N; const float* W; };
    auto load_layer = [&](It& it) {
        const gi_chain_layer& Ly = P.layer[it.l];
        it.K = Ly.K; it.N = Ly.N; it.W = Ly.W[g];
        it.nk = (Ly.K + CH_KT - 1) / CH_KT;
    };
    auto advance = [&](It& it) {
        if (++it.kt == it.nk) {
            if (it.l + 1 < L) { ++it.l; it.kt = 0; load_layer(it); }
            else it.kt = it.nk - 1;
        }
    };

    // global -> registers (raw, clamped addresses; nothing consumes the data here); `part` of
    // `nparts` equal shares of the thread's NS vectors, so the loads can be spread over MFMA groups
    auto gload = [&](v4f (&rw)[NS], const It& it, int part, int nparts) {
        const float* __restrict__ W = it.W;
        const int K = it.K, N = it.N, k0 = it.kt * CH_KT;
        const int per = NS / nparts;
#pragma unroll
        for (int i = part * per; i < (part + 1) * per; ++i) {
            if (!BWD) {
                const int n = min(crow + RPP * i, N - 1);
                rw[i] = gi_load4_raw(W + (long long)n * K, k0 + 4 * cc4, K - 4);
            } else {
                const int kr = min(k0 + mrow + 8 * i, K - 1);
                rw[i] = gi_load4_raw(W + (long long)kr * N, 4 * mc4, N - 4);
            }
        }
    };
    // registers -> LDS; zero fill along the REDUCTION dimension only (garbage along the output
    // dimension feeds output elements the epilogue discards).  Always through the branch-free fix-up
    // (a few v_cndmask in the shadow of the MFMAs) so the whole step stays one basic block and the
    // compiler can count vmcnt instead of draining it.
    auto sstore = [&](v4f (&rw)[NS], int buf, const It& it, int part, int nparts) {
        const int K = it.K, N = it.N, k0 = it.kt * CH_KT;
        float* b = Bs + buf * CH_BSZ;
        const int per = NS / nparts;
#pragma unroll
        for (int i = part * per; i < (part + 1) * per; ++i) {
            if (!BWD) {
                *(v4f*)&b[(crow + RPP * i) * CH_BLD_C + 4 * cc4] =
                    gi_fix4(rw[i], k0 + 4 * cc4, K - 4, K, true);
            } else {
                const bool ok = k0 + mrow + 8 * i < K;
                *(v4f*)&b[(mrow + 8 * i) * CH_BLD_M + 4 * mc4] = gi_fix4(rw[i], 4 * mc4, N - 4, N, ok);
            }
        }
    };

    f32x16 acc;
    auto read_frags = [&](int buf, int kt, int k8, float (&af)[4], float (&bf)[4]) {
        const v4f a = *(const v4f*)&As[l31 * CH_ALD + kt * CH_KT + k8 * 8 + 4 * lhi];
        af[0] = a.x; af[1] = a.y; af[2] = a.z; af[3] = a.w;
        const float* b = Bs + buf * CH_BSZ;
        if (!BWD) {
            const v4f v = *(const v4f*)&b[(wid * 32 + l31) * CH_BLD_C + k8 * 8 + 4 * lhi];
            bf[0] = v.x; bf[1] = v.y; bf[2] = v.z; bf[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = b[(k8 * 8 + j + 4 * lhi) * CH_BLD_M + wid * 32 + l31];
        }
    };
    auto mma = [&](const float (&af)[4], const float (&bf)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], acc, 0, 0, 0);
    };

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    // HBM side through buffer resources sized to the tile's valid rows: the hardware drops
    // out-of-range stores and returns 0 for out-of-range loads, so there is neither a predicated
    // store (a basic-block boundary that drains vmcnt) nor a clamped address per element.
    auto epilogue = [&](int l) {
        const gi_chain_layer& Ly = P.layer[l];
        const int N = Ly.N, ldo = Ly.ldo;
        const int col = wid * 32 + l31;
        const bool col_ok = col < N;
        const int nrows = min(hi - r0, CH_ROWS);
        const int coff = col_ok ? 4 * col : 0x40000000;     // beyond any tile: dropped / reads 0
        float av[16];
        const bool dselu = BWD && Ly.act != nullptr;
        if (dselu) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(Ly.act + (long long)r0 * Ly.ldact), 0, nrows * Ly.ldact * 4, 0x00020000);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                av[r] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(ra, row * Ly.ldact * 4 + coff, 0, 0));
            }
        }
        const float bv = BWD ? 0.f : Ly.bias[g][col_ok ? col : N - 1];
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = acc[r] + bv;
            if (!BWD) x = gi_selu(x);
            if (dselu) x *= gi_selu_grad(av[r]);
            v[r] = col_ok ? x : 0.f;                         // zero = the next layer's k padding
        }
        if (l + 1 < L) {                                     // next layer's A operand, in place
#pragma unroll
            for (int r = 0; r < 16; ++r) As[((r & 3) + 8 * (r >> 2) + 4 * lhi) * CH_ALD + col] = v[r];
        }
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Ly.out + (long long)r0 * ldo), 0, nrows * ldo * 4, 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), ro,
                                                  row * ldo * 4 + coff, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (l + 1 < L) __syncthreads();
    };

    // ---- prologue: the 32 input rows -> LDS (zero beyond K0), first two weight tiles ------------
    {
        const int K0 = P.layer[0].K, cmax = ((K0 + 3) & ~3) - 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = mrow + 8 * i;
            const int grow = min(r0 + row, hi - 1);
            const long long src = P.x_idx ? P.x_idx[grow] : grow;
            v4f v = gi_load4_raw(P.X + src * P.ldx, 4 * mc4, cmax);
            const int c = 4 * mc4;
            v.x = (c < K0) ? v.x : 0.f; v.y = (c + 1 < K0) ? v.y : 0.f;
            v.z = (c + 2 < K0) ? v.z : 0.f; v.w = (c + 3 < K0) ? v.w : 0.f;
            *(v4f*)&As[row * CH_ALD + c] = v;
        }
    }
    v4f rw0[NS], rw1[NS];
    int T = 0;                                               // steps of the whole chain
    for (int l = 0; l < L; ++l) T += (P.layer[l].K + CH_KT - 1) / CH_KT;
    It ic, is, il;
    ic.l = 0; ic.kt = 0; load_layer(ic);
    is = ic;
    gload(rw0, is, 0, 1);
    advance(is);
    gload(rw1, is, 0, 1);
    il = is; advance(il);
    sstore(rw0, 0, ic, 0, 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __syncthreads();
    long long t_phase[GI_CHAIN_MAXL + 1];
    if (args.trace) t_phase[0] = (long long)wall_clock64();

    // ---- main loop over (layer, k tile) steps -----------------------------------------------------
    // Step s computes tile s from LDS buffer s & 1, writes tile s+1 (loaded during step s-1) to the
    // other buffer and fetches tile s+2 into the register stage that has just been drained; every
    // memory instruction is pinned between MFMA groups (sched_barrier), see gi_gemm.hip: the loads go
    // with the first half of the step's 8-deep MFMA groups, the LDS writes with the second half.
    const int swid = __builtin_amdgcn_readfirstlane(wid);   // wave-uniform: scalar branch below
    float af[2][4], bf[2][4];
    constexpr int HALF = NG / 2;
#define GI_CHAIN_STEP(BUF, RS, RL)                                                                 \
    {                                                                                              \
        const int kt = ic.kt;                                                                      \
        if (swid * 32 < ic.N) {                    /* this wave owns output columns of the layer */ \
            read_frags(BUF, kt, 0, af[0], bf[0]);                                                  \
            _Pragma("unroll") for (int q = 0; q < NG; ++q) {                                       \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                mma(af[q & 1], bf[q & 1]);                                                         \
                if (q + 1 < NG) read_frags(BUF, kt, q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);      \
                if (q < HALF) gload(RL, il, q, HALF);                                              \
                else sstore(RS, (BUF) ^ 1, is, q - HALF, HALF);                                    \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        } else {                                   /* narrow layer: only stage the weight stream */ \
            gload(RL, il, 0, 1);                                                                   \
            sstore(RS, (BUF) ^ 1, is, 0, 1);                                                       \
        }                                                                                          \
        __syncthreads();                                                                           \
        if (kt == ic.nk - 1) {                                                                     \
            epilogue(ic.l);                                                                        \
            if (args.trace) t_phase[ic.l + 1] = (long long)wall_clock64();                         \
        }                                                                                          \
        advance(ic); advance(is); advance(il);                                                     \
    }
    for (int s = 0; s < T; s += 2) {
        GI_CHAIN_STEP(0, rw1, rw0)
        if (s + 1 >= T) break;
        GI_CHAIN_STEP(1, rw0, rw1)
    }
#undef GI_CHAIN_STEP
    if (args.trace && threadIdx.x == 0) {   // 100 MHz wall clock: start, end, placement, rows, phases
        long long* t = args.trace + 16 * (long long)blockIdx.x;
        t[0] = t_start; t[1] = (long long)wall_clock64();
        t[2] = ((long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) << 8) |
               (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xff);   // HW_ID, XCC_ID
        t[3] = hi - r0;
        for (int l = 0; l <= L && l <= GI_CHAIN_MAXL; ++l) t[4 + l] = t_phase[l];
    }
}

int validate_chain(const gi_chain_params& p) {
    if (p.nlayers < 1 || p.nlayers > GI_CHAIN_MAXL || !p.X || p.rows < 0) return GI_EINVAL;
    if (p.ngroups < 1 || p.ngroups > GI_MAX_GROUPS) return GI_EINVAL;
    if (p.ngroups > 1 && !p.grp_off) return GI_EINVAL;
    if (p.ldx < ((p.layer[0].K + 3) & ~3)) return GI_EINVAL;   // 16-byte reads end inside the row
    for (int l = 0; l < p.nlayers; ++l) {
        const gi_chain_layer& q = p.layer[l];
        if (q.K < 4 || q.N < 4 || q.K > GI_CHAIN_MAXW || q.N > GI_CHAIN_MAXW) return GI_ELIMIT;
        if (l > 0 && q.K != p.layer[l - 1].N) return GI_EINVAL;
        if (!q.out || q.ldo < q.N) return GI_EINVAL;
        if (q.act && q.ldact < q.N) return GI_EINVAL;
        for (int t = 0; t < p.ngroups; ++t) {
            if (!q.W[t]) return GI_EINVAL;
            if (!p.backward && !q.bias[t]) return GI_EINVAL;
        }
    }
    return 0;
}

}  // namespace

extern "C" int gi_mlp_chain(const gi_chain_params* chains, int nchains, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (!chains || nchains < 1 || nchains > 2) return GI_EINVAL;
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    double flops = 0;
    int total = 0;
    for (int c = 0; c < nchains; ++c) {
        const gi_chain_params& p = chains[c];
        const int rc = validate_chain(p);
        if (rc) return rc;
        if (p.backward != chains[0].backward) return GI_EINVAL;
        a.c[c] = p;
        a.chain_off[c] = total;
        int t = 0;
        for (int g = 0; g < p.ngroups; ++g) {
            a.tile_off[c][g] = t;
            const int rows = p.ngroups > 1 ? p.group_rows[g] : p.rows;
            if (rows < 0) return GI_EINVAL;
            t += gi_cdiv(rows, CH_ROWS);
        }
        a.tile_off[c][p.ngroups] = t;
        total += t;
        for (int l = 0; l < p.nlayers; ++l)
            flops += 2.0 * (double)p.rows * (double)p.layer[l].K * (double)p.layer[l].N;
    }
    a.chain_off[nchains] = total;
    if (nchains == 1) a.chain_off[2] = total;
    a.nchains = nchains;
    if (total == 0) return 0;
    // GI_CHAIN_TRACE=<address of a device buffer of 4 * total int64>: per-workgroup timestamps
    a.trace = getenv("GI_CHAIN_TRACE") ? (long long*)strtoull(getenv("GI_CHAIN_TRACE"), nullptr, 0) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    GiProfScope prof(st, GI_PROF_GEMM, flops);
    // 16-deep tiles (two workgroups per CU) unless told otherwise: measured tools/bench_chain.py
    static const int kt = getenv("GI_CHAIN_KT") ? atoi(getenv("GI_CHAIN_KT")) : 16;
    const dim3 grid(total), block(512);
    if (chains[0].backward) {
        if (kt == 32) hipLaunchKernelGGL((gi_chain_kernel<true, 32>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gi_chain_kernel<true, 16>), grid, block, 0, st, a);
    } else {
        if (kt == 32) hipLaunchKernelGGL((gi_chain_kernel<false, 32>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gi_chain_kernel<false, 16>), grid, block, 0, st, a);
    }
    return gi_launch_status();
}
