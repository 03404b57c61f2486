// Resident-activation MLP chain (gfx950): a whole per-bond-type message MLP — every Linear + SELU of
// `MLP.forward` (gnn/modules.py:166-170) as called from `GGNN.message_terms` (gnn/mpnn.py:284-294) and
// `AttentionGGNN.aggregate_message` (gnn/mpnn.py:370-389) — or its whole dZ chain, in ONE launch.
//
// Why: layer by layer these are K <= 250 GEMMs over ~8 k rows, five dependent launches per message
// pass (+ five for the dZ chain) that each leave the MFMA pipes 86 % idle (launch ramp, prologue,
// epilogue; profiles/r01).  A row block's chain depends on nothing but its own rows, so one workgroup
// carries 32 rows through all layers:
//   * the 32 x (<= 256) activation tile lives in LDS (33 KB) and is rewritten in place by each
//     layer's epilogue — it is the MFMA A operand of the next layer, never re-read from HBM;
//   * the layer's whole output row (<= 256 columns) is held in accumulators: 8 waves x one 32x32
//     v_mfma_f32_32x32x2_f32 tile (exact fp32, the parity bar); 2 waves per SIMD;
//   * the weights arrive as a pre-packed IMAGE (gi_mlp_chain_pack, a few microseconds per training
//     step): per group one linear stream of 32 KB tiles = the exact bytes of a 32-deep LDS weight tile
//     (zero padded to 256 x 32, forward tiles with the bank swizzle baked in), all layers back to back.
//     The kernel streams it with LDS-DMA (global_load_lds_dwordx4: L2 -> LDS, no VGPRs, no ds_write,
//     fully coalesced 1 KB per wave instruction) into a THREE-deep ring, two tiles ahead of the MFMAs
//     and straight across layer boundaries.  Measured on the first, register-staged version of this
//     kernel (tools/trace_chain.py): with one tile in flight a step cost MFMA time + load time
//     (1.8 us against 0.85 us of MFMA work) — the weight stream is latency-bound per CU (bytes in
//     flight / ~1 us), so the ring depth, not the instruction mix, is what buys the overlap;
//   * row blocks and CUs rarely divide evenly: at the headline batch U/32 = 264 blocks met 256 CUs and
//     a second, nearly empty round doubled the launch time.  A workgroup therefore takes up to
//     32 + 4 rows: rows 32.. ride on the otherwise idle VALU (fp32 fma dot products against the same
//     LDS weight tile, one output column per thread, the reduction split over the two halves of the
//     workgroup and summed in a fixed order) at ~3 % of the MFMA time per extra row; the host picks
//     the smallest block height that saves a round;
//   * every layer's output is also written to HBM once (activations for the backward / dZ for the
//     deferred weight-gradient GEMMs) through a bounds-checked buffer descriptor.
//
//   forward  (BWD = false): Y_l = selu(A W_l^T + b_l),  W_l stored [N][K]
//   backward (BWD = true) : dZ_{l-1} = (dZ_l W_l) * selu'(act_{l-1}),  W_l stored [K][N]
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "gi_mfma.h"
#include "gi_x2.h"

namespace {

constexpr int CH_ROWS = 32;                 // rows per workgroup on the MFMA path
constexpr int CH_XMAX = 4;                  // + up to this many extra rows on the VALU (see below)
constexpr int CH_W = GI_CHAIN_MAXW;         // widest layer (8 waves x 32 columns)
constexpr int CH_KT = 32;                   // reduction depth of one weight tile
constexpr int CH_ALD = CH_W + 4;            // activation tile row stride (conflict-free ds_read_b128)
constexpr int CH_TILE = CH_W * CH_KT;       // floats per weight tile image (32 KB), either layout
constexpr int CH_RING = 3;                  // LDS weight buffers

struct ChainArgs {
    gi_chain_params c[2];
    int nchains;
    int tile_rows;                          // rows per workgroup: 32 .. 32 + CH_XMAX
    int tile_off[2][GI_MAX_GROUPS + 1];     // prefix of 32-row tiles over the groups of a chain
    int chain_off[3];                       // prefix of tiles over the chains
    long long* trace;                       // measurement aid (gi_mlp_chain_config): 16 words per workgroup
    // bounded (host-sync-free) launch: the grid is sized from row BOUNDS, the block -> (group, row block) map
    // is computed here from grp_off, and the row-block height comes from the device (gfix dims[1], 0: tile_rows)
    int dev_tiles;
    const int* tile_rows_dev;
    // XCD-aware row-block order (ordinary launches): the dispatcher deals consecutive workgroup ids round-robin to the 8
    // XCDs (private 4 MB L2 each).  Row blocks are sorted by bond type, and a block streams its TYPE's whole ~1 MB
    // weight image: in dispatch order every XCD's L2 sees all three images (3.5 MB + the activations: 17 % / 32 % of
    // the L2 requests of the fp32 / fp16x2 chain missed, profiles/r04/pmc_chain_kernels.txt); with the bijective remap
    // of gi_gemm.hip one XCD walks CONSECUTIVE blocks, i.e. one or two types.  GI_CHAIN_XCD=0: dispatch order.
    int remap;
    int dbg;                                // TIMING-ONLY lab switches of the fp16x2 kernel (GI_DBG_X2 bit mask; results wrong)
};
// The TIMING-ONLY lab switches (results wrong) exist in a lab build only (make EXTRA=-DGI_CHAIN_X2_LAB): release kernels
// carry no such branches (round-5 advisor).
#ifdef GI_CHAIN_X2_LAB
#define CH_DBG(m) (args.dbg & (m))
#else
#define CH_DBG(m) 0
#endif
__device__ __forceinline__ int chain_block_id(const ChainArgs& a, int bid) {
    const int total = a.chain_off[a.nchains];
    if (!a.remap || (int)gridDim.x != total) return bid;
    const int q = total >> 3, r = total & 7, xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

__host__ __device__ inline int chain_tiles(const gi_chain_params& p) {
    int t = 0;
    for (int l = 0; l < p.nlayers; ++l) t += (p.layer[l].K + CH_KT - 1) / CH_KT;
    return t;
}

// ---- weight image ------------------------------------------------------------------------------------
// forward tile (layer l, k tile kt): 256 rows n x 8 chunks of 4 floats; LDS position p of row n holds the
// k chunk  p ^ ((n >> 1) & 7)  — with 128-byte rows, rows n, n+2, ... would otherwise share banks; the XOR
// makes every 16-lane group of a ds_read_b128 fragment read conflict-free.
// backward tile: 32 reduction rows k x 256 columns n (fragments by ds_read_b32, consecutive lanes =
// consecutive n: conflict-free as is).
struct PackArgs {
    gi_chain_params c;
    int tiles;
    long long stride16;                     // fp16x2 image: 16-byte units per group
};

__global__ __launch_bounds__(256) void gi_chain_pack_kernel(const PackArgs a) {
    const gi_chain_params& P = a.c;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;      // one float4 of the image each
    const long long per_group = (long long)a.tiles * (CH_TILE / 4);
    if (id >= per_group * P.ngroups) return;
    const int g = (int)(id / per_group);
    const int rem = (int)(id - (long long)g * per_group);
    int tile = rem / (CH_TILE / 4);
    const int q = rem - tile * (CH_TILE / 4);
    int l = 0;
    for (;;) {                                                            // which layer / k tile
        const int nk = (P.layer[l].K + CH_KT - 1) / CH_KT;
        if (tile < nk) break;
        tile -= nk; ++l;
    }
    const gi_chain_layer& Ly = P.layer[l];
    const float* __restrict__ W = Ly.W[g];
    const int K = Ly.K, N = Ly.N;
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (!P.backward) {
        const int n = q >> 3, p = q & 7;
        const int k = tile * CH_KT + 4 * (p ^ ((n >> 1) & 7));
        if (n < N) {
            const float* src = W + (long long)n * K;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (k + j < K) ? src[k + j] : 0.f;
        }
    } else {
        const int kr = q >> 6, n = 4 * (q & 63);
        const int k = tile * CH_KT + kr;
        if (k < K) {
            const float* src = W + (long long)k * N;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (n + j < N) ? src[n + j] : 0.f;
        }
    }
    ((v4f*)P.image)[id] = v;
}

// LDS-DMA of one 1 KB piece: every lane's 16 bytes land at LDS (m0 base) + 16 * lane.  Invisible to
// the compiler's s_waitcnt bookkeeping on purpose (its own LDS-DMA handling drains vmcnt(0) at every
// barrier): the kernel counts these loads itself, see GI_CHAIN_WAIT below.
// m0 is on the clobber list instead of being saved and restored around every piece (4 scalar instructions of a k step's
// ~50): clang warns that it is a reserved register it "may not preserve" — nothing else in this file uses m0 (gfx9+ LDS
// instructions do not), which the disassembly shows (the only writes of m0 are these).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma_1k(const float* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory", "m0");
}
#pragma clang diagnostic pop

// RB = 32-row MFMA blocks per workgroup, RING = weight tiles in LDS.  <1, 3>: the kernel described above.
// <2, 2>: 64 rows per workgroup for batches of several rounds of row blocks (ZINC / ChEMBL shapes): every
// streamed weight byte feeds twice the MFMA work, so the workgroup is MFMA-bound instead of bound by its
// weight stream (the second activation tile takes the LDS of the third ring slot: one tile of look-ahead,
// hidden behind 2 x the MFMA time per tile); no VALU rows in this variant.
template <bool BWD, int RB, int RING>
__global__ __launch_bounds__(512) void gi_chain_kernel(const ChainArgs args) {
    constexpr int MROWS = CH_ROWS * RB;                       // rows on the MFMA path
    __shared__ __attribute__((aligned(16))) float As[(MROWS + 8) * CH_ALD];   // MFMA rows + 4 extra (+ 4 scratch)
    __shared__ __attribute__((aligned(1024))) float Bs[RING * CH_TILE];
    __shared__ float Xs[2 * CH_XMAX * CH_W];                 // extra rows: partial sums of the k halves

    // A workgroup walks row blocks id = blockIdx.x, + gridDim.x, ...: ONE each in an ordinary launch (the grid is
    // the number of row blocks); in a bounded launch the grid is capped at the number of CUs and the bound's
    // surplus blocks cost an iteration of a few scalar loads instead of a 114 KB-LDS workgroup launch each
    // (measured: 1 364 surplus workgroups per chain launch made the bounded forward 0.3 ms slower).
    if (args.c[0].skip_flag && *args.c[0].skip_flag != 0) return;   // rows served from gi_graph.p0_cache
    for (int bid = blockIdx.x; bid < args.chain_off[args.nchains]; bid += gridDim.x) {
    const int id = chain_block_id(args, bid);
    // ---- which (chain, group, row block) -------------------------------------------------------
    const int ci = (args.nchains > 1 && id >= args.chain_off[1]) ? 1 : 0;
    const gi_chain_params& P = args.c[ci];
    const int local = id - args.chain_off[ci];
    int g = 0, first = 0;
    const int tile_rows = args.tile_rows_dev ? __builtin_amdgcn_readfirstlane(max(*args.tile_rows_dev, CH_ROWS))
                                             : args.tile_rows;
    if (args.dev_tiles) {                                    // row blocks of the groups, counted on the device
        for (; g < P.ngroups; ++g) {
            const int rows = __builtin_amdgcn_readfirstlane(P.grp_off[g + 1] - P.grp_off[g]);
            const int nb = (rows + tile_rows - 1) / tile_rows;
            if (local < first + nb) break;
            first += nb;
        }
        if (g == P.ngroups) continue;                        // beyond the real row blocks
    } else {
        while (g < P.ngroups - 1 && local >= args.tile_off[ci][g + 1]) ++g;
        first = args.tile_off[ci][g];
    }
    // (device loads land in VGPRs; readfirstlane tells the compiler the row range is wave-uniform,
    // so everything derived from it — buffer descriptors included — stays scalar)
    const int lo = P.grp_off ? __builtin_amdgcn_readfirstlane(P.grp_off[g]) : 0;
    const int hi = P.grp_off ? __builtin_amdgcn_readfirstlane(P.grp_off[g + 1]) : P.rows;
    const int r0 = lo + tile_rows * (local - first);
    if (r0 >= hi) continue;                                 // block-uniform, before any barrier
    const int nvalid = min(hi - r0, tile_rows);             // rows of this block
    const int nx = __builtin_amdgcn_readfirstlane(max(nvalid - MROWS, 0));     // on the VALU path
    const int L = P.nlayers;
    const long long t_start = args.trace ? (long long)wall_clock64() : 0;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int swid = __builtin_amdgcn_readfirstlane(wid);   // wave-uniform copy for scalar use
    const int l31 = lane & 31, lhi = lane >> 5;
    const int T = chain_tiles(P);                            // steps = weight tiles of the whole chain

    // ---- weight stream: tile t of this group -> ring slot t % 3; a wave moves 4 of the 32 pieces ----
    const float* const img = P.image + (long long)g * (P.image_stride ? P.image_stride : (long long)T * CH_TILE) +
                             (swid * 4) * 256 + lane * 4;
    const unsigned bs_lds = (unsigned)(uintptr_t)Bs + (unsigned)(swid * 4) * 1024u;
    auto dma_tile = [&](int t) {
        t = min(t, T - 1);                                   // past the end: re-fetch the last tile
        const float* src = img + (long long)t * CH_TILE;     // (keeps the outstanding-load count fixed)
        const unsigned dst = bs_lds + (unsigned)(t % RING) * (unsigned)(CH_TILE * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_dma_1k(src + q * 256, dst + q * 1024u);
    };

    f32x16 acc[RB];
    float xacc[CH_XMAX] = {0.f, 0.f, 0.f, 0.f};             // extra rows: column xn, k half xh
    const int xn = tid & (CH_W - 1), xh = tid >> 8;
    auto read_frags = [&](int slot, int kt, int k8, float (&af)[RB][4], float (&bf)[4]) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const v4f a = *(const v4f*)&As[(rb * CH_ROWS + l31) * CH_ALD + kt * CH_KT + k8 * 8 + 4 * lhi];
            af[rb][0] = a.x; af[rb][1] = a.y; af[rb][2] = a.z; af[rb][3] = a.w;
        }
        const float* b = Bs + slot * CH_TILE;
        if (!BWD) {
            const int row = wid * 32 + l31;
            const v4f v = *(const v4f*)&b[row * CH_KT + 4 * ((2 * k8 + lhi) ^ ((row >> 1) & 7))];
            bf[0] = v.x; bf[1] = v.y; bf[2] = v.z; bf[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = b[(k8 * 8 + j + 4 * lhi) * CH_W + wid * 32 + l31];
        }
    };
    auto mma = [&](const float (&af)[RB][4], const float (&bf)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rb][j], bf[j], acc[rb], 0, 0, 0);
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
        const int nrows = min(nvalid, MROWS);
        const int coff = col_ok ? 4 * col : 0x40000000;     // beyond any tile: dropped / reads 0
        float av[RB][16];
        const bool dselu = BWD && Ly.act != nullptr;
        if (dselu && CH_DBG(1)) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) av[rb][r] = 1.f;
        } else
        if (dselu) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(Ly.act + (long long)r0 * Ly.ldact), 0, nrows * Ly.ldact * 4, 0x00020000);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb * CH_ROWS + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    av[rb][r] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(ra, row * Ly.ldact * 4 + coff, 0, 0));
                }
        }
        const float bv = BWD ? 0.f : Ly.bias[g][col_ok ? col : N - 1];
        float v[RB][16];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[rb][r] + bv;
                if (!BWD) x = gi_selu(x);
                if (dselu) x *= gi_selu_grad(av[rb][r]);
                v[rb][r] = col_ok ? x : 0.f;                 // zero = the next layer's k padding
            }
        float xb = 0.f, xact[CH_XMAX];
        if (nx > 0) {                                        // extra rows: publish both k halves
#pragma unroll
            for (int xi = 0; xi < CH_XMAX; ++xi) Xs[(xh * CH_XMAX + xi) * CH_W + xn] = xacc[xi];
            if (tid < CH_W) {                                // their bias / activation loads: in flight
                const int nc = tid < N ? tid : N - 1;        // across the barrier below
                if (!BWD) xb = Ly.bias[g][nc];
#pragma unroll
                for (int xi = 0; xi < CH_XMAX; ++xi)
                    xact[xi] = dselu ? Ly.act[(long long)(r0 + MROWS + min(xi, nx - 1)) * Ly.ldact + nc]
                                     : 0.f;
            }
        }
        // Drain this wave's DMA queue (the two tiles in flight belong to the next two steps; their
        // being complete is what lets those steps' waits ignore the stores issued below), then: every
        // wave is past its last read of the activation tile.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nx > 0 && tid < CH_W) {                          // one output column per thread
            const int n = tid;
#pragma unroll
            for (int xi = 0; xi < CH_XMAX; ++xi) {
                if (xi < nx) {
                    const long long grow = r0 + MROWS + xi;
                    float x = (Xs[xi * CH_W + n] + Xs[(CH_XMAX + xi) * CH_W + n]) + xb;
                    if (!BWD) x = gi_selu(x);
                    if (dselu) x *= gi_selu_grad(xact[xi]);
                    x = (n < N) ? x : 0.f;
                    if (l + 1 < L) As[(MROWS + xi) * CH_ALD + n] = x;
                    if (n < N) Ly.out[grow * ldo + n] = x;
                }
            }
        }
#pragma unroll
        for (int xi = 0; xi < CH_XMAX; ++xi) xacc[xi] = 0.f;
        if (l + 1 < L) {                                     // next layer's A operand, in place
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    As[(rb * CH_ROWS + (r & 3) + 8 * (r >> 2) + 4 * lhi) * CH_ALD + col] = v[rb][r];
        }
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Ly.out + (long long)r0 * ldo), 0, nrows * ldo * 4, 0x00020000);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * CH_ROWS + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[rb][r]), ro,
                                                      row * ldo * 4 + coff, 0, 0);
            }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
    };

    // ---- prologue: the input rows -> LDS (zero beyond K0); then the first two weight tiles ----------
    {
        const int mc4 = tid & 63, mrow = tid >> 6;
        const int K0 = P.layer[0].K, cmax = ((K0 + 3) & ~3) - 4;
        const int c = 4 * mc4;
        constexpr int NP = 4 * RB + 1;                      // passes of 8 rows: MROWS + 8
        long long src[NP];
        v4f v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) src[i] = min(r0 + mrow + 8 * i, hi - 1);
        if (P.x_idx) {
#pragma unroll
            for (int i = 0; i < NP; ++i) src[i] = P.x_idx[src[i]];
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i] = gi_load4_raw(P.X + src[i] * P.ldx, c, cmax);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            v4f w = v[i];
            w.x = (c < K0) ? w.x : 0.f; w.y = (c + 1 < K0) ? w.y : 0.f;
            w.z = (c + 2 < K0) ? w.z : 0.f; w.w = (c + 3 < K0) ? w.w : 0.f;
            *(v4f*)&As[(mrow + 8 * i) * CH_ALD + c] = w;
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
    __syncthreads();                        // the A tile is in LDS (and every plain load has landed)
    dma_tile(0);
    if (RING == 3) dma_tile(1);
    long long t_phase[GI_CHAIN_MAXL + 1];
    if (args.trace) t_phase[0] = (long long)wall_clock64();

    // ---- main loop over the weight tiles -----------------------------------------------------------
    // Step s: wait until THIS wave's pieces of tile s have landed, barrier (=> the whole tile has, and
    // every wave is done with tile s-1, whose ring slot is the one tile s+2 goes to), start the DMA of
    // tile s+2, multiply tile s.  Counting the loads: they complete in issue order, and right before
    // the wait of step s the youngest four are tile s+1's, so "at most 4 outstanding" means tile s is
    // complete (other outstanding operations — the epilogue's stores — only make the wait longer).
    // For the two steps after an epilogue the tiles needed were already drained by its __syncthreads,
    // and up to 16 + 4 stores + 4 loads are younger: waiting for them would only stall.
    // lgkmcnt(0): this wave's LDS writes (epilogue) are visible before the barrier releases readers.
#define GI_CHAIN_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
    float af[2][RB][4], bf[2][4];
    // (loop state kept provably wave-uniform — readfirstlane — so that every branch on it is scalar)
    int l = 0, kt = 0, nk = (P.layer[0].K + CH_KT - 1) / CH_KT, lN = P.layer[0].N;
    int since_epi = 2;
    for (int s = 0; s < T; ++s) {
        if (RING == 3) {
            if (__builtin_amdgcn_readfirstlane(since_epi) < 2) { GI_CHAIN_WAIT(24); } else { GI_CHAIN_WAIT(4); }
        } else {
            // two slots: tile s is the youngest load (issued in step s-1) unless an epilogue came after it
            // — which drained it (vmcnt(0) + barrier) and left only its own stores (<= 16 RB + 4) in flight
            if (__builtin_amdgcn_readfirstlane(since_epi) < 1) { GI_CHAIN_WAIT(40); } else { GI_CHAIN_WAIT(0); }
        }
        since_epi = __builtin_amdgcn_readfirstlane(since_epi + 1);
        dma_tile(s + RING - 1);
        const int slot = s % RING;
        if (swid * 32 < __builtin_amdgcn_readfirstlane(lN)) {   // this wave owns output columns of the layer
            read_frags(slot, kt, 0, af[0], bf[0]);
#pragma unroll
            for (int q = 0; q < CH_KT / 8; ++q) {
                __builtin_amdgcn_sched_barrier(0);
                mma(af[q & 1], bf[q & 1]);
                if (q + 1 < CH_KT / 8) read_frags(slot, kt, q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nx > 0) {                              // rows 32..: fp32 fma dot products on the VALU
            const float* b = Bs + slot * CH_TILE;
            float w[16];
            if (!BWD) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const v4f v = *(const v4f*)&b[xn * CH_KT + 4 * ((4 * xh + c) ^ ((xn >> 1) & 7))];
                    w[4 * c] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) w[k] = b[(16 * xh + k) * CH_W + xn];
            }
#pragma unroll
            for (int xi = 0; xi < CH_XMAX; ++xi) {
                if (xi < nx) {
                    const float* a = &As[(MROWS + xi) * CH_ALD + kt * CH_KT + 16 * xh];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const v4f v = *(const v4f*)&a[4 * c];    // same address in every lane: broadcast
                        xacc[xi] = fmaf(v.x, w[4 * c], xacc[xi]);
                        xacc[xi] = fmaf(v.y, w[4 * c + 1], xacc[xi]);
                        xacc[xi] = fmaf(v.z, w[4 * c + 2], xacc[xi]);
                        xacc[xi] = fmaf(v.w, w[4 * c + 3], xacc[xi]);
                    }
                }
            }
        }
        kt = __builtin_amdgcn_readfirstlane(kt + 1);
        if (kt == __builtin_amdgcn_readfirstlane(nk)) {       // layer done
            epilogue(l);
            since_epi = 0;
            if (args.trace) t_phase[l + 1] = (long long)wall_clock64();
            l = __builtin_amdgcn_readfirstlane(l + 1);
            if (l < L) {
                kt = 0;
                nk = __builtin_amdgcn_readfirstlane((P.layer[l].K + CH_KT - 1) / CH_KT);
                lN = __builtin_amdgcn_readfirstlane(P.layer[l].N);
            }
        }
    }
#undef GI_CHAIN_WAIT
    if (args.trace && threadIdx.x == 0) {   // 100 MHz wall clock: start, end, placement, rows, phases
        long long* t = args.trace + 16 * (long long)id;
        t[0] = t_start; t[1] = (long long)wall_clock64();
        t[2] = ((long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) << 8) |
               (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xff);   // HW_ID, XCC_ID
        t[3] = nvalid;
        for (int i = 0; i <= L && i <= GI_CHAIN_MAXL; ++i) t[4 + i] = t_phase[i];
    }
    __syncthreads();                        // every wave is done with this block's LDS before the next one's
    }
}

// ======================================================================================================
// fp16x2 variant (gi_chain_params.x2_wamax != NULL; csrc/gi_x2.h): the same chain with every fp32 operand as two scaled
// fp16 values and three v_mfma_f32_32x32x16_f16 products per fp32 product — 12 MFMAs of 32 cycles per wave and 32-deep
// weight tile for 64 rows where the fp32 kernel spends 16 of 64 cycles for 32.  With the matrix work out of the way a
// workgroup can carry 64 rows without becoming MFMA-bound: HALF the weight stream per row (the stream is what bounds the
// fp32 kernel at one 32-row block per CU) on half the CUs, the other half free for the weight-gradient queue that runs
// beside every backward chain.
//   * weight image: 16 KB per 16-deep tile (one MFMA k step), [plane 2][k chunk of 8: 2][column 256][8 halves] — the B
//     fragment of column n and k chunk c is one ds_read_b128 — scaled per (layer, bond type) by a power of two from
//     max |W| (gi_mlp_chain_pack computes it into x2_wamax first); streamed by LDS-DMA into a FOUR-slot ring of those
//     tiles (the same 64 KB as two 32-deep slots): with the matrix work down to 6 MFMAs per tile a step is one DMA
//     round trip, so what counts is how many tiles are in flight — three instead of one (measured with two 32-deep
//     slots: 60 us per launch alone, 5.6 % MFMA busy);
//   * activation tile: [plane 2][k chunk of 8: 32][row 64][8 halves] = 64 KB, scaled PER ROW BLOCK AND LAYER by a power
//     of two from the largest magnitude of the 64 x 256 tile (a maximum per wave -> LDS -> the epilogue's own barrier):
//     no global atomics; the epilogue splits the new activations and rewrites the tile in place.  The last bits of a
//     row therefore depend on which rows share its block — which is why gi_ggnn_backward uses this variant for its dZ
//     chains and gi_ggnn_forward keeps the fp32 chain, whose rows are bit-independent (blocking against host-sync-free
//     forward, forwards with and without a tape, the pass-0 row cache).  A per-ROW scale was built and measured: exact row independence,
//     and 160 cross-lane maxima per layer and thread that cost more than the variant gains (profiles/r04/chain_x2/);
//   * C = acc / (sa sb) + bias in the epilogue; everything behind it (SELU, SELU', HBM stores) as in the fp32 kernel.
// 64 rows per workgroup, no extra VALU rows; bounded launches walk 64-row blocks (the device-side height is not used).
constexpr int CX_ROWS = 64;
constexpr int CX_PLANE = 32 * CX_ROWS * 16;                 // bytes of one activation plane (32 KB)
constexpr int CX_KT = 16;                                   // reduction depth of one weight tile
constexpr int CX_TILE = CH_W * CX_KT;                       // floats (= 4-byte units) per weight tile image: 16 KB
constexpr int CX_RING = 4;
// one workgroup's largest magnitude -> its slot of an amax cell (ONE thread calls; gi_x2.h: 64 slots, a line apart)
__device__ __forceinline__ void cx_amax_publish_wg(float m, float* cell) {
    float* slot = cell + (blockIdx.x & (GX_AMAX_SLOTS - 1)) * GX_AMAX_STRIDE;
    if (m > *reinterpret_cast<volatile float*>(slot))
        atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, m));
}
__host__ __device__ inline int cx_tiles(const gi_chain_params& p) {
    int t = 0;
    for (int l = 0; l < p.nlayers; ++l) t += (p.layer[l].K + CX_KT - 1) / CX_KT;
    return t;
}
// rows rotate by two (32 B = 8 banks) per k chunk: the epilogue's lanes write four adjacent chunks (1 KB apart: the same
// banks without the rotation) at once; the fragment reads stay 32 consecutive rows (mod 64) of one chunk
__device__ __forceinline__ unsigned cx_a_off(int plane, int chunk, int row) {
    return (unsigned)(plane * CX_PLANE + (chunk * CX_ROWS + ((row + 2 * chunk) & (CX_ROWS - 1))) * 16);
}
__device__ __forceinline__ unsigned cx_b_off(int plane, int chunk, int col) {
    return (unsigned)(((plane * 2 + chunk) * CH_W + col) * 16);
}

__global__ __launch_bounds__(256) void gi_chain_pack_x2_kernel(const PackArgs a) {
    const gi_chain_params& P = a.c;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece (8 halves of a plane) each
    const long long per_group = (long long)a.tiles * (CX_TILE / 4);      // (whole waves only: 1 024 pieces per tile)
    const int g = (int)(id / per_group);
    const int rem = (int)(id - (long long)g * per_group);
    int tile = rem / (CX_TILE / 4);
    const int q = rem - tile * (CX_TILE / 4);
    const int tile_in_group = tile;
    int l = 0;
    for (;;) {
        const int nk = (P.layer[l].K + CX_KT - 1) / CX_KT;
        if (tile < nk) break;
        tile -= nk; ++l;
    }
    const gi_chain_layer& Ly = P.layer[l];
    const float* __restrict__ W = Ly.W[g];
    const int K = Ly.K, N = Ly.N;
    const int plane = q >> 9, chunk = (q >> 8) & 1, n = q & (CH_W - 1);
    const int k0 = tile * CX_KT + chunk * 8;
    float s, inv;
    gx_scale(gx_amax_read(P.x2_wamax + ((long long)l * P.ngroups + g) * GI_AMAX_WORDS), s, inv);
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const bool ok = n < N && k < K;
        const long long at = !P.backward ? (long long)n * K + k : (long long)k * N + n;   // B[n][k] = W[n][k] / W[k][n]
        w[j] = ok ? W[ok ? at : 0] : 0.f;
    }
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned p0, p1;
        gx_split2(w[2 * j], w[2 * j + 1], s, p0, p1);
        o[j] = plane ? p1 : p0;
    }
    typedef unsigned cx_u32x4 __attribute__((ext_vector_type(4)));
    cx_u32x4 v = {o[0], o[1], o[2], o[3]};
    // a group's tiles start at the group stride of the fp32 layout (what image_stride means to every caller)
    reinterpret_cast<cx_u32x4*>(P.image)[(long long)g * a.stride16 + (long long)tile_in_group * (CX_TILE / 4) + q] = v;
}

// The same image AND the max |W| cells it is scaled by, in ONE launch (round 6): memset + gi_absmax + the kernel above were
// three dependent launches (3 + 7 + 12 us and two gaps) in front of the forward's first chain, i.e. on the critical path
// of every training step.  CXF_SPLIT workgroups per weight matrix: each takes the largest magnitude of the WHOLE matrix
// itself (<= 256 KB out of L2; the same reduction order in every workgroup, so they agree bit for bit), workgroup 0 of
// the matrix writes all 64 slots of its cell (nothing to zero beforehand), and each packs every CXF_SPLIT-th tile.
constexpr int CXF_SPLIT = 8;
__global__ __launch_bounds__(1024) void gi_chain_pack_x2_fused_kernel(const PackArgs a) {
    static_assert(CX_TILE / 4 == 1024, "one 16-byte piece of a tile per thread");
    const gi_chain_params& P = a.c;
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int mat = blockIdx.x / CXF_SPLIT, part = blockIdx.x - mat * CXF_SPLIT;
    const int l = mat / P.ngroups, g = mat - l * P.ngroups;
    const gi_chain_layer& Ly = P.layer[l];
    const float* __restrict__ W = Ly.W[g];
    const int K = Ly.K, N = Ly.N, n_el = K * N;
    float m = 0.f;
    // eight loads in flight (clamped: a repeated element changes no maximum); 16-byte loads when the matrix allows
    if ((((uintptr_t)W & 15) | (n_el & 3)) == 0) {
        const int n4 = n_el >> 2;
        for (int base = tid; base < n4; base += 8 * 1024) {
            v4f v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const v4f*>(W)[min(base + u * 1024, n4 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
        }
    } else {
        for (int base = tid; base < n_el; base += 8 * 1024) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = W[min(base + u * 1024, n_el - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(m, fabsf(v[u]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wid] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    float* cell = P.x2_wamax + ((long long)l * P.ngroups + g) * GI_AMAX_WORDS;
    if (part == 0 && tid < GX_AMAX_SLOTS) cell[tid * GX_AMAX_STRIDE] = tid ? 0.f : m;
    float s, inv;
    gx_scale(m, s, inv);
    int tile0 = 0;                                           // the layer's first tile within its group's image
    for (int j = 0; j < l; ++j) tile0 += (P.layer[j].K + CX_KT - 1) / CX_KT;
    const int nk = (K + CX_KT - 1) / CX_KT;
    const int q = tid, plane = q >> 9, chunk = (q >> 8) & 1, n = q & (CH_W - 1);
    typedef unsigned cx_u32x4 __attribute__((ext_vector_type(4)));
    for (int tile = part; tile < nk; tile += CXF_SPLIT) {
        const int k0 = tile * CX_KT + chunk * 8;
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            const bool ok = n < N && k < K;
            const long long at = !P.backward ? (long long)n * K + k : (long long)k * N + n;   // B[n][k] = W[n][k] / W[k][n]
            w[j] = ok ? W[ok ? at : 0] : 0.f;
        }
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned p0, p1;
            gx_split2(w[2 * j], w[2 * j + 1], s, p0, p1);
            o[j] = plane ? p1 : p0;
        }
        cx_u32x4 v = {o[0], o[1], o[2], o[3]};
        reinterpret_cast<cx_u32x4*>(P.image)[(long long)g * a.stride16 + (long long)(tile0 + tile) * (CX_TILE / 4) + q] = v;
    }
}

template <bool BWD>
__global__ __launch_bounds__(512) void gi_chain_x2_kernel(const ChainArgs args) {
    __shared__ __attribute__((aligned(16))) unsigned char Ah[2 * CX_PLANE];
    __shared__ __attribute__((aligned(1024))) float Bs[CX_RING * CX_TILE];
    __shared__ float red[8];                                // per wave: max |new activation|
    typedef unsigned cx_u32x2 __attribute__((ext_vector_type(2)));
    if (args.c[0].skip_flag && *args.c[0].skip_flag != 0) return;
    for (int bid = blockIdx.x; bid < args.chain_off[args.nchains]; bid += gridDim.x) {
    const int id = chain_block_id(args, bid);
    const int ci = (args.nchains > 1 && id >= args.chain_off[1]) ? 1 : 0;
    const gi_chain_params& P = args.c[ci];
    const int local = id - args.chain_off[ci];
    int g = 0, first = 0;
    if (args.dev_tiles) {                                    // row blocks of the groups, counted on the device
        for (; g < P.ngroups; ++g) {
            const int rows = __builtin_amdgcn_readfirstlane(P.grp_off[g + 1] - P.grp_off[g]);
            const int nb = (rows + CX_ROWS - 1) / CX_ROWS;
            if (local < first + nb) break;
            first += nb;
        }
        if (g == P.ngroups) continue;                        // beyond the real row blocks
    } else {
        while (g < P.ngroups - 1 && local >= args.tile_off[ci][g + 1]) ++g;
        first = args.tile_off[ci][g];
    }
    const int lo = P.grp_off ? __builtin_amdgcn_readfirstlane(P.grp_off[g]) : 0;
    const int hi = P.grp_off ? __builtin_amdgcn_readfirstlane(P.grp_off[g + 1]) : P.rows;
    const int r0 = lo + CX_ROWS * (local - first);
    if (r0 >= hi) continue;                                 // block-uniform, before any barrier
    const int nvalid = min(hi - r0, CX_ROWS);
    const int L = P.nlayers;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int swid = __builtin_amdgcn_readfirstlane(wid);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int T = cx_tiles(P);                               // steps = 16-deep weight tiles of the whole chain

    // weight stream: tile t of this group -> ring slot t % 4.  WAVE-PRIVATE (round 5, see gi_chain_x2r_kernel): a wave
    // multiplies output columns 32 wid .. 32 wid + 31, i.e. it needs 2 KB of every tile — [plane][k half][its 32
    // columns][8 halves] — and fetches exactly those (lanes 0-31: k half 0, lanes 32-63: k half 1; one instruction per
    // plane) into its own 2 KB of the slot, in fragment order: no barrier in the k loop.
    const float* const img = P.image + (long long)g * (P.image_stride ? P.image_stride
                                                                      : (long long)chain_tiles(P) * CH_TILE) +
                             (lhi * CH_W + swid * 32 + l31) * 4;
    const unsigned bs_lds = (unsigned)(uintptr_t)Bs + (unsigned)swid * 2048u;
    auto dma_tile = [&](int t) {
        t = min(t, T - 1);                                   // past the end: re-fetch the last tile (fixed load count)
        const float* src = img + (long long)t * CX_TILE;
        const unsigned dst = bs_lds + (unsigned)(t & (CX_RING - 1)) * (unsigned)(CX_TILE * 4);
#pragma unroll
        for (int q = 0; q < 2; ++q) lds_dma_1k(src + q * (2 * CH_W * 4), dst + q * 1024u);      // plane q
    };
    // 1 / scale of a layer's weights (this bond type): layer 0 here, layer l + 1 at the end of epilogue l
    auto w_inv_scale = [&](int l) {
        float s_, inv_;
        gx_scale(gx_amax_read(P.x2_wamax + ((long long)l * P.ngroups + g) * GI_AMAX_WORDS), s_, inv_);
        return inv_;
    };
    float ib = w_inv_scale(0);
    // largest magnitude over the workgroup: wave maximum -> LDS; the caller's barrier; then every thread reads all 8
    auto wave_max_to_lds = [&](float m) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[wid] = m;
    };
    auto wg_max_from_lds = [&]() {
        float m = red[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
        return m;
    };
    f32x16 acc[2];
    float sa = 1.f, ia = 1.f;                               // scale of the activation tile in LDS and its inverse
    gx_f16x8 af[2][2], bf[2];                                // [row block][plane], [plane]
    auto read_frags = [&](int slot, int kt, gx_f16x8 (&a)[2][2], gx_f16x8 (&b)[2]) {
        const unsigned char* bt = reinterpret_cast<const unsigned char*>(Bs) + slot * (CX_TILE * 4);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                a[rb][pl] = *reinterpret_cast<const gx_f16x8*>(Ah + cx_a_off(pl, kt * 2 + lhi, rb * 32 + l31));
            b[pl] = *reinterpret_cast<const gx_f16x8*>(bt + wid * 2048 + pl * 1024 + lane * 16);
        }
    };
    auto mma = [&](const gx_f16x8 (&a)[2][2], const gx_f16x8 (&b)[2]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {                     // a2 b1 + a1 b2 + a1 b1 (smallest terms first)
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rb][1], b[0], acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rb][0], b[1], acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rb][0], b[0], acc[rb], 0, 0, 0);
        }
    };

    // (backward) the stored activations a layer's epilogue needs are requested a whole k loop ahead — behind the
    // previous epilogue / the prologue — instead of at its start (an HBM round trip in front of every epilogue)
    float av[2][16];
    auto prefetch_acts = [&](int l) {
        if (!BWD || l >= L || !P.layer[l].act) return;
        const gi_chain_layer& Ly = P.layer[l];
        const int col = wid * 32 + l31;
        const int coff = col < Ly.N ? 4 * col : 0x40000000;
        if (CH_DBG(1)) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) av[rb][r] = 1.f;
            return;
        }
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Ly.act + (long long)r0 * Ly.ldact), 0, nvalid * Ly.ldact * 4, 0x00020000);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                av[rb][r] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(ra, row * Ly.ldact * 4 + coff, 0, 0));
            }
    };

    auto epilogue = [&](int l) {
        const gi_chain_layer& Ly = P.layer[l];
        const int N = Ly.N, ldo = Ly.ldo;
        const int col = wid * 32 + l31;
        const bool col_ok = col < N;
        const int coff = col_ok ? 4 * col : 0x40000000;
        const bool dselu = BWD && Ly.act != nullptr;
        const float bv = BWD ? 0.f : Ly.bias[g][col_ok ? col : N - 1];
        float v[2][16];
        float m = 0.f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = (acc[rb][r] * ia) * ib + bv;
                if (!BWD) x = gi_selu(x);
                if (dselu) x *= gi_selu_grad(av[rb][r]);
                // zero = the next layer's k padding; rows beyond the block's (copies of its last row, with selu'(0) for a
                // factor in the dZ chain) stay out of the block's scale and of the published maximum
                x = (col_ok && rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi < nvalid) ? x : 0.f;
                v[rb][r] = x;
                m = fmaxf(m, fabsf(x));
            }
        wave_max_to_lds(m);
        // Drain this wave's DMA queue, then: every wave is past its last read of the activation tile, and the eight
        // wave maxima are in LDS.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (Ly.out_amax && tid == 0) cx_amax_publish_wg(wg_max_from_lds(), Ly.out_amax);   // for the stack's fp16x2 weight gradients
        if (l + 1 < L && !CH_DBG(4)) {                  // next layer's A operand: split, in place
            gx_scale(wg_max_from_lds(), sa, ia);
            // A lane holds ONE column of 16 rows per row block: written one by one that is 64 two-byte stores per
            // thread into 8 of the 32 banks (round 4: 45 % of the kernel's LDS cycles were bank conflicts).  Neighbouring
            // lanes (columns c, c + 1) exchange one value per ROW PAIR instead — the even lane takes both columns of the
            // pair's first row, the odd lane both of its second — and store packed dwords: half the stores, and with the
            // per-chunk row rotation of cx_a_off the 64 lanes of a wave cover all 32 banks twice (the minimum for 256 B).
            const int odd = lane & 1;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float mine0 = v[rb][2 * t], mine1 = v[rb][2 * t + 1];
                    const float got = __shfl_xor(odd ? mine0 : mine1, 1);        // the neighbour's value of MY row
                    const float lo = odd ? got : mine0, hi = odd ? mine1 : got;   // columns c & ~1, c | 1 of that row
                    const int r = 2 * t + odd;
                    const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    unsigned p0, p1;
                    gx_split2(lo, hi, sa, p0, p1);
                    const unsigned at = cx_a_off(0, col >> 3, row) + ((col & 7) >> 1) * 4;
                    *reinterpret_cast<unsigned*>(Ah + at) = p0;
                    *reinterpret_cast<unsigned*>(Ah + CX_PLANE + at) = p1;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the new planes are written (no barrier in the k loop)
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Ly.out + (long long)r0 * ldo), 0, nvalid * ldo * 4, 0x00020000);
        if (!CH_DBG(2))
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[rb][r]), ro,
                                                      row * ldo * 4 + coff, 0, 0);
            }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
        if (l + 1 < L) ib = w_inv_scale(l + 1);             // (lands under the next layer's k loop)
        prefetch_acts(l + 1);
    };

    // ---- prologue: the 64 input rows -> scale -> two fp16 planes in LDS (zero beyond K0) ---------------
    {
        const int mc4 = tid & 63, mrow = tid >> 6;
        const int K0 = P.layer[0].K, cmax = ((K0 + 3) & ~3) - 4;
        const int c = 4 * mc4;
        long long src[8];
        v4f v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) src[i] = min(r0 + mrow + 8 * i, hi - 1);
        if (P.x_idx) {
#pragma unroll
            for (int i = 0; i < 8; ++i) src[i] = P.x_idx[src[i]];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gi_load4_raw(P.X + src[i] * P.ldx, c, cmax);
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v4f w = v[i];
            w.x = (c < K0) ? w.x : 0.f; w.y = (c + 1 < K0) ? w.y : 0.f;
            w.z = (c + 2 < K0) ? w.z : 0.f; w.w = (c + 3 < K0) ? w.w : 0.f;
            v[i] = w;
            m = fmaxf(fmaxf(m, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
        }
        wave_max_to_lds(m);
        __syncthreads();
        gx_scale(wg_max_from_lds(), sa, ia);
        if (P.x_amax && tid == 0) cx_amax_publish_wg(wg_max_from_lds(), P.x_amax);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned p0a, p1a, p0b, p1b;
            gx_split2(v[i].x, v[i].y, sa, p0a, p1a);
            gx_split2(v[i].z, v[i].w, sa, p0b, p1b);
            const unsigned at = cx_a_off(0, c >> 3, mrow + 8 * i) + (c & 7) * 2;
            cx_u32x2 w0 = {p0a, p0b}, w1 = {p1a, p1b};
            *reinterpret_cast<cx_u32x2*>(Ah + at) = w0;
            *reinterpret_cast<cx_u32x2*>(Ah + CX_PLANE + at) = w1;
        }
        prefetch_acts(0);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
    __syncthreads();                        // the A planes are in LDS (and every plain load has landed)
    dma_tile(0); dma_tile(1); dma_tile(2);

    // ---- main loop over the weight tiles: four-slot ring, three tiles in flight -----------------------------
    // Step s: wait for THIS wave's share of tile s (nobody else reads it) and for its own fragment reads of tile s - 1,
    // whose slot tile s + 3 goes to; start the DMA of tile s + 3; multiply tile s.  Loads complete in issue order: right
    // before the wait of step s the youngest four are tiles s + 1 and s + 2, so "at most 4 outstanding" means tile s is
    // complete.  An epilogue drains everything (vmcnt(0): tiles up to s + 3 are in LDS) and leaves <= 36 stores and (dZ
    // chain) the next layer's 32 activation loads in flight; the three steps after it need no load to complete, and
    // waiting for those (older than the 2 + 2 loads issued since) would only stall: "at most 60".  (vmcnt is a 6-bit counter: with 36 stores + 32 prefetch loads + 4 DMA pieces
    // in flight more than 63 operations can be outstanding — the counter then saturates and vmcnt(60) waits for the oldest
    // few, which only delays issue; correctness does not depend on it: the tile a step reads was drained by the epilogue.)
#define GI_CHAIN_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory")
    int l = 0, kt = 0, nk = (P.layer[0].K + CX_KT - 1) / CX_KT, lN = P.layer[0].N;
    int since_epi = 3;
    for (int s = 0; s < T; ++s) {
        if (__builtin_amdgcn_readfirstlane(since_epi) < 3) { GI_CHAIN_WAIT(60); } else { GI_CHAIN_WAIT(4); }
        since_epi = __builtin_amdgcn_readfirstlane(since_epi + 1);
        dma_tile(s + 3);
        const int slot = s & (CX_RING - 1);
        if (swid * 32 < __builtin_amdgcn_readfirstlane(lN) && !CH_DBG(8)) {
            read_frags(slot, kt, af, bf);
            __builtin_amdgcn_sched_barrier(0);
            mma(af, bf);
            __builtin_amdgcn_sched_barrier(0);
        }
        kt = __builtin_amdgcn_readfirstlane(kt + 1);
        if (kt == __builtin_amdgcn_readfirstlane(nk)) {       // layer done
            if (CH_DBG(16)) {                              // (lab: no epilogue at all)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
            } else
            epilogue(l);
            since_epi = 0;
            l = __builtin_amdgcn_readfirstlane(l + 1);
            if (l < L) {
                kt = 0;
                nk = __builtin_amdgcn_readfirstlane((P.layer[l].K + CX_KT - 1) / CX_KT);
                lN = __builtin_amdgcn_readfirstlane(P.layer[l].N);
            }
        }
    }
#undef GI_CHAIN_WAIT
    __syncthreads();
    }
}

// ======================================================================================================
// fp16x2 chain, ROW-INDEPENDENT variant (gi_chain_params.x2_rows32; round 5): 32 rows per workgroup, every ROW of the
// activation tile scaled by its own power of two.  What the measurements of this round say about the chain kernels
// (profiles/r05: stream_lab.txt, x2_chain_breakdown.txt, chain_scaling.txt, x2r_chain_breakdown.txt):
//   * one launch takes the same time with 8 workgroups as with 256 (fp32 49 -> 58 us, fp16x2 64-row 51 -> 64 us): a
//     workgroup is a DEPENDENT CHAIN of phases — 72 k steps of (wait for the tile, fragment reads, MFMAs), five
//     epilogues of ~1000 instructions per wave — and nothing on the chip is saturated: not the L2 (the bare stream of a
//     1.15 MB image per workgroup is 12.5 us), not the matrix pipe (3 - 6 us), not HBM;
//   * so the levers are instructions and waits per workgroup, and a SECOND workgroup per CU to fill the gaps.
// This variant
//   * transposes the product: C^T = W . A^T — the MFMA's A operand is the WEIGHT fragment (32 output channels x 16 k),
//     its B operand the ACTIVATION fragment (16 k x 32 rows); same fragments, same LDS reads, operands swapped.  A lane
//     then holds ONE row (lane & 31) and 16 output channels of it: the row's largest magnitude is 16 in-register maxima,
//     one cross-half exchange and the other waves' 7 values through LDS — the per-row scale that cost "160 cross-lane
//     maxima per thread" in the row-major layout (round 4) is nearly free, and 1 / scale of the row is a per-lane
//     constant in the epilogue;
//   * so every row's result depends on nothing but the row and the weights: bit-independent of which rows share its
//     block — usable by gi_ggnn_forward (pass-0 row cache, blocking == host-sync-free, tape == no tape);
//   * WAVE-PRIVATE weight rings: a wave multiplies output channels 32 wid .. 32 wid + 31, i.e. it needs 2 KB of every
//     16 KB tile, and fetches exactly those bytes itself, in fragment order, into its own part of the ring slot.  No
//     wave reads what another loaded: the k loop has NO barrier (the 64-row kernel has 72), only the wave's own vmcnt;
//   * biases of all layers sit in LDS from the prologue on (16 dependent global loads per lane and layer before);
//     the backward's stored activations are requested a whole k loop ahead of the epilogue that needs them;
//   * writes the next layer's planes as 8-byte pieces, 32 lanes = 32 adjacent rows of one k chunk: conflict-free by
//     construction;
//   * DUAL = false (launches of at most one workgroup per CU): four-slot ring, three tiles in flight; fp32 outputs
//     staged through LDS and stored as whole rows — 138 KB of LDS, one workgroup per CU, 43 us at 256 workgroups;
//   * DUAL = true (more row blocks than CUs): TWO workgroups per CU — two-slot ring with two tiles in flight (a slot is
//     re-requested as soon as its fragments are in registers),
//     no staging tile (a lane stores its row's 4 x 16 bytes per layer itself, the L2 merges them into lines), 73 KB of
//     LDS and <= 128 VGPRs.  264 blocks: 56 us against 76 (two rounds); 512: 64 against 82; 1024: 121 against 159.
constexpr int CR_ROWS = 32;                                 // rows of the (transposed) MFMA tile
constexpr int CR_OLD = CH_W + 4;                            // floats per row of the output staging tile

template <bool BWD, bool DUAL>
__global__ __launch_bounds__(512, DUAL ? 4 : 1) void gi_chain_x2r_kernel(const ChainArgs args) {
    constexpr int PLANE = 32 * CR_ROWS * 16;                // bytes of one activation plane: [k chunk 32][row][8 halves]
    constexpr int RING = DUAL ? 2 : CX_RING;
    __shared__ __attribute__((aligned(16))) unsigned char Ah[2 * PLANE];
    __shared__ __attribute__((aligned(1024))) float Bs[RING * CX_TILE];
    __shared__ __attribute__((aligned(16))) float Os[DUAL ? 4 : CR_ROWS * CR_OLD];
    __shared__ float red[8][CR_ROWS];                       // per wave: max |new activation| of every row
    __shared__ __attribute__((aligned(16))) float bias_s[BWD ? 4 : GI_CHAIN_MAXL * CH_W];   // every layer's bias (0 beyond N)
    typedef unsigned cx_u32x2 __attribute__((ext_vector_type(2)));
    auto a_off = [](int plane, int chunk, int row) { return (unsigned)(plane * PLANE + (chunk * CR_ROWS + row) * 16); };
    if (args.c[0].skip_flag && *args.c[0].skip_flag != 0) return;
    for (int bid = blockIdx.x; bid < args.chain_off[args.nchains]; bid += gridDim.x) {
    const int id = chain_block_id(args, bid);
    const int ci = (args.nchains > 1 && id >= args.chain_off[1]) ? 1 : 0;
    const gi_chain_params& P = args.c[ci];
    const int local = id - args.chain_off[ci];
    int g = 0, first = 0;
    if (args.dev_tiles) {                                    // row blocks of the groups, counted on the device
        for (; g < P.ngroups; ++g) {
            const int rows = __builtin_amdgcn_readfirstlane(P.grp_off[g + 1] - P.grp_off[g]);
            const int nb = (rows + CR_ROWS - 1) / CR_ROWS;
            if (local < first + nb) break;
            first += nb;
        }
        if (g == P.ngroups) continue;                        // beyond the real row blocks
    } else {
        while (g < P.ngroups - 1 && local >= args.tile_off[ci][g + 1]) ++g;
        first = args.tile_off[ci][g];
    }
    const int lo = P.grp_off ? __builtin_amdgcn_readfirstlane(P.grp_off[g]) : 0;
    const int hi = P.grp_off ? __builtin_amdgcn_readfirstlane(P.grp_off[g + 1]) : P.rows;
    const int r0 = lo + CR_ROWS * (local - first);
    if (r0 >= hi) continue;                                 // block-uniform, before any barrier
    const int nvalid = min(hi - r0, CR_ROWS);
    const int L = P.nlayers;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int swid = __builtin_amdgcn_readfirstlane(wid);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nb = wid * 32 + 4 * lhi;                       // this lane's output channels: nb + 8 j + {0..3}, j < 4
    const int T = cx_tiles(P);

    // this wave's 2 KB of a tile: [plane][k half][its 32 channels][8 halves]; lanes 0-31 fetch k half 0, lanes 32-63
    // k half 1 (512 contiguous bytes each), one instruction per plane, into the wave's own 2 KB of the ring slot
    const float* const img = P.image + (long long)g * (P.image_stride ? P.image_stride
                                                                      : (long long)chain_tiles(P) * CH_TILE) +
                             (lhi * CH_W + swid * 32 + l31) * 4;
    const unsigned bs_lds = (unsigned)(uintptr_t)Bs + (unsigned)swid * 2048u;
    auto dma_tile = [&](int t) {
        t = min(t, T - 1);                                   // past the end: re-fetch the last tile (fixed load count)
        const float* src = img + (long long)t * CX_TILE;
        const unsigned dst = bs_lds + (unsigned)(t & (RING - 1)) * (unsigned)(CX_TILE * 4);
#pragma unroll
        for (int q = 0; q < 2; ++q) lds_dma_1k(src + q * (2 * CH_W * 4), dst + q * 1024u);      // plane q
    };
    auto w_inv_scale = [&](int l) {
        float s_, inv_;
        gx_scale(gx_amax_read(P.x2_wamax + ((long long)l * P.ngroups + g) * GI_AMAX_WORDS), s_, inv_);
        return inv_;
    };
    float ib = w_inv_scale(0);
    f32x16 acc;
    float sa = 1.f, ia = 1.f;                                // scale of THIS LANE's row (l31) in the LDS planes, its inverse
    gx_f16x8 af[2], wf[2];                                   // activation / weight fragment, [plane]
    auto read_frags = [&](int slot, int kt) {
        const unsigned char* bt = reinterpret_cast<const unsigned char*>(Bs) + slot * (CX_TILE * 4);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            af[pl] = *reinterpret_cast<const gx_f16x8*>(Ah + a_off(pl, kt * 2 + lhi, l31));
            wf[pl] = *reinterpret_cast<const gx_f16x8*>(bt + wid * 2048 + pl * 1024 + lane * 16);
        }
    };
    auto mma = [&]() {                                       // C^T: w2 a1 + w1 a2 + w1 a1 (smallest terms first)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1], af[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], af[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], af[0], acc, 0, 0, 0);
    };
    // the scale of one row from the 8 waves' row maxima (after a barrier behind their red[] writes)
    auto row_scale_from_lds = [&](int row, float& s_, float& i_) {
        float m = red[0][row];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w][row]);
        gx_scale(m, s_, i_);
    };
    // (backward) the stored activations a layer's epilogue needs — this lane's row, its channels — are requested a
    // whole k loop ahead, behind the previous epilogue / the prologue: an HBM round trip is most of a 3-us epilogue
    v4f ypre[4];
    auto prefetch_acts = [&](int l) {
        if (!BWD || l >= L || !P.layer[l].act) return;
        const gi_chain_layer& Ly = P.layer[l];
        const float* arow = Ly.act + (long long)(r0 + (l31 < nvalid ? l31 : 0)) * Ly.ldact;
#pragma unroll
        for (int j = 0; j < 4; ++j) ypre[j] = gi_load4_raw(arow, nb + 8 * j, ((Ly.N + 3) & ~3) - 4);
    };

    // a layer's epilogue.  Part one: x <- the new activations of this lane's row, their largest magnitude to red[]
    auto epilogue = [&](int l) {
        const gi_chain_layer& Ly = P.layer[l];
        const int N = Ly.N, ldo = Ly.ldo;
        const bool live = l31 < nvalid;
        const bool dselu = BWD && Ly.act != nullptr;
        v4f x[4];
        {
            const float iab = ia * ib;
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v4f fac = {1.f, 1.f, 1.f, 1.f}, bv = {0.f, 0.f, 0.f, 0.f};
                if (dselu) fac = v4f{gi_selu_grad(ypre[j].x), gi_selu_grad(ypre[j].y), gi_selu_grad(ypre[j].z), gi_selu_grad(ypre[j].w)};
                if (!BWD) bv = *reinterpret_cast<const v4f*>(&bias_s[l * CH_W + nb + 8 * j]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[4 * j + e] * iab + bv[e];
                    if (!BWD) v = gi_selu(v);
                    if (dselu) v *= fac[e];
                    v = (nb + 8 * j + e < N && live) ? v : 0.f;      // zero = the next layer's k padding / rows beyond the block
                    x[j][e] = v;
                    m = fmaxf(m, fabsf(v));
                }
            }
            m = fmaxf(m, __shfl_xor(m, 32));                 // both channel halves of the row
            if (lhi == 0) red[wid][l31] = m;
        }
        // Every wave is past its last read of the activation planes, and the eight row-maximum vectors are in LDS.
        // This wave's DMA queue drains here (the tiles in flight had the VALU work above to land), which is what the
        // vmcnt arithmetic of the k loop assumes.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // Part two: the row's planes for the next layer (its own scale), its outputs
        const bool more = l + 1 < L && !CH_DBG(8);      // (lab bit 8: no rewrite of the activation planes)
        if (more) {
            row_scale_from_lds(l31, sa, ia);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned p0a, p1a, p0b, p1b;
                gx_split2(x[j].x, x[j].y, sa, p0a, p1a);
                gx_split2(x[j].z, x[j].w, sa, p0b, p1b);
                const unsigned at = a_off(0, wid * 4 + j, l31) + 8 * lhi;
                cx_u32x2 w0 = {p0a, p0b}, w1 = {p1a, p1b};
                *reinterpret_cast<cx_u32x2*>(Ah + at) = w0;
                *reinterpret_cast<cx_u32x2*>(Ah + PLANE + at) = w1;
            }
        }
        if (!DUAL) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<v4f*>(&Os[l31 * CR_OLD + nb + 8 * j]) = x[j];
        }
        float blk_max = 0.f;                                 // the block's largest new activation, for the stack's fp16x2 weight
        if (Ly.out_amax && swid == 0) {                      // gradients (read between the barriers: red[] is stable there)
#pragma unroll
            for (int w = 0; w < 4; ++w) blk_max = fmaxf(blk_max, red[4 * lhi + w][l31]);   // lane: one row, four of the eight waves
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) blk_max = fmaxf(blk_max, __shfl_xor(blk_max, o));
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the new planes are written (the output tile is staged)
        if (Ly.out_amax && swid == 0 && lane == 0) cx_amax_publish_wg(blk_max, Ly.out_amax);
        if (!CH_DBG(16)) {
            if (DUAL) {                                      // the lane's 4 x 4 channels of its row straight to HBM
                if (live) {
                    float* orow = Ly.out + (long long)(r0 + l31) * ldo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = nb + 8 * j;
                        if (c + 3 < N) *reinterpret_cast<v4f_u*>(orow + c) = x[j];
                        else {
                            if (c < N) orow[c] = x[j].x;
                            if (c + 1 < N) orow[c + 1] = x[j].y;
                            if (c + 2 < N) orow[c + 2] = x[j].z;
                        }
                    }
                }
            } else {                                         // whole rows: 16 lanes x 16 bytes per row, 4 passes over the 256 columns
                const int row = tid >> 4, c0 = 4 * (tid & 15);
                if (row < nvalid) {
                    float* orow = Ly.out + (long long)(r0 + row) * ldo;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = c0 + 64 * q;
                        const v4f v = *reinterpret_cast<const v4f*>(&Os[row * CR_OLD + c]);
                        if (c + 3 < N) *reinterpret_cast<v4f_u*>(orow + c) = v;
                        else {
                            if (c < N) orow[c] = v.x;
                            if (c + 1 < N) orow[c + 1] = v.y;
                            if (c + 2 < N) orow[c + 2] = v.z;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (more) ib = w_inv_scale(l + 1);                  // (lands under the next layer's k loop)
        prefetch_acts(l + 1);
    };

    // ---- prologue: the input rows -> per-row scale -> two fp16 planes in LDS (zero beyond K0); the biases ------------
    {
        const int mc4 = tid & 63, mrow = tid >> 6;           // a wave = one row per pass, its lanes the row's 64 float4
        const int K0 = P.layer[0].K, cmax = ((K0 + 3) & ~3) - 4;
        const int c = 4 * mc4;
        constexpr int NP = CR_ROWS / 8;
        long long src[NP];
        v4f v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) src[i] = min(r0 + mrow + 8 * i, hi - 1);
        if (P.x_idx) {
#pragma unroll
            for (int i = 0; i < NP; ++i) src[i] = P.x_idx[src[i]];
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i] = gi_load4_raw(P.X + src[i] * P.ldx, c, cmax);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            v4f w = v[i];
            const bool rv = mrow + 8 * i < nvalid;
            w.x = (rv && c < K0) ? w.x : 0.f; w.y = (rv && c + 1 < K0) ? w.y : 0.f;
            w.z = (rv && c + 2 < K0) ? w.z : 0.f; w.w = (rv && c + 3 < K0) ? w.w : 0.f;
            float m = fmaxf(fmaxf(fabsf(w.x), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w)));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));      // the row's largest magnitude
            float s_, i_;
            gx_scale(m, s_, i_);
            unsigned p0a, p1a, p0b, p1b;
            gx_split2(w.x, w.y, s_, p0a, p1a);
            gx_split2(w.z, w.w, s_, p0b, p1b);
            const unsigned at = a_off(0, c >> 3, mrow + 8 * i) + (c & 7) * 2;
            cx_u32x2 w0 = {p0a, p0b}, w1 = {p1a, p1b};
            *reinterpret_cast<cx_u32x2*>(Ah + at) = w0;
            *reinterpret_cast<cx_u32x2*>(Ah + PLANE + at) = w1;
            if (lane == 0) red[0][mrow + 8 * i] = m;         // (red[1..7] = 0: row_scale_from_lds takes the maximum)
        }
        if (wid > 0 && lane < CR_ROWS) red[wid][lane] = 0.f;
        if (!BWD)
            for (int i = tid; i < L * CH_W; i += 512) {
                const int li = i / CH_W, n = i % CH_W;
                bias_s[i] = n < P.layer[li].N ? P.layer[li].bias[g][n] : 0.f;
            }
        prefetch_acts(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __syncthreads();                        // the A planes, the row maxima and the biases are in LDS (and every plain load has landed)
    row_scale_from_lds(l31, sa, ia);
    if (P.x_amax && swid == 0) {            // largest |input| of the block (rows beyond the block were zeroed)
        float m = red[0][l31];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) cx_amax_publish_wg(m, P.x_amax);
    }
    dma_tile(0); dma_tile(1);
    if (!DUAL) dma_tile(2);

    // ---- main loop over the weight tiles.  !DUAL: four slots, three tiles in flight.  DUAL: two slots and TWO tiles in
    // flight — a slot is free as soon as the wave's fragment reads of it have returned (the MFMAs need that wait
    // anyway), so tile s + 2 is requested into the slot of tile s in front of the MFMAs of tile s.  (One tile in flight
    // left 20 us of the launch waiting for the stream, profiles/r05/x2_chain_breakdown_final.txt.)
    // No barrier in here: the rings are wave-private, the activation planes are read-only until the epilogue.
    // vmcnt arithmetic: the epilogue drained the queue, so the steps right behind it (three; DUAL: two) find their tiles
    // landed and must not wait for the epilogue's stores / the next layer's activation loads (vmcnt counts them too):
    // vmcnt(60) = no wait; from then on at most the youngest two tiles (DUAL: one) may be outstanding.
#define GI_CHAIN_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory")
    int l = 0, kt = 0, nk = (P.layer[0].K + CX_KT - 1) / CX_KT, lN = P.layer[0].N;
    int since_epi = 3;
    for (int s = 0; s < T; ++s) {
        if (__builtin_amdgcn_readfirstlane(since_epi) < (DUAL ? 2 : 3)) { GI_CHAIN_WAIT(60); }
        else if (DUAL) { GI_CHAIN_WAIT(2); } else { GI_CHAIN_WAIT(4); }
        since_epi = __builtin_amdgcn_readfirstlane(since_epi + 1);
        if (!DUAL && !CH_DBG(1)) dma_tile(s + 3);
        const int slot = s & (RING - 1);
        const bool active = swid * 32 < __builtin_amdgcn_readfirstlane(lN);
        if (active && !CH_DBG(2)) read_frags(slot, kt);
        if (DUAL) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments are in registers: the slot is free
            if (!CH_DBG(1)) dma_tile(s + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (active && !CH_DBG(4)) mma();
        __builtin_amdgcn_sched_barrier(0);
        kt = __builtin_amdgcn_readfirstlane(kt + 1);
        if (kt == __builtin_amdgcn_readfirstlane(nk)) {       // layer done
            if (CH_DBG(32)) {                              // (lab: no epilogue at all)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
            } else
            epilogue(l);
            since_epi = 0;
            l = __builtin_amdgcn_readfirstlane(l + 1);
            if (l < L) {
                kt = 0;
                nk = __builtin_amdgcn_readfirstlane((P.layer[l].K + CX_KT - 1) / CX_KT);
                lN = __builtin_amdgcn_readfirstlane(P.layer[l].N);
            }
        }
    }
#undef GI_CHAIN_WAIT
    __syncthreads();
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
    if (p.x2_wamax && ((uintptr_t)p.x2_wamax & 3)) return GI_EINVAL;
    return 0;
}

}  // namespace

extern "C" long long gi_mlp_chain_image_floats(const gi_chain_params* p) {
    if (!p) return GI_EINVAL;
    if (p->nlayers < 1 || p->nlayers > GI_CHAIN_MAXL || p->ngroups < 1 || p->ngroups > GI_MAX_GROUPS)
        return GI_EINVAL;
    return (long long)p->ngroups * chain_tiles(*p) * CH_TILE;
}

extern "C" int gi_mlp_chain_pack(const gi_chain_params* chains, int nchains, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (!chains || nchains < 1 || nchains > 2) return GI_EINVAL;
    for (int c = 0; c < nchains; ++c) {
        const gi_chain_params& p = chains[c];
        if (p.nlayers < 1 || p.nlayers > GI_CHAIN_MAXL || p.ngroups < 1 || p.ngroups > GI_MAX_GROUPS)
            return GI_EINVAL;
        for (int l = 0; l < p.nlayers; ++l) {
            const gi_chain_layer& q = p.layer[l];
            if (q.K < 4 || q.N < 4 || q.K > GI_CHAIN_MAXW || q.N > GI_CHAIN_MAXW) return GI_ELIMIT;
            for (int t = 0; t < p.ngroups; ++t)
                if (!q.W[t]) return GI_EINVAL;
        }
        if (!p.image || ((uintptr_t)p.image & 15)) return GI_EINVAL;
        if (p.image_stride && p.image_stride != (long long)chain_tiles(p) * CH_TILE) return GI_EINVAL;
        PackArgs a;
        a.c = p;
        a.tiles = chain_tiles(p);
        const long long n4 = (long long)p.ngroups * a.tiles * (CH_TILE / 4);
        const bool fused = !(getenv("GI_CHAIN_PACK_FUSED") && atoi(getenv("GI_CHAIN_PACK_FUSED")) == 0);   // (read per call: a test compares both)
        if (p.x2_wamax && fused) {          // fp16x2 image and its max |W| cells in one launch
            a.tiles = cx_tiles(p);
            a.stride16 = (p.image_stride ? p.image_stride : (long long)chain_tiles(p) * CH_TILE) / 4;
            hipLaunchKernelGGL(gi_chain_pack_x2_fused_kernel, dim3((unsigned)(p.nlayers * p.ngroups * CXF_SPLIT)), dim3(1024), 0,
                               (hipStream_t)stream, a);
        } else if (p.x2_wamax) {            // (GI_CHAIN_PACK_FUSED=0) max |W| per (layer, bond type) first, then the two planes
            const size_t cells = (size_t)p.nlayers * p.ngroups;
            if (hipMemsetAsync(p.x2_wamax, 0, cells * GI_AMAX_WORDS * sizeof(float), (hipStream_t)stream) != hipSuccess)
                return (int)hipGetLastError();
            gi_absmax_desc d[GI_ABSMAX_MAX];
            int nd = 0;
            for (int l = 0; l < p.nlayers; ++l)
                for (int t = 0; t < p.ngroups; ++t) {
                    d[nd].x = p.layer[l].W[t]; d[nd].rows = 1; d[nd].cols = p.layer[l].K * p.layer[l].N;
                    d[nd].ld = d[nd].cols; d[nd].out = p.x2_wamax + ((size_t)l * p.ngroups + t) * GI_AMAX_WORDS;
                    if (++nd == GI_ABSMAX_MAX) { const int e = gi_absmax(d, nd, stream); if (e) return e; nd = 0; }
                }
            if (nd) { const int e = gi_absmax(d, nd, stream); if (e) return e; }
            a.tiles = cx_tiles(p);
            a.stride16 = (p.image_stride ? p.image_stride : (long long)chain_tiles(p) * CH_TILE) / 4;
            const long long n16 = (long long)p.ngroups * a.tiles * (CX_TILE / 4);
            hipLaunchKernelGGL(gi_chain_pack_x2_kernel, dim3((unsigned)(n16 / 256)), dim3(256), 0, (hipStream_t)stream, a);
        } else
        hipLaunchKernelGGL(gi_chain_pack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, a);
        const int e = gi_launch_status();
        if (e) return e;
    }
    return 0;
}

// Test / measurement hook (process-wide), see the header
static struct { int tile_rows, rows64, ring; long long* trace; } g_chain_cfg = {0, -1, 2, nullptr};
extern "C" int gi_mlp_chain_config(int tile_rows, int rows64, int ring, void* trace) {
    if (ring != 2 && ring != 3) return GI_EINVAL;
    g_chain_cfg.tile_rows = tile_rows > 0 ? tile_rows : 0;
    g_chain_cfg.rows64 = rows64 < 0 ? -1 : (rows64 ? 1 : 0);
    g_chain_cfg.ring = ring;
    g_chain_cfg.trace = (long long*)trace;
    return 0;
}

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
        if ((p.x2_wamax != nullptr) != (chains[0].x2_wamax != nullptr)) return GI_EINVAL;   // one arithmetic per launch
        if (!p.image || ((uintptr_t)p.image & 15)) return GI_EINVAL;   // gi_mlp_chain_pack'ed weights
        if (p.image_stride && p.image_stride < (long long)chain_tiles(p) * CH_TILE) return GI_EINVAL;
        for (int g = 0; g < p.ngroups; ++g)
            if ((p.ngroups > 1 ? p.group_rows[g] : p.rows) < 0) return GI_EINVAL;
    }
    // Block height: 32 rows, or up to 32 + CH_XMAX when the taller blocks need one round of
    // workgroups less on this chip (one workgroup per CU).
    auto blocks = [&](int h) {
        int n = 0;
        for (int c = 0; c < nchains; ++c)
            for (int g = 0; g < chains[c].ngroups; ++g)
                n += gi_cdiv(chains[c].ngroups > 1 ? chains[c].group_rows[g] : chains[c].rows, h);
        return n;
    };
    static const int ncu = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    int h = CH_ROWS;
    int rounds = gi_cdiv(blocks(CH_ROWS), ncu);
    if (rounds > 1)
        for (int hh = CH_ROWS + 1; hh <= CH_ROWS + CH_XMAX; ++hh)
            if (gi_cdiv(blocks(hh), ncu) < rounds) { h = hh; rounds = gi_cdiv(blocks(hh), ncu); break; }
    // 64-row workgroups (two MFMA row blocks, <2, 2>: half the weight stream per row) when they need fewer
    // rounds AND the weight images of the launch do not fit one XCD's 4 MB L2 — AttentionGGNN's two stacks
    // in one launch: 6.9 MB, ChEMBL shape B=250 3.95 -> 3.81 ms per step.  A single stack (3.5 MB) streams
    // from L2 fast enough that the 32-row blocks' three full rounds tie with two rounds of 64-row blocks
    // (ZINC shape: 4.92 vs 4.98 ms; forced everywhere in round 3: headline 2.46 against 2.31 ms).
    const int rows64 = g_chain_cfg.rows64;                 // -1 auto, 0 / 1 forced (gi_mlp_chain_config)
    bool big = rows64 > 0;
    if (rows64 < 0) {
        long long image_bytes = 0;
        for (int c = 0; c < nchains; ++c)
            image_bytes += (long long)chains[c].ngroups * chain_tiles(chains[c]) * CH_TILE * 4;
        big = image_bytes > (4LL << 20) && 1.15 * gi_cdiv(blocks(2 * CH_ROWS), ncu) < 0.9 * rounds;
    }
    {   // (measurement aid) GI_CHAIN_BWD64=1: 64-row blocks for the dZ chains of the message passes only — half the
        // workgroups, so half the CUs stay free for the weight-gradient queue that runs beside every backward chain
        static const bool bwd64 = getenv("GI_CHAIN_BWD64") && atoi(getenv("GI_CHAIN_BWD64"));
        if (bwd64 && rows64 < 0 && chains[0].backward && blocks(CH_ROWS) >= ncu / 2) big = true;
    }
    if (g_chain_cfg.tile_rows > 0) {                        // tests / measurements: force a height
        h = std::min(std::max(g_chain_cfg.tile_rows, CH_ROWS), CH_ROWS + CH_XMAX);
        big = false;
    }
    // bounded launch (tile_rows_dev != NULL): `rows` is the bound of the rows of all groups together; the grid
    // covers cdiv(bound, 32) + ngroups row blocks and each workgroup finds its group and block on the device
    const bool bounded = chains[0].tile_rows_dev != nullptr;
    if (bounded) {
        for (int c = 0; c < nchains; ++c)
            if (!chains[c].tile_rows_dev || !chains[c].grp_off) return GI_EINVAL;
        big = false; h = CH_ROWS;
        a.dev_tiles = 1; a.tile_rows_dev = chains[0].tile_rows_dev;
    }
    const bool x2 = chains[0].x2_wamax != nullptr;
    const bool x2r = x2 && chains[0].x2_rows32 != 0;       // the row-independent 32-row variant
    for (int c = 0; c < nchains; ++c)
        if ((chains[c].x2_rows32 != 0) != (chains[0].x2_rows32 != 0)) return GI_EINVAL;
    if (x2) { big = false; h = x2r ? CR_ROWS : CX_ROWS; }  // (bounded too: fixed block height, the device-side height is not used)
    if (big) h = 2 * CH_ROWS;
    a.tile_rows = h;
    for (int c = 0; c < nchains; ++c) {
        const gi_chain_params& p = chains[c];
        a.c[c] = p;
        a.chain_off[c] = total;
        int t = 0;
        for (int g = 0; g < p.ngroups; ++g) {
            a.tile_off[c][g] = t;
            if (!bounded) t += gi_cdiv(p.ngroups > 1 ? p.group_rows[g] : p.rows, h);
        }
        if (bounded) t = gi_cdiv(p.rows, x2 ? (x2r ? CR_ROWS : CX_ROWS) : CH_ROWS) + p.ngroups;
        a.tile_off[c][p.ngroups] = t;
        total += t;
        for (int l = 0; l < p.nlayers; ++l)
            flops += 2.0 * (double)p.rows * (double)p.layer[l].K * (double)p.layer[l].N;
    }
    a.chain_off[nchains] = total;
    if (nchains == 1) a.chain_off[2] = total;
    a.nchains = nchains;
    if (total == 0) return 0;
    a.trace = g_chain_cfg.trace;                            // per-workgroup timestamps (tools/trace_chain.py)
    {
        static const bool xcd = !(getenv("GI_CHAIN_XCD") && atoi(getenv("GI_CHAIN_XCD")) == 0);
        a.remap = (xcd && !bounded && total >= 16) ? 1 : 0;
#ifdef GI_CHAIN_X2_LAB      // (make EXTRA=-DGI_CHAIN_X2_LAB: the breakdowns of profiles/r05/x2_chain_breakdown.txt, x2r_chain_breakdown.txt; results WRONG with a mask)
        a.dbg = getenv("GI_DBG_X2") ? atoi(getenv("GI_DBG_X2")) : 0;
#endif
    }
    hipStream_t st = (hipStream_t)stream;
    GiProfScope prof(st, GI_PROF_GEMM | (x2 ? GI_PROF_PIPE_X2 : 0), flops);
    const dim3 grid(bounded ? std::min(total, ncu) : total), block(512);
    // 32-row blocks stream their weights through a TWO-slot ring (114 KB of LDS instead of 146 KB with three):
    // one workgroup of the GEMM family (37 KB) fits on the CU beside a chain workgroup, which is what the
    // weight-gradient stream needs to overlap with the chains (every overlap schedule of round 2 was bounded by
    // chain workgroups owning their CU).  One weight tile of look-ahead less costs nothing measurable alone;
    // the step gains 1.6 % (headline 2.312 / 2.332 -> 2.277 / 2.295 ms, ZINC shape 5.044 -> 4.957, ChEMBL shape
    // 3.854 -> 3.813; profiles/r03/chain_ring_ab.txt).  gi_mlp_chain_config(ring = 3): the three-slot ring.
    // (measurement aid GI_CHAIN_RING3_SMALL=<n>: launches of at most n row blocks — the pass-0 rows: a dozen workgroups,
    // pure weight-stream latency, nothing else wants the CUs' LDS — take the three-slot ring: one more tile in flight)
    static const int ring3_small = getenv("GI_CHAIN_RING3_SMALL") ? atoi(getenv("GI_CHAIN_RING3_SMALL")) : 0;
    const bool ring2 = !big && g_chain_cfg.ring != 3 && !(total <= ring3_small);
    if (x2r) {
        // more row blocks than CUs: two workgroups per CU (GI_CHAIN_X2R_DUAL=0 / 1: never / always — measurements)
        static const int dual_env = getenv("GI_CHAIN_X2R_DUAL") ? atoi(getenv("GI_CHAIN_X2R_DUAL")) : -1;
        const bool dual = dual_env >= 0 ? dual_env != 0 : total > ncu;
        const dim3 gridx(bounded ? std::min(total, dual ? 2 * ncu : ncu) : total);
        if (chains[0].backward) {
            if (dual) hipLaunchKernelGGL((gi_chain_x2r_kernel<true, true>), gridx, block, 0, st, a);
            else hipLaunchKernelGGL((gi_chain_x2r_kernel<true, false>), gridx, block, 0, st, a);
        } else {
            if (dual) hipLaunchKernelGGL((gi_chain_x2r_kernel<false, true>), gridx, block, 0, st, a);
            else hipLaunchKernelGGL((gi_chain_x2r_kernel<false, false>), gridx, block, 0, st, a);
        }
    } else if (x2) {
        if (chains[0].backward) hipLaunchKernelGGL((gi_chain_x2_kernel<true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gi_chain_x2_kernel<false>), grid, block, 0, st, a);
    } else if (big) {
        if (chains[0].backward) hipLaunchKernelGGL((gi_chain_kernel<true, 2, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gi_chain_kernel<false, 2, 2>), grid, block, 0, st, a);
    } else if (ring2) {
        if (chains[0].backward) hipLaunchKernelGGL((gi_chain_kernel<true, 1, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gi_chain_kernel<false, 1, 2>), grid, block, 0, st, a);
    } else {
        if (chains[0].backward) hipLaunchKernelGGL((gi_chain_kernel<true, 1, CH_RING>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gi_chain_kernel<false, 1, CH_RING>), grid, block, 0, st, a);
    }
    return gi_launch_status();
}
