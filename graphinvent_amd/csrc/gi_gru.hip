// Fused message aggregation + GRU node update (gfx950): `messages = summation @ terms`
// (gnn/summation_mpnn.py:141) and `self.gru(messages, nodes)` (gnn/mpnn.py:296-297,
// torch.nn.GRUCell) for the rows of one message pass, in ONE launch.
//
// Layer by layer this was: seg_sum (12.7 us, latency-bound on 3.6 MB) -> one batched launch of the
// two projections gi = a W_ih^T + b_ih, gh = h W_hh^T + b_hh -> the gate kernel.  Here one workgroup
// owns 32 compact node rows:
//   * prologue: the segmented sum a_v = sum_{e -> v} m[in_perm[e]] over the row's dst-CSR segment
//     goes straight into LDS (the MFMA A operand; also written to HBM once, the W_ih weight gradient
//     reads it), next to the rows' hidden state h;
//   * main loop: W_ih and W_hh arrive as a pre-packed image (gi_gru_pack, once per forward: the bytes
//     of the 16-deep LDS weight tiles, zero padded, bank swizzle baked in) streamed by LDS-DMA into a
//     three-deep ring two tiles ahead of the MFMAs (see gi_chain.hip for the why); each of the
//     4 waves owns 32 hidden columns j and keeps FOUR 32x32 fp32 accumulators — r and z pre-activations
//     (gi + gh summed in the accumulator), W_in a and W_hn h — so all four values of an output
//     element meet in one lane and consecutive MFMAs are independent;
//   * epilogue: r = sigmoid, z = sigmoid, n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h for rows with
//     an incoming edge (others keep h, gnn/summation_mpnn.py:107,124,143-144); (r, z, n) and gh_n are
//     saved where gi_gru_gates_bwd expects them.
// Limits: H, M <= 128 (4 waves x 32 columns; LDS 95 KB).  Wider models use seg_sum + gi_gemm_batch +
// gi_gru_gates_fwd (same arithmetic up to fp32 summation order: gi + gh are added in the accumulator
// instead of after the two GEMMs).
#include <stdlib.h>
#include <string.h>

#include "gi_mfma.h"

namespace {

constexpr int GR_ROWS = 32;
constexpr int GR_W = GI_GRU_MAXW;            // 128: widest H / M
constexpr int GR_KT = 16;
constexpr int GR_ALD = GR_W + 4;
constexpr int GR_BROWS = 3 * GR_W;           // weight rows per tile (3 gates x H, padded)
constexpr int GR_TILE = GR_BROWS * GR_KT;    // floats per weight tile image (24 KB)
constexpr int GR_RING = 3;
constexpr int GR_PIECES = GR_TILE / 256 / 4; // 1 KB LDS-DMA pieces per wave per tile = 6

// ---- weight image: tile t < ceil(M/16): 16 reduction columns of W_ih [3H][M], then those of W_hh
// [3H][H]; 384 rows x 4 chunks of 4 floats, LDS position p of row n holds chunk p ^ ((n >> 2) & 3)
// (64-byte rows: rows n, n+4, ... would otherwise share banks in a ds_read_b128 fragment read).
__global__ __launch_bounds__(256) void gi_gru_pack_kernel(const gi_gru_params p) {
    const int H = p.H, M = p.M;
    const int nk1 = (M + GR_KT - 1) / GR_KT, nk2 = (H + GR_KT - 1) / GR_KT;
    const int id = blockIdx.x * 256 + threadIdx.x;           // one float4 of the image each
    if (id >= (nk1 + nk2) * (GR_TILE / 4)) return;
    const int t = id / (GR_TILE / 4), q = id - t * (GR_TILE / 4);
    const int n = q >> 2, pos = q & 3;
    const bool ih = t < nk1;
    const float* __restrict__ W = ih ? p.W_ih : p.W_hh;
    const int K = ih ? M : H;
    const int k = (ih ? t : t - nk1) * GR_KT + 4 * (pos ^ ((n >> 2) & 3));
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (n < 3 * H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (k + j < K) ? W[(long long)n * K + k + j] : 0.f;
    }
    ((v4f*)p.image)[id] = v;
}

__device__ __forceinline__ void gru_lds_dma_1k(const float* gsrc, unsigned lds_dst) {
    unsigned keep;          // invisible to the compiler's s_waitcnt bookkeeping on purpose (gi_chain.hip)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__global__ __launch_bounds__(256) void gi_gru_fused_kernel(const gi_gru_params p) {
    __shared__ __attribute__((aligned(16))) float A1[GR_ROWS * GR_ALD];    // aggregated messages
    __shared__ __attribute__((aligned(16))) float A2[GR_ROWS * GR_ALD];    // previous hidden state
    __shared__ __attribute__((aligned(1024))) float Bs[GR_RING * GR_TILE];
    __shared__ int edge_s[GR_ROWS];

    const long long t_start = p.trace ? (long long)wall_clock64() : 0;
    const int r0 = blockIdx.x * GR_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int H = p.H, M = p.M, R = p.R;
    const int nrows = min(R - r0, GR_ROWS);
    const int nk1 = (M + GR_KT - 1) / GR_KT, nk2 = (H + GR_KT - 1) / GR_KT, T = nk1 + nk2;

    // ---- weight stream: image tile t -> ring slot t % 3; a wave moves 6 of the 24 pieces ------------
    const int swid = __builtin_amdgcn_readfirstlane(wid);
    const float* const img = p.image + (swid * GR_PIECES) * 256 + lane * 4;
    const unsigned bs_lds = (unsigned)(uintptr_t)Bs + (unsigned)(swid * GR_PIECES) * 1024u;
    auto dma_tile = [&](int t) {
        t = min(t, T - 1);                                    // past the end: re-fetch the last tile
        const float* src = img + (long long)t * GR_TILE;
        const unsigned dst = bs_lds + (unsigned)(t % GR_RING) * (unsigned)(GR_TILE * 4);
#pragma unroll
        for (int q = 0; q < GR_PIECES; ++q) gru_lds_dma_1k(src + q * 256, dst + q * 1024u);
    };
    dma_tile(0);                    // in flight under the prologue's dependent loads
    dma_tile(1);

    // ---- prologue: A1 = segmented sum of the incoming messages (or the ready aggregate), A2 = h ----
    // Four rows per thread; the first four edges of every row are fetched branch-free (index clamped,
    // contribution selected away), so the 4 x 4 index loads and then the 4 x 4 row loads are all in
    // flight together instead of one dependent pair at a time; longer segments finish in a loop.
    {
        const int c4 = tid & 31, rr = tid >> 5;               // 32 float4 chunks per row, rows rr + 8 i
        const int col = 4 * c4;
        const bool in_m = col < M, in_h = col < H;
        int lo[4], hi[4];
        v4f a[4], h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = r0 + rr + 8 * i;
            const int cc = min(c, R - 1);
            lo[i] = p.seg_off[cc]; hi[i] = p.seg_off[cc + 1];
            if (c >= R) hi[i] = lo[i];
            a[i] = v4f{0.f, 0.f, 0.f, 0.f};
            h[i] = in_h ? gi_load4_raw(p.hx_prev + (long long)cc * p.ldhx, col, ((H + 3) & ~3) - 4)
                        : v4f{0.f, 0.f, 0.f, 0.f};
            if (c >= R) h[i] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        if (p.agg_ready) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cc = min(r0 + rr + 8 * i, R - 1);
                if (in_m) a[i] = gi_load4_raw(p.agg + (long long)cc * p.ldagg, col, ((M + 3) & ~3) - 4);
                if (r0 + rr + 8 * i >= R) a[i] = v4f{0.f, 0.f, 0.f, 0.f};
            }
        } else if (in_m) {
            int src[4][4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    src[i][e] = p.in_perm[(lo[i] + e < hi[i]) ? lo[i] + e : 0];
#pragma unroll
            for (int e = 0; e < 4; ++e)                       // ascending edge order, like seg_sum
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v4f x = *(const v4f*)(p.m + (long long)src[i][e] * p.ldm + col);
                    if (lo[i] + e < hi[i]) a[i] += x;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                for (int k = lo[i] + 4; k < hi[i]; ++k)
                    a[i] += *(const v4f*)(p.m + (long long)p.in_perm[k] * p.ldm + col);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rr + 8 * i, c = r0 + row;
            v4f av = a[i], hv = h[i];
            av.x = (col < M) ? av.x : 0.f; av.y = (col + 1 < M) ? av.y : 0.f;
            av.z = (col + 2 < M) ? av.z : 0.f; av.w = (col + 3 < M) ? av.w : 0.f;
            hv.x = (col < H) ? hv.x : 0.f; hv.y = (col + 1 < H) ? hv.y : 0.f;
            hv.z = (col + 2 < H) ? hv.z : 0.f; hv.w = (col + 3 < H) ? hv.w : 0.f;
            *(v4f*)&A1[row * GR_ALD + col] = av;
            *(v4f*)&A2[row * GR_ALD + col] = hv;
            if (c < R && !p.agg_ready && col < p.ldagg)       // saved for the W_ih weight gradient
                *(v4f*)(p.agg + (long long)c * p.ldagg + col) = av;
            if (c4 == 0) edge_s[row] = hi[i] > lo[i];
        }
        // feature tail (and padding) of the new state: plain copy
        const int tail = p.ldhx - H;
        for (int e = tid; e < GR_ROWS * tail; e += 256) {
            const int row = e / tail, c = r0 + row, j = H + (e - row * tail);
            if (c < R) p.hx_new[(long long)c * p.ldhx + j] = p.hx_prev[(long long)c * p.ldhx + j];
        }
    }
    __syncthreads();
    const long long t_pro = p.trace ? (long long)wall_clock64() : 0;

    // ---- main loop -------------------------------------------------------------------------------
    f32x16 acc_r, acc_z, acc_in, acc_hn;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_r[r] = 0.f; acc_z[r] = 0.f; acc_in[r] = 0.f; acc_hn[r] = 0.f; }
    auto frags = [&](int buf, const float* A, int kk, int k8, float (&af)[4], float (&br)[4],
                     float (&bz)[4], float (&bn)[4]) {
        const v4f a = *(const v4f*)&A[l31 * GR_ALD + kk + k8 * 8 + 4 * lhi];
        af[0] = a.x; af[1] = a.y; af[2] = a.z; af[3] = a.w;
        const float* b = Bs + buf * GR_TILE;
        const int c = 2 * k8 + lhi, row = wid * 32 + l31;                  // chunk, row inside a gate
        const int rz = row + H, rn = row + 2 * H;
        const v4f vr = *(const v4f*)&b[row * GR_KT + 4 * (c ^ ((row >> 2) & 3))];
        const v4f vz = *(const v4f*)&b[rz * GR_KT + 4 * (c ^ ((rz >> 2) & 3))];
        const v4f vn = *(const v4f*)&b[rn * GR_KT + 4 * (c ^ ((rn >> 2) & 3))];
        br[0] = vr.x; br[1] = vr.y; br[2] = vr.z; br[3] = vr.w;
        bz[0] = vz.x; bz[1] = vz.y; bz[2] = vz.z; bz[3] = vz.w;
        bn[0] = vn.x; bn[1] = vn.y; bn[2] = vn.z; bn[3] = vn.w;
    };
    float af0[4], br0[4], bz0[4], bn0[4], af1[4], br1[4], bz1[4], bn1[4];
#define GI_GRU_MMA(AF, BR, BZ, BN, ACCN)                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[j], BR[j], acc_r, 0, 0, 0);                \
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[j], BZ[j], acc_z, 0, 0, 0);                \
        ACCN = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[j], BN[j], ACCN, 0, 0, 0);                  \
    }
#define GI_GRU_STEP(SLOT, A_, KK_, ACCN)                                                          \
    {                                                                                              \
        frags(SLOT, A_, KK_, 0, af0, br0, bz0, bn0);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        GI_GRU_MMA(af0, br0, bz0, bn0, ACCN)                                                       \
        frags(SLOT, A_, KK_, 1, af1, br1, bz1, bn1);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        GI_GRU_MMA(af1, br1, bz1, bn1, ACCN)                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    // Step t: this wave's pieces of tile t have landed (loads complete in issue order and the six
    // youngest are tile t+1's), barrier (the whole tile is there, every wave is done with tile t-1,
    // whose ring slot tile t+2 goes to), start the DMA of tile t+2, multiply.  Tiles 0 .. nk1-1
    // multiply the aggregated messages by W_ih, the rest the hidden state by W_hh.
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        dma_tile(t + 2);
        const bool ih = t < nk1;
        const float* A = ih ? A1 : A2;
        const int kk = (ih ? t : t - nk1) * GR_KT;
        const int slot = t % GR_RING;
        if (ih) GI_GRU_STEP(slot, A, kk, acc_in) else GI_GRU_STEP(slot, A, kk, acc_hn)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
#undef GI_GRU_STEP
#undef GI_GRU_MMA

    const long long t_loop = p.trace ? (long long)wall_clock64() : 0;
    // ---- epilogue: gates; C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
    const int j = wid * 32 + l31;
    const bool j_ok = j < H;
    const int jc = j_ok ? j : H - 1;
    const float b_r = p.b_ih[jc] + p.b_hh[jc], b_z = p.b_ih[H + jc] + p.b_hh[H + jc];
    const float b_in = p.b_ih[2 * H + jc], b_hn = p.b_hh[2 * H + jc];
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.gi + (long long)r0 * p.ldg), 0, nrows * p.ldg * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.gh + (long long)r0 * p.ldg), 0, nrows * p.ldg * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.hx_new + (long long)r0 * p.ldhx), 0, nrows * p.ldhx * 4, 0x00020000);
    const int drop = 0x40000000;                             // beyond any tile: the store is dropped
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const float hp = A2[row * GR_ALD + j];
        const bool upd = edge_s[row] != 0;
        const float rr = gi_sigmoid(acc_r[r] + b_r);
        const float zz = gi_sigmoid(acc_z[r] + b_z);
        const float hn = acc_hn[r] + b_hn;
        const float nn = tanhf((acc_in[r] + b_in) + rr * hn);
        const float hnew = upd ? (1.f - zz) * nn + zz * hp : hp;
        const int og = (j_ok & upd) ? (row * p.ldg + j) * 4 : drop;      // saved for backward
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rr), rg, og, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, zz), rg, og + 4 * H, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nn), rg, og + 8 * H, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn), rh, og + 8 * H, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hnew), rx,
                                              j_ok ? (row * p.ldhx + j) * 4 : drop, 0, 0);
    }
    if (p.trace && tid == 0) {              // measurement aid (GI_GRU_TRACE): 100 MHz wall clock
        long long* t = p.trace + 4 * (long long)blockIdx.x;
        t[0] = t_start; t[1] = t_pro; t[2] = t_loop; t[3] = (long long)wall_clock64();
    }
}

}  // namespace

extern "C" long long gi_gru_image_floats(int H, int M) {
    if (H < 4 || M < 4 || H > GI_GRU_MAXW || M > GI_GRU_MAXW) return GI_ELIMIT;
    return (long long)(gi_cdiv(M, GR_KT) + gi_cdiv(H, GR_KT)) * GR_TILE;
}

extern "C" int gi_gru_pack(const gi_gru_params* pp, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (!pp) return GI_EINVAL;
    const gi_gru_params& p = *pp;
    if (p.H < 4 || p.M < 4 || p.H > GI_GRU_MAXW || p.M > GI_GRU_MAXW) return GI_ELIMIT;
    if (!p.W_ih || !p.W_hh || !p.image || ((uintptr_t)p.image & 15)) return GI_EINVAL;
    const long long n4 = gi_gru_image_floats(p.H, p.M) / 4;
    hipLaunchKernelGGL(gi_gru_pack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, p);
    return gi_launch_status();
}

extern "C" int gi_gru_fused_fwd(const gi_gru_params* pp, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (!pp) return GI_EINVAL;
    const gi_gru_params& p = *pp;
    if (p.R <= 0) return 0;
    if (p.H < 4 || p.M < 4 || p.H > GI_GRU_MAXW || p.M > GI_GRU_MAXW) return GI_ELIMIT;
    if (!p.seg_off || !p.agg || !p.hx_prev || !p.hx_new || !p.W_ih || !p.W_hh || !p.b_ih || !p.b_hh ||
        !p.gi || !p.gh || !p.image || ((uintptr_t)p.image & 15))
        return GI_EINVAL;
    if (!p.agg_ready && (!p.m || !p.in_perm || (p.ldm & 3) || p.ldm < p.M || ((uintptr_t)p.m & 15)))
        return GI_EINVAL;
    if (p.ldg < 3 * p.H || p.ldhx < ((p.H + 3) & ~3) || p.ldagg < ((p.M + 3) & ~3) || (p.ldagg & 3) ||
        ((uintptr_t)p.agg & 15))
        return GI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    gi_gru_params q = p;
    q.trace = getenv("GI_GRU_TRACE") ? (long long*)strtoull(getenv("GI_GRU_TRACE"), nullptr, 0) : nullptr;
    // useful flops of the two projections (the aggregation is bandwidth work)
    GiProfScope prof(st, GI_PROF_GEMM, 2.0 * p.R * 3.0 * p.H * ((double)p.M + p.H));
    hipLaunchKernelGGL(gi_gru_fused_kernel, dim3(gi_cdiv(p.R, GR_ROWS)), dim3(256), 0, st, q);
    return gi_launch_status();
}
