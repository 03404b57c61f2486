// Graph-structured and pointwise kernels of the GGNN hot path (gfx950).  All HBM/L2-bound
// streaming kernels: coalesced row reads, 16-byte vectors where the layout allows, no atomics,
// deterministic summation order.
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include <string.h>

#include "gi_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

// ---- K4 seg_sum -----------------------------------------------------------------------------
// one thread per (compact row, 16-byte column group); lanes of a row group share perm[k] (broadcast
// load) and read one contiguous row of `vals` -> fully coalesced row gathers.
// A molecular graph's node has ~1.8 incoming edges: a thread's work is a chain of three DEPENDENT loads (offsets ->
// edge index -> message row) and one store, so the rate is bytes in flight over latency.  With one output per thread
// 2 048 resident threads x 16 B = 32 KB per CU were in flight: 4.0 TB/s = 0.50 of the HBM peak beyond the Infinity
// Cache (round 2-4).  Each thread now carries U independent outputs (rows c, c + stride, ...) through the three load
// stages together: U x the bytes in flight — 3.85 (U = 1), 4.43 (2), 4.58 (4), 3.3 TB/s (8: registers cost occupancy)
// on the 567 MB probe of bench.py (profiles/r05/segsum_probe.txt): U = 4, 0.57 of 8 TB/s.  The order in which ONE
// output's rows are added is unchanged (first pair, further pairs, tail), so results are bit-identical for every U
// (GI_SEGSUM_U = 1 / 2 / 4: measurement aid).
template <int U>
__global__ __launch_bounds__(256) void seg_sum_kernel(
    const float* __restrict__ vals, int ldv, const int* __restrict__ perm,
    const int* __restrict__ off, int rows, int c4n, float* out, int ldo, int accumulate,
    const int* __restrict__ rows_dev) {
    if (rows_dev) rows = min(rows, *rows_dev);          // bounded launch: the real row count is on the device
    const long long stride = (long long)gridDim.x * 256;
    const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x;
    int c[U], q[U], lo[U], hi[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long t = t0 + u * stride;
        c[u] = (int)(t / c4n); q[u] = (int)(t - (long long)c[u] * c4n);
        ok[u] = c[u] < rows;
        lo[u] = ok[u] ? off[c[u]] : 0;
        hi[u] = ok[u] ? off[c[u] + 1] : 0;
    }
    int p0[U], p1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                        // the first pair of every output: all loads in flight together
        p0[u] = lo[u] < hi[u] ? perm[lo[u]] : -1;
        p1[u] = lo[u] + 1 < hi[u] ? perm[lo[u] + 1] : -1;
    }
    v4f a[U], b[U], acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        a[u] = p0[u] >= 0 ? __builtin_nontemporal_load((const v4f*)(vals + (long long)p0[u] * ldv + 4 * q[u])) : v4f{0.f, 0.f, 0.f, 0.f};
        b[u] = p1[u] >= 0 ? __builtin_nontemporal_load((const v4f*)(vals + (long long)p1[u] * ldv + 4 * q[u])) : v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        acc[u] = v4f{0.f, 0.f, 0.f, 0.f};
        if (p0[u] >= 0) acc[u] += a[u];
        if (p1[u] >= 0) acc[u] += b[u];
        int k = lo[u] + 2;
        for (; k + 1 < hi[u]; k += 2) {                  // longer segments: two independent row loads in flight
            const int r0 = perm[k], r1 = perm[k + 1];
            const v4f x = *(const v4f*)(vals + (long long)r0 * ldv + 4 * q[u]);
            const v4f y = *(const v4f*)(vals + (long long)r1 * ldv + 4 * q[u]);
            acc[u] += x;
            acc[u] += y;
        }
        if (k < hi[u]) acc[u] += *(const v4f*)(vals + (long long)perm[k] * ldv + 4 * q[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        v4f* dst = (v4f*)(out + (long long)c[u] * ldo + 4 * q[u]);
        if (accumulate) acc[u] += *dst;
        *dst = acc[u];           // (a non-temporal store measured no gain: 4.44 against 4.49 TB/s)
    }
}

__device__ __forceinline__ v4f v4_selu_grad(v4f y) {
    return v4f{gi_selu_grad(y.x), gi_selu_grad(y.y), gi_selu_grad(y.z), gi_selu_grad(y.w)};
}
// d(layer output)/d(pre-activation) of the element stored at y: selu'(through its output) normally;
// in AlphaDropout training mode (fshift != 0) the factor keep * a * selu' that the forward stored
// `fshift` floats behind the activation (gi_alpha_dropout_fwd).
__device__ __forceinline__ float gi_dact(const float* y, long long fshift) {
    return fshift ? y[fshift] : gi_selu_grad(*y);
}
__device__ __forceinline__ v4f v4_dact(const float* y, long long fshift) {
    return fshift ? *(const v4f*)(y + fshift) : v4_selu_grad(*(const v4f*)y);
}

// seg_sum over the message CSR with the SELU backward of the destination buffer fused in
__global__ __launch_bounds__(256) void seg_sum_dselu_kernel(
    const float* __restrict__ vals, int ldv, const int* __restrict__ perm,
    const int* __restrict__ off, int rows, int c4n, float* y, int ldy, long long fshift) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(t / c4n), q = (int)(t - (long long)c * c4n);
    if (c >= rows) return;
    const int lo = off[c], hi = off[c + 1];
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    int k = lo;
    for (; k + 1 < hi; k += 2) {
        const int p0 = perm[k], p1 = perm[k + 1];
        const v4f a = *(const v4f*)(vals + (long long)p0 * ldv + 4 * q);
        const v4f b = *(const v4f*)(vals + (long long)p1 * ldv + 4 * q);
        acc += a;
        acc += b;
    }
    if (k < hi) acc += *(const v4f*)(vals + (long long)perm[k] * ldv + 4 * q);
    float* dst = y + (long long)c * ldy + 4 * q;
    *(v4f*)dst = acc * v4_dact(dst, fshift);
}

// y[r, c] = selu'(y[r, c]) * sum_s slabs[s * stride + r * ld + c]   (pass-0 shortcut, tiny).
// 16 lanes share one output: lane q sums splits q, q+16, ... , then a fixed shuffle tree.
__global__ __launch_bounds__(256) void slab_sum_dselu_kernel(const float* __restrict__ slabs,
                                                             int nsplit, long long stride, int rows,
                                                             int cols, int ld, float* y, int ldy) {
    const int t = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15;
    const int r = t / cols, c = t - r * cols;
    float acc = 0.f;
    if (r < rows)
        for (int s = q; s < nsplit; s += 16) acc += slabs[s * stride + (long long)r * ld + c];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_down(acc, o, 16);
    if (r < rows && q == 0) {
        float* dst = y + (long long)r * ldy + c;
        *dst = acc * gi_selu_grad(*dst);
    }
}

// out[r, c] = epilogue(sum_s slabs[s * stride + r * ld + c]): finishes a split-K GEMM whose epilogue could
// not run per slab (bias + SELU of a forward layer; the SELU-backward factor of a dgrad).  Splits are
// summed in index order, two accumulators (even / odd), like gi_reduce_slabs.
__global__ __launch_bounds__(256) void slab_epilogue_kernel(
    const float* __restrict__ slabs, int nsplit, long long stride, int rows, int cols, int ld,
    int flags, const float* __restrict__ bias, const float* __restrict__ act, int ldact,
    float* __restrict__ out, int ldo) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(t / cols), c = (int)(t - (long long)r * cols);
    if (r >= rows) return;
    const float* src = slabs + (long long)r * ld + c;
    float s0 = 0.f, s1 = 0.f;
    int j = 0;
    for (; j + 1 < nsplit; j += 2) {
        s0 += src[(long long)j * stride];
        s1 += src[(long long)(j + 1) * stride];
    }
    if (j < nsplit) s0 += src[(long long)j * stride];
    float x = s0 + s1;
    if (flags & GI_EPI_BIAS) x += bias[c];
    if (flags & GI_EPI_SELU) x = gi_selu(x);
    if (flags & GI_EPI_DSELU) x *= gi_selu_grad(act[(long long)r * ldact + c]);
    if (flags & GI_EPI_MULACT) x *= act[(long long)r * ldact + c];
    float* dst = out + (long long)r * ldo + c;
    if (flags & GI_EPI_ACCUM) x += *dst;
    *dst = x;
}

// ---- AttGGNN attention aggregation (gnn/mpnn.py:370-389) ------------------------------------------
// One thread per (destination row, 16-byte feature group).  The reference pads every node's
// neighbour list to the batch's maximum degree and masks with -1e6; here the softmax runs over the
// node's own CSR segment (padding slots contribute exp(-1e6 - max) == 0 in fp32 there too).
// Three short passes over the segment (<= max degree rows): max, sum of exp, weighted sum.
__device__ __forceinline__ v4f v4_max(v4f a, v4f b) {
    return v4f{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
}
__device__ __forceinline__ v4f v4_exp(v4f a) {
    return v4f{expf(a.x), expf(a.y), expf(a.z), expf(a.w)};
}
__device__ __forceinline__ v4f v4_rcp(v4f a) {
    return v4f{1.f / a.x, 1.f / a.y, 1.f / a.z, 1.f / a.w};
}

__global__ __launch_bounds__(256) void seg_softmax_fwd_kernel(
    const float* __restrict__ en, const float* __restrict__ emb, int ld,
    const int* __restrict__ perm, const int* __restrict__ off, int rows, int c4n,
    float* __restrict__ out, int ldo, const int* __restrict__ rows_dev) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(t / c4n), q = (int)(t - (long long)c * c4n);
    if (rows_dev) rows = min(rows, *rows_dev);
    if (c >= rows) return;
    const int lo = off[c], hi = off[c + 1];
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    if (hi > lo) {
        v4f mx = *(const v4f*)(en + (long long)perm[lo] * ld + 4 * q);
        for (int k = lo + 1; k < hi; ++k)
            mx = v4_max(mx, *(const v4f*)(en + (long long)perm[k] * ld + 4 * q));
        v4f den = {0.f, 0.f, 0.f, 0.f};
        for (int k = lo; k < hi; ++k) {
            const long long r = (long long)perm[k] * ld + 4 * q;
            const v4f e = v4_exp(*(const v4f*)(en + r) - mx);
            den += e;
            acc += e * *(const v4f*)(emb + r);
        }
        acc = acc * v4_rcp(den);
    }
    *(v4f*)(out + (long long)c * ldo + 4 * q) = acc;
}

// Backward of the attention aggregation: per-edge contributions in dst-CSR slot order (a message
// row may feed several destinations; its gradient is the seg_sum_dselu of these over the message CSR).
__global__ __launch_bounds__(256) void seg_softmax_bwd_kernel(
    const float* __restrict__ en, const float* __restrict__ emb, int ld,
    const int* __restrict__ perm, const int* __restrict__ off, int rows, int c4n,
    const float* __restrict__ dagg, int ldd, float* __restrict__ d_en_e,
    float* __restrict__ d_emb_e, int lde) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(t / c4n), q = (int)(t - (long long)c * c4n);
    if (c >= rows) return;
    const int lo = off[c], hi = off[c + 1];
    if (hi <= lo) return;
    const v4f d = *(const v4f*)(dagg + (long long)c * ldd + 4 * q);
    v4f mx = *(const v4f*)(en + (long long)perm[lo] * ld + 4 * q);
    for (int k = lo + 1; k < hi; ++k)
        mx = v4_max(mx, *(const v4f*)(en + (long long)perm[k] * ld + 4 * q));
    v4f den = {0.f, 0.f, 0.f, 0.f}, inner = {0.f, 0.f, 0.f, 0.f};
    for (int k = lo; k < hi; ++k) {
        const long long r = (long long)perm[k] * ld + 4 * q;
        const v4f e = v4_exp(*(const v4f*)(en + r) - mx);
        den += e;
        inner += e * *(const v4f*)(emb + r) * d;          // sum_k exp_k * d att_k
    }
    const v4f inv = v4_rcp(den);
    inner = inner * inv;                                  // sum_k att_k * d att_k
    for (int k = lo; k < hi; ++k) {
        const long long r = (long long)perm[k] * ld + 4 * q;
        const v4f att = v4_exp(*(const v4f*)(en + r) - mx) * inv;
        *(v4f*)(d_en_e + (long long)k * lde + 4 * q) = att * (*(const v4f*)(emb + r) * d - inner);
        *(v4f*)(d_emb_e + (long long)k * lde + 4 * q) = att * d;
    }
}

__global__ __launch_bounds__(256) void selu_bwd_rows_kernel(
    const float* __restrict__ dY, int lddy, const int* __restrict__ idx, const float* Y, int ldy,
    float* out, int ldo, int rows, int cols, long long fshift) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(t / cols), c = (int)(t - (long long)r * cols);
    if (r >= rows) return;
    const long long sr = idx ? idx[r] : r;
    out[(long long)r * ldo + c] = dY[sr * lddy + c] * gi_dact(Y + (long long)r * ldy + c, fshift);
}

// The three column ranges of the logits (fAddNet2 | fConnNet2 | fTermNet2 outputs, gnn/modules.py:265-279)
// in one launch: out_k[r, c - start_k] = dY[r, c] * dact(Y[r, c]) for c in range k.
__global__ __launch_bounds__(256) void selu_bwd_cols3_kernel(
    const float* __restrict__ dY, int lddy, const float* Y, int ldy, long long fshift, int rows, int n0,
    int n1, int n2, float* __restrict__ o0, int ld0, float* __restrict__ o1, int ld1,
    float* __restrict__ o2, int ld2) {
    const int W = n0 + n1 + n2;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(t / W), c = (int)(t - (long long)r * W);
    if (r >= rows) return;
    const float x = dY[(long long)r * lddy + c] * gi_dact(Y + (long long)r * ldy + c, fshift);
    if (c < n0) o0[(long long)r * ld0 + c] = x;
    else if (c < n0 + n1) o1[(long long)r * ld1 + (c - n0)] = x;
    else o2[(long long)r * ld2 + (c - n0 - n1)] = x;
}

// ---- AlphaDropout (training mode of gnn/modules.py:130-142, p > 0) ------------------------------
// torch.nn.AlphaDropout after every Linear+SELU: y = s * (keep * a) + b with b = (keep - 1) * alpha' * a
// + alpha' * a * p, evaluated exactly like ATen's _dropout_impl (two roundings: fl(fl(s * a) + b));
// the keep bit is a counter-based hash of (seed, layer id, row, column) — nothing is stored but the
// factor d y / d z = keep * a * selu'(z), written `fshift` floats behind the activation for the
// backward (gi_dact above, GI_EPI_MULACT in gi_gemm).
__device__ __forceinline__ bool gi_dropout_keep(unsigned long long seed, unsigned id, unsigned thresh,
                                                int row, int col) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)id + 1ull);
    z ^= ((unsigned long long)(unsigned)row << 32) | (unsigned)col;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;          // splitmix64 finaliser
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32) >= thresh;
}

__global__ __launch_bounds__(256) void alpha_dropout_fwd_kernel(float* y, int ldy, int rows, int cols,
                                                                long long fshift,
                                                                const gi_dropout_params q) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(t / cols), c = (int)(t - (long long)r * cols);
    if (r >= rows) return;
    float* p = y + (long long)r * ldy + c;
    const float s = *p;
    const bool keep = gi_dropout_keep(q.seed, q.id, q.thresh, r, c);
    float sa = s * q.a;                      // ATen: mul, then add — two roundings, never an fma
    asm volatile("" : "+v"(sa));             // (keeps hipcc from contracting the pair)
    *p = keep ? sa + q.b_keep : q.b_drop;
    p[fshift] = keep ? q.a * gi_selu_grad(s) : 0.f;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(const gi_dropout_params q, int rows,
                                                           int cols, unsigned char* keep, int ld) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(t / cols), c = (int)(t - (long long)r * cols);
    if (r >= rows) return;
    keep[(long long)r * ld + c] = gi_dropout_keep(q.seed, q.id, q.thresh, r, c) ? 1 : 0;
}

// ---- GRU gates --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(
    float* gi, const float* __restrict__ gh, int ldg, const float* __restrict__ hx_prev,
    float* __restrict__ hx_new, int ldh, const int* __restrict__ seg_off, int rows, int H,
    const int* __restrict__ rows_dev) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(t / ldh), j = (int)(t - (long long)row * ldh);
    if (rows_dev) rows = min(rows, *rows_dev);
    if (row >= rows) return;
    const float hp = hx_prev[(long long)row * ldh + j];
    float hn_out = hp;                                  // feature tail / padding: plain copy
    if (j < H && seg_off[row + 1] > seg_off[row]) {
        float* g = gi + (long long)row * ldg;
        const float* h = gh + (long long)row * ldg;
        const float r = gi_sigmoid(g[j] + h[j]);
        const float z = gi_sigmoid(g[H + j] + h[H + j]);
        const float n = tanhf(g[2 * H + j] + r * h[2 * H + j]);
        hn_out = (1.f - z) * n + z * hp;
        g[j] = r; g[H + j] = z; g[2 * H + j] = n;       // saved for backward
    }
    hx_new[(long long)row * ldh + j] = hn_out;
}

__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(
    float* gi, float* gh, int ldg, const float* __restrict__ hx_prev, int ldh,
    const float* __restrict__ dh_new, const float* __restrict__ dh_b, const float* __restrict__ dh_c,
    const float* __restrict__ dh_d, float* __restrict__ dh_prev, int lddh,
    const int* __restrict__ seg_off, int rows, int H) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(t / H), j = (int)(t - (long long)row * H);
    if (row >= rows) return;
    float* g = gi + (long long)row * ldg;
    float* h = gh + (long long)row * ldg;
    float d = dh_new[(long long)row * lddh + j];             // + the sibling stacks' contributions
    if (dh_b) d += dh_b[(long long)row * lddh + j];
    if (dh_c) d += dh_c[(long long)row * lddh + j];
    if (dh_d) d += dh_d[(long long)row * lddh + j];
    if (seg_off[row + 1] > seg_off[row]) {
        const float r = g[j], z = g[H + j], n = g[2 * H + j], hn = h[2 * H + j];
        const float hp = hx_prev[(long long)row * ldh + j];
        const float dn = d * (1.f - z);
        const float dz = d * (hp - n);
        const float dpn = dn * (1.f - n * n);
        const float dpr = dpn * hn * r * (1.f - r);
        const float dpz = dz * z * (1.f - z);
        g[j] = dpr; g[H + j] = dpz; g[2 * H + j] = dpn;
        h[j] = dpr; h[H + j] = dpz; h[2 * H + j] = dpn * r;
        dh_prev[(long long)row * lddh + j] = d * z;
    } else {
        g[j] = 0.f; g[H + j] = 0.f; g[2 * H + j] = 0.f;
        h[j] = 0.f; h[H + j] = 0.f; h[2 * H + j] = 0.f;
        dh_prev[(long long)row * lddh + j] = d;
    }
}

// Four hidden units per thread (16-byte accesses) — same arithmetic per element as the scalar kernels
// above, which stay the fallback for H % 4 != 0 (the three gate blocks of a row then do not start on
// 16-byte boundaries).
__device__ __forceinline__ v4f v4_sigmoid(v4f a) {
    return v4f{gi_sigmoid(a.x), gi_sigmoid(a.y), gi_sigmoid(a.z), gi_sigmoid(a.w)};
}
__device__ __forceinline__ v4f v4_tanh(v4f a) {
    return v4f{tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w)};
}

__global__ __launch_bounds__(256) void gru_gates_fwd_v4_kernel(
    float* gi, const float* __restrict__ gh, int ldg, const float* __restrict__ hx_prev,
    float* __restrict__ hx_new, int ldh, const int* __restrict__ seg_off, int rows, int H,
    const int* __restrict__ rows_dev) {
    const int q4 = ldh >> 2;                             // ldh is a multiple of 4
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(t / q4), j = 4 * (int)(t - (long long)row * q4);
    if (rows_dev) rows = min(rows, *rows_dev);
    if (row >= rows) return;
    const v4f hp = *(const v4f*)(hx_prev + (long long)row * ldh + j);
    v4f hn_out = hp;                                    // feature tail / padding: plain copy
    if (j < H && seg_off[row + 1] > seg_off[row]) {     // H % 4 == 0: a group never straddles H
        float* g = gi + (long long)row * ldg;
        const float* h = gh + (long long)row * ldg;
        const v4f r = v4_sigmoid(*(const v4f*)(g + j) + *(const v4f*)(h + j));
        const v4f z = v4_sigmoid(*(const v4f*)(g + H + j) + *(const v4f*)(h + H + j));
        const v4f n = v4_tanh(*(const v4f*)(g + 2 * H + j) + r * *(const v4f*)(h + 2 * H + j));
        hn_out = (1.f - z) * n + z * hp;
        *(v4f*)(g + j) = r; *(v4f*)(g + H + j) = z; *(v4f*)(g + 2 * H + j) = n;
    }
    *(v4f*)(hx_new + (long long)row * ldh + j) = hn_out;
}

// The backward, with — optionally — the scatter of the message stacks' input gradients back to their
// source nodes folded into the load of d h: otherwise its own launch in front of this one,
// gi_seg_sum(sc, sc_perm, sc_off, ..., dh_new, accumulate) over the source CSR (one per stack;
// AttentionGGNN has two).  The sums are formed exactly like seg_sum_kernel forms them (pairs, tail,
// then + the destination), so the fused and the two-launch paths agree bit for bit.
__global__ __launch_bounds__(256) void gru_gates_bwd_v4_kernel(
    float* gi, float* gh, int ldg, const float* __restrict__ hx_prev, int ldh,
    const float* __restrict__ dh_new, const float* __restrict__ dh_b, const float* __restrict__ dh_c,
    const float* __restrict__ dh_d, float* __restrict__ dh_prev, int lddh,
    const int* __restrict__ seg_off, int rows, int H, const float* __restrict__ sc0,
    const float* __restrict__ sc1, int ldsc, const int* __restrict__ sc_perm,
    const int* __restrict__ sc_off) {
    const int q4 = H >> 2;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(t / q4), j = 4 * (int)(t - (long long)row * q4);
    if (row >= rows) return;
    float* g = gi + (long long)row * ldg;
    float* h = gh + (long long)row * ldg;
    v4f d = *(const v4f*)(dh_new + (long long)row * lddh + j);
    if (sc0) {
        const int lo = sc_off[row], hi = sc_off[row + 1];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float* sc = s ? sc1 : sc0;
            if (!sc) break;
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            int k = lo;
            for (; k + 1 < hi; k += 2) {
                const int p0 = sc_perm[k], p1 = sc_perm[k + 1];
                const v4f a = *(const v4f*)(sc + (long long)p0 * ldsc + j);
                const v4f b = *(const v4f*)(sc + (long long)p1 * ldsc + j);
                acc += a;
                acc += b;
            }
            if (k < hi) acc += *(const v4f*)(sc + (long long)sc_perm[k] * ldsc + j);
            d = acc + d;
        }
    }
    if (dh_b) d += *(const v4f*)(dh_b + (long long)row * lddh + j);   // + the sibling stacks' contributions
    if (dh_c) d += *(const v4f*)(dh_c + (long long)row * lddh + j);
    if (dh_d) d += *(const v4f*)(dh_d + (long long)row * lddh + j);
    if (seg_off[row + 1] > seg_off[row]) {
        const v4f r = *(const v4f*)(g + j), z = *(const v4f*)(g + H + j), n = *(const v4f*)(g + 2 * H + j);
        const v4f hn = *(const v4f*)(h + 2 * H + j);
        const v4f hp = *(const v4f*)(hx_prev + (long long)row * ldh + j);
        const v4f dn = d * (1.f - z);
        const v4f dz = d * (hp - n);
        const v4f dpn = dn * (1.f - n * n);
        const v4f dpr = dpn * hn * r * (1.f - r);
        const v4f dpz = dz * z * (1.f - z);
        *(v4f*)(g + j) = dpr; *(v4f*)(g + H + j) = dpz; *(v4f*)(g + 2 * H + j) = dpn;
        *(v4f*)(h + j) = dpr; *(v4f*)(h + H + j) = dpz; *(v4f*)(h + 2 * H + j) = dpn * r;
        *(v4f*)(dh_prev + (long long)row * lddh + j) = d * z;
    } else {
        const v4f zero = {0.f, 0.f, 0.f, 0.f};
        *(v4f*)(g + j) = zero; *(v4f*)(g + H + j) = zero; *(v4f*)(g + 2 * H + j) = zero;
        *(v4f*)(h + j) = zero; *(v4f*)(h + H + j) = zero; *(v4f*)(h + 2 * H + j) = zero;
        *(v4f*)(dh_prev + (long long)row * lddh + j) = d;
    }
}

// ---- K7 gather readout ------------------------------------------------------------------------
// One workgroup per (graph, chunk of GI_GATHER_GC feature columns); the graph's energy / embedding rows of
// the chunk are first staged in LDS by all threads (coalesced row gathers, many loads in flight) —
// the three passes over the node axis then run out of LDS, one thread per column, in the same node
// order as before (a thread used to walk the N rows three times through global memory with one load
// in flight: 38 us per launch at N = 88, latency bound).  Dynamic LDS: 2 * N * GC floats (GC = 64
// columns per workgroup, 32 for N > 112: at most 56 KB).
template <int GI_GATHER_GC>
__global__ __launch_bounds__(GI_GATHER_GC) void gather_fwd_kernel(
    const float* __restrict__ en, const float* __restrict__ emb, int ld,
    const int* __restrict__ cidx, const int* __restrict__ mask, int N, int G, float big,
    float* out0, int ld0, float* out1, int ld1, float* out2, int ld2) {
    extern __shared__ float gather_lds[];
    __shared__ int c_s[GI_MAX_NODES];
    __shared__ float pen_s[GI_MAX_NODES];
    float* e_s = gather_lds;                               // [N][GC] energies minus the mask penalty
    float* m_s = gather_lds + N * GI_GATHER_GC;            // [N][GC] embeddings
    const int b = blockIdx.x, g0 = blockIdx.y * GI_GATHER_GC, tid = threadIdx.x;
    for (int n = tid; n < N; n += GI_GATHER_GC) {
        c_s[n] = cidx[b * N + n];
        pen_s[n] = mask[b * N + n] ? 0.f : big;       // (node_mask == 0) * big_positive, modules.py:46
    }
    __syncthreads();
    const int g = g0 + tid;
    if (g < G) {
#pragma unroll 4
        for (int n = 0; n < N; ++n) {
            const long long o = (long long)c_s[n] * ld + g;
            e_s[n * GI_GATHER_GC + tid] = en[o] - pen_s[n];
            m_s[n * GI_GATHER_GC + tid] = emb[o];
        }
        float m = -INFINITY;
        for (int n = 0; n < N; ++n) m = fmaxf(m, e_s[n * GI_GATHER_GC + tid]);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += expf(e_s[n * GI_GATHER_GC + tid] - m);
        float acc = 0.f;
        for (int n = 0; n < N; ++n)
            acc += (expf(e_s[n * GI_GATHER_GC + tid] - m) / s) * m_s[n * GI_GATHER_GC + tid];
        if (out0) out0[(long long)b * ld0 + g] = acc;
        if (out1) out1[(long long)b * ld1 + g] = acc;
        if (out2) out2[(long long)b * ld2 + g] = acc;
    }
}

template <int GI_GATHER_GC>
__global__ __launch_bounds__(GI_GATHER_GC) void gather_bwd_kernel(
    float* en, float* emb, int ld, const int* __restrict__ cidx, const int* __restrict__ mask,
    int N, int G, int S, float big, const float* __restrict__ dg0, int ld0,
    const float* __restrict__ dg1, int ld1, const float* __restrict__ dg2, int ld2,
    float* __restrict__ zpart, long long fshift) {
    extern __shared__ float gather_lds[];
    __shared__ int c_s[GI_MAX_NODES];
    __shared__ float pen_s[GI_MAX_NODES];
    float* e_s = gather_lds;                               // raw energies
    float* m_s = gather_lds + N * GI_GATHER_GC;
    const int b = blockIdx.x, g0 = blockIdx.y * GI_GATHER_GC, tid = threadIdx.x;
    for (int n = tid; n < N; n += GI_GATHER_GC) {
        c_s[n] = cidx[b * N + n];
        pen_s[n] = mask[b * N + n] ? 0.f : big;
    }
    __syncthreads();
    const int ldz = 2 * G;
    const int g = g0 + tid;
    if (g >= G) return;
    // each thread stages and later overwrites only its own column: the slots of a graph that share the
    // zero row (c >= S) read the same, never written, row
#pragma unroll 4
    for (int n = 0; n < N; ++n) {
        const long long o = (long long)c_s[n] * ld + g;
        e_s[n * GI_GATHER_GC + tid] = en[o];
        m_s[n * GI_GATHER_GC + tid] = emb[o];
    }
    float dg = 0.f;
    if (dg0) dg += dg0[(long long)b * ld0 + g];
    if (dg1) dg += dg1[(long long)b * ld1 + g];
    if (dg2) dg += dg2[(long long)b * ld2 + g];
    float m = -INFINITY;
    for (int n = 0; n < N; ++n) m = fmaxf(m, e_s[n * GI_GATHER_GC + tid] - pen_s[n]);
    float s = 0.f, sd = 0.f;                                 // sum p, sum p * (emb*dg)
    for (int n = 0; n < N; ++n) {
        const float p = expf((e_s[n * GI_GATHER_GC + tid] - pen_s[n]) - m);
        s += p;
        sd += p * (m_s[n * GI_GATHER_GC + tid] * dg);
    }
    const float dot = sd / s;                                // sum_n att_n * datt_n
    float zen = 0.f, zemb = 0.f;
    for (int n = 0; n < N; ++n) {
        const long long o = (long long)c_s[n] * ld + g;
        const float ev = e_s[n * GI_GATHER_GC + tid], mv = m_s[n * GI_GATHER_GC + tid];
        const float att = expf((ev - pen_s[n]) - m) / s;
        const float de = att * (mv * dg - dot);
        const float dm = att * dg;
        if (c_s[n] < S) {                                    // this slot owns its compact row
            en[o] = de * (fshift ? en[o + fshift] : gi_selu_grad(ev));
            emb[o] = dm * (fshift ? emb[o + fshift] : gi_selu_grad(mv));
        } else {                                             // shared zero row: per-graph partial
            zen += de;
            zemb += dm;
        }
    }
    zpart[(long long)b * ldz + g] = zen;
    zpart[(long long)b * ldz + G + g] = zemb;
}

// ---- tier-1 <-> tier-2 glue ------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_slots_kernel(
    const float* __restrict__ t1, int ldt, const int* __restrict__ cidx, int NW, int W,
    float* __restrict__ cat, int ldc, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int b = (int)(t / NW), r = (int)(t - (long long)b * NW);
    const int n = r / W, w = r - n * W;
    const int N = NW / W;
    cat[(long long)b * ldc + r] = t1[(long long)cidx[b * N + n] * ldt + w];
}

// both tier-1 outputs (fAddNet1 / fConnNet1) in one launch: elements [0, total_a) belong to problem a
struct ExpandPair { const float* t1[2]; int ldt[2]; int W[2]; float* cat[2]; int ldc[2]; };

__global__ __launch_bounds__(256) void expand_slots2_kernel(const ExpandPair a, const int* __restrict__ cidx,
                                                            int N, long long total_a, long long total) {
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int k = t >= total_a ? 1 : 0;
    t -= k ? total_a : 0;
    const int W = a.W[k], NW = N * W;
    const int b = (int)(t / NW), r = (int)(t - (long long)b * NW);
    const int n = r / W, w = r - n * W;
    a.cat[k][(long long)b * a.ldc[k] + r] = a.t1[k][(long long)cidx[b * N + n] * a.ldt[k] + w];
}

// One workgroup per (graph, quarter of its N*W elements): the slots that own a compact row are
// independent elementwise work (the old kernel walked the N slots of a column serially from 64 threads:
// 45 us per launch at N = 88, B = 250); the per-graph sums over the slots sharing the zero row stay a
// serial loop in slot order, done by the first workgroup of the graph.
__global__ __launch_bounds__(256) void compress_slots_kernel(
    float* t1, int ldt, const int* __restrict__ cidx, int N, int W, int S,
    const float* __restrict__ dcat, int ldc, float* __restrict__ zpart, int ldz, long long fshift) {
    __shared__ int c_s[GI_MAX_NODES];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int n = tid; n < N; n += 256) c_s[n] = cidx[b * N + n];
    __syncthreads();
    const int total = N * W, chunk = (total + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * chunk, hi = min(lo + chunk, total);
    for (int idx = lo + tid; idx < hi; idx += 256) {
        const int n = idx / W, w = idx - n * W;
        const int c = c_s[n];
        if (c < S) {
            float* p = t1 + (long long)c * ldt + w;
            *p = dcat[(long long)b * ldc + idx] * gi_dact(p, fshift);
        }
    }
    if (blockIdx.y == 0)
        for (int w = tid; w < W; w += 256) {
            float z = 0.f;
            for (int n = 0; n < N; ++n)
                if (c_s[n] >= S) z += dcat[(long long)b * ldc + n * W + w];
            zpart[(long long)b * ldz + w] = z;
        }
}

// the same for the two tier-1 stacks in one launch (blockIdx.z = problem)
struct CompressPair { float* t1[2]; int ldt[2]; int W[2]; const float* dcat[2]; int ldc[2]; float* zpart[2]; int ldz[2]; };

__global__ __launch_bounds__(256) void compress_slots2_kernel(const CompressPair a, const int* __restrict__ cidx,
                                                              int N, int S, long long fshift) {
    __shared__ int c_s[GI_MAX_NODES];
    const int k = blockIdx.z;
    float* t1 = a.t1[k];
    const float* __restrict__ dcat = a.dcat[k];
    float* __restrict__ zpart = a.zpart[k];
    const int ldt = a.ldt[k], W = a.W[k], ldc = a.ldc[k], ldz = a.ldz[k];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int n = tid; n < N; n += 256) c_s[n] = cidx[b * N + n];
    __syncthreads();
    const int total = N * W, chunk = (total + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * chunk, hi = min(lo + chunk, total);
    for (int idx = lo + tid; idx < hi; idx += 256) {
        const int n = idx / W, w = idx - n * W;
        const int c = c_s[n];
        if (c < S) {
            float* p = t1 + (long long)c * ldt + w;
            *p = dcat[(long long)b * ldc + idx] * gi_dact(p, fshift);
        }
    }
    if (blockIdx.y == 0)
        for (int w = tid; w < W; w += 256) {
            float z = 0.f;
            for (int n = 0; n < N; ++n)
                if (c_s[n] >= S) z += dcat[(long long)b * ldc + n * W + w];
            zpart[(long long)b * ldz + w] = z;
        }
}

// out[c] = (sum_r part[r, c]) * selu'(y[c]); 64 columns per block, 16 row groups, fixed tree.
// blockIdx.y selects one of up to 8 independent problems.
struct ColsumTable { gi_colsum_desc d[8]; };

__global__ __launch_bounds__(1024) void colsum_kernel(const ColsumTable tab) {
    __shared__ float red[16][64];
    const gi_colsum_desc& q = tab.d[blockIdx.y];
    const float* __restrict__ part = q.part;
    const int ldp = q.ldp, rows = q.rows, cols = q.cols;
    const float* y = q.y;
    float* out = q.out;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (blockIdx.x * 64 >= cols) return;                  // block-uniform
    float acc = 0.f;
    if (c < cols) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // four loads in flight; fixed order
        int r = rg;
        for (; r + 48 < rows; r += 64) {
            a0 += part[(long long)r * ldp + c];
            a1 += part[(long long)(r + 16) * ldp + c];
            a2 += part[(long long)(r + 32) * ldp + c];
            a3 += part[(long long)(r + 48) * ldp + c];
        }
        for (; r < rows; r += 16) a0 += part[(long long)r * ldp + c];
        acc = (a0 + a1) + (a2 + a3);
    }
    red[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][cl];
        if (y) s *= gi_selu_grad(y[c]);
        out[c] = s;
    }
}

// y[d, c] = selu'(y[d, c]) * sum_{k in [off[d], off[d+1])} vals[idx[k], c]   — long segments (hundreds of
// rows per output row): one workgroup per (output row, 64 columns), 16 row groups, fixed tree.
// blockIdx.z selects one of two problems (the message and the energy stack share one launch).
struct ClassSumArgs {
    const float* vals[2]; float* y[2];
    int ldv, ldy, cols;
    const int* idx; const int* off;
};

__global__ __launch_bounds__(1024) void class_sum_dselu_kernel(const ClassSumArgs a) {
    __shared__ float red[16][64];
    const float* __restrict__ vals = a.vals[blockIdx.z];
    float* y = a.y[blockIdx.z];
    const int d = blockIdx.x, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int lo = a.off[d], hi = a.off[d + 1];
    float a0 = 0.f, a1 = 0.f;
    if (c < a.cols) {
        int k = lo + rg;
        for (; k + 16 < hi; k += 32) {                       // two loads in flight; fixed order
            a0 += vals[(long long)a.idx[k] * a.ldv + c];
            a1 += vals[(long long)a.idx[k + 16] * a.ldv + c];
        }
        if (k < hi) a0 += vals[(long long)a.idx[k] * a.ldv + c];
    }
    red[rg][cl] = a0 + a1;
    __syncthreads();
    if (rg == 0 && c < a.cols) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][cl];
        float* dst = y + (long long)d * a.ldy + c;
        *dst = s * gi_selu_grad(*dst);
    }
}

// ---- wgrad slab reduction ----------------------------------------------------------------------
#define GI_REDUCE_MAX 40
struct ReduceTable { gi_reduce_desc d[GI_REDUCE_MAX]; int start[GI_REDUCE_MAX + 1]; int n; };

// one thread per output element (dW[n,k] or db[n]); blocks are dealt to descriptors through a
// prefix table so big weight matrices get proportionally many workgroups
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const ReduceTable tab) {
    int i = 0;
    const int id = blockIdx.x;
    while (i < tab.n - 1 && id >= tab.start[i + 1]) ++i;
    const gi_reduce_desc& d = tab.d[i];
    const int K1 = d.K + 1;
    const long long total = (long long)d.N * K1;
    const long long idx = (long long)(id - tab.start[i]) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n = (int)(idx / K1), k = (int)(idx - (long long)n * K1);
    const float* src = d.slabs + (long long)n * d.ld + k;
    float s0 = 0.f, s1 = 0.f;
    int j = 0;
    for (; j + 1 < d.n_slabs; j += 2) {                 // two independent loads in flight
        s0 += src[(long long)j * d.slab_stride];
        s1 += src[(long long)(j + 1) * d.slab_stride];
    }
    if (j < d.n_slabs) s0 += src[(long long)j * d.slab_stride];
    const float sum = s0 + s1;
    if (k < d.K) d.dW[(long long)n * d.K + k] = sum;
    else if (d.db) d.db[n] = sum;
}

// ---- Adam (torch.optim.Adam semantics, one flat fp32 bucket) ------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   long long n4, float step_size, float b1,
                                                   float omb1, float b2, float omb2, float eps,
                                                   float wd, float bc2_sqrt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4;
         i += (long long)gridDim.x * 256) {
        v4f pp = ((v4f*)p)[i], gg = ((const v4f*)g)[i], mm = ((v4f*)m)[i], vv = ((v4f*)v)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gr = gg[c] + wd * pp[c];
            mm[c] = mm[c] + omb1 * (gr - mm[c]);              // lerp_(grad, 1 - beta1)
            vv[c] = b2 * vv[c] + omb2 * gr * gr;              // mul_(beta2).addcmul_(g, g, 1 - beta2)
            const float denom = sqrtf(vv[c]) / bc2_sqrt + eps;
            pp[c] -= step_size * (mm[c] / denom);
        }
        ((v4f*)p)[i] = pp; ((v4f*)m)[i] = mm; ((v4f*)v)[i] = vv;
    }
}

// ---- KL-divergence training loss (Workflow.py:850-858), forward + gradient in one pass ----------
// one workgroup per graph: t = target / sum(target); logp = log_softmax(out);
// row_loss = sum_j xlogy(t_j, t_j) - t_j * logp_j;  d_out_j = (softmax_j * sum(t) - t_j) / B
template <typename T, int NT>
__global__ __launch_bounds__(NT) void kl_loss_kernel(const float* __restrict__ out, int ldo,
                                                      const T* __restrict__ tgt, int ldt,
                                                      int width, float inv_b,
                                                      float* __restrict__ row_loss,
                                                      float* __restrict__ d_out, int ldd) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* o = out + (long long)b * ldo;
    const T* t = tgt + (long long)b * ldt;
    auto block_reduce = [&](float x, bool is_max) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const float y = __shfl_xor(x, s);
            x = is_max ? fmaxf(x, y) : x + y;
        }
        __syncthreads();
        if (lane == 0) red[wid] = x;
        __syncthreads();
        if (NT == 256)
            return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))
                          : (red[0] + red[1]) + (red[2] + red[3]);
        float r = red[0];                                     // wide rows: the waves' partials in order
#pragma unroll
        for (int i = 1; i < NT / 64; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
        return r;
    };
    float mx = -INFINITY, ts = 0.f;
    for (int j = tid; j < width; j += NT) { mx = fmaxf(mx, o[j]); ts += (float)t[j]; }
    mx = block_reduce(mx, true);
    ts = block_reduce(ts, false);
    float se = 0.f;
    for (int j = tid; j < width; j += NT) se += expf(o[j] - mx);
    se = block_reduce(se, false);
    const float lse = logf(se);
    float loss = 0.f;
    for (int j = tid; j < width; j += NT) {
        const float logp = (o[j] - mx) - lse;
        const float tn = (float)t[j] / ts;                // 0/0 = NaN on all-zero rows, as the reference
        loss += ((tn > 0.f) ? tn * logf(tn) : (tn == 0.f ? 0.f : tn)) - tn * logp;
        if (d_out) d_out[(long long)b * ldd + j] = (expf(logp) * (ts / ts) - tn) * inv_b;
    }
    loss = block_reduce(loss, false);
    if (tid == 0) row_loss[b] = loss;
}

// loss = (sum_b row_loss[b]) * inv_b in a fixed order (one workgroup; deterministic)
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ row, int B,
                                                        float inv_b, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) acc += row[i];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = ((red[0] + red[1]) + (red[2] + red[3])) * inv_b;
}

// x *= *scale (the upstream gradient of the scalar loss, read on the device)
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(float* __restrict__ x, long long n4,
                                                              long long n,
                                                              const float* __restrict__ scale) {
    const float s = *scale;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        v4f v = ((v4f*)x)[i];
        ((v4f*)x)[i] = v * s;
    } else if (i == n4) {
        for (long long k = 4 * n4; k < n; ++k) x[k] *= s;
    }
}

}  // namespace

// ---- profiling registry ---------------------------------------------------------------------------
#include <vector>
namespace {
struct ProfRec { int kind; double work; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
double g_pipe_ms[GI_PROF_PIPES], g_pipe_work[GI_PROF_PIPES];
int g_pipe_n[GI_PROF_PIPES];
}  // namespace
bool gi_prof_on() { return g_prof_on; }
void gi_prof_push(int kind, double work, hipEvent_t a, hipEvent_t b) { g_prof.push_back({kind, work, a, b}); }

extern "C" int gi_prof_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}
// ms[k], work[k], launches[k] for k in {GEMM (work = flops), SEGSUM (work = bytes)}; clears the log
extern "C" int gi_prof_collect(double* ms, double* busy_ms, double* work, int* launches) {
    if (!ms || !busy_ms || !work || !launches) return GI_EINVAL;
    for (int k = 0; k < GI_PROF_KINDS; ++k) { ms[k] = 0; busy_ms[k] = 0; work[k] = 0; launches[k] = 0; }
    for (int k = 0; k < GI_PROF_PIPES; ++k) { g_pipe_ms[k] = 0; g_pipe_work[k] = 0; g_pipe_n[k] = 0; }
    int rc = 0;
    if (g_prof.empty()) return 0;
    for (ProfRec& r : g_prof) (void)hipEventSynchronize(r.b);
    // absolute [start, stop] of every launch relative to the first recorded event; launches on the
    // two streams of the backward overlap, so the per-family busy time is the UNION of intervals
    std::vector<std::pair<double, double>> iv[GI_PROF_KINDS];
    const hipEvent_t base = g_prof.front().a;
    for (ProfRec& r : g_prof) {
        float t = 0.f, t0 = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) rc = GI_EINVAL;
        if (hipEventElapsedTime(&t0, base, r.a) != hipSuccess) rc = GI_EINVAL;
        const int kind = r.kind & 0xff, pipe = std::min(r.kind >> 8, GI_PROF_PIPES - 1);
        ms[kind] += t; work[kind] += r.work; launches[kind] += 1;
        iv[kind].push_back({(double)t0, (double)t0 + t});
        if (kind == GI_PROF_GEMM) { g_pipe_ms[pipe] += t; g_pipe_work[pipe] += r.work; g_pipe_n[pipe] += 1; }
    }
    for (int k = 0; k < GI_PROF_KINDS; ++k) {
        std::sort(iv[k].begin(), iv[k].end());
        double end = -1e300;
        for (const auto& x : iv[k]) {
            if (x.first > end) { busy_ms[k] += x.second - x.first; end = x.second; }
            else if (x.second > end) { busy_ms[k] += x.second - end; end = x.second; }
        }
    }
    for (ProfRec& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    return rc;
}

// the GEMM family of the LAST gi_prof_collect by matrix pipe: [0] fp32 MFMA, [1] bf16 MFMA (bf16x3), [2] f16 MFMA (fp16x2)
extern "C" int gi_prof_pipes(double* ms, double* work, int* launches) {
    if (!ms || !work || !launches) return GI_EINVAL;
    for (int k = 0; k < GI_PROF_PIPES; ++k) { ms[k] = g_pipe_ms[k]; work[k] = g_pipe_work[k]; launches[k] = g_pipe_n[k]; }
    return 0;
}

// ================================ C ABI ==========================================================
extern "C" int gi_seg_sum(const float* vals, int ldv, const int* perm, const int* off, int rows,
                          int cols, float* out, int ldo, int accumulate, void* stream) {
    return gi_seg_sum_n(vals, ldv, perm, off, rows, cols, out, ldo, accumulate, nullptr, stream);
}
// rows_dev != NULL: bounded launch — `rows` sizes the grid, min(rows, *rows_dev) rows are processed
int gi_seg_sum_n(const float* vals, int ldv, const int* perm, const int* off, int rows, int cols, float* out,
                 int ldo, int accumulate, const int* rows_dev, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0) return 0;
    if (!vals || !off || !out || cols <= 0 || (ldv & 3) || (ldo & 3) || ldv < cols || ldo < cols)
        return GI_EINVAL;
    if (((uintptr_t)vals & 15) || ((uintptr_t)out & 15)) return GI_EINVAL;
    const int c4n = (cols + 3) / 4;
    const long long threads = (long long)rows * c4n;
    // (algorithmic bytes depend on the device-side segment lengths; the caller knows them)
    GiProfScope prof((hipStream_t)stream, GI_PROF_SEGSUM, 0.0);
    // U outputs per thread pay when the launch has far more threads than the device holds at once (HBM-resident
    // operands: the 30 M-thread probe); the step's own launches (230 k threads on 10 MB that sit in L2) are faster with
    // every output on its own thread: 5.0 against 9.6 us (profiles/r05)
    static const int U_env = [] { const char* e = getenv("GI_SEGSUM_U"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
    const int U = U_env ? U_env : (threads >= (4LL << 20) ? 4 : 1);
    const unsigned blocks = (unsigned)((threads + 256LL * U - 1) / (256LL * U));
#define GI_SEGSUM_LAUNCH(UU) hipLaunchKernelGGL(seg_sum_kernel<UU>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, vals, \
                                                ldv, perm, off, rows, c4n, out, ldo, accumulate, rows_dev)
    if (U == 1) GI_SEGSUM_LAUNCH(1); else if (U == 2) GI_SEGSUM_LAUNCH(2); else GI_SEGSUM_LAUNCH(4);
#undef GI_SEGSUM_LAUNCH
    return gi_launch_status();
}

static int seg_softmax_args_ok(const void* en, const void* emb, int ld, const int* perm,
                               const int* off, int cols) {
    if (!en || !emb || !perm || !off || cols <= 0 || (ld & 3) || ld < cols) return 0;
    return !(((uintptr_t)en & 15) || ((uintptr_t)emb & 15));
}

extern "C" int gi_seg_softmax_fwd(const float* en, const float* emb, int ld, const int* perm,
                                  const int* off, int rows, int cols, float* out, int ldo,
                                  void* stream) {
    return gi_seg_softmax_fwd_n(en, emb, ld, perm, off, rows, cols, out, ldo, nullptr, stream);
}
int gi_seg_softmax_fwd_n(const float* en, const float* emb, int ld, const int* perm, const int* off, int rows,
                         int cols, float* out, int ldo, const int* rows_dev, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0) return 0;
    if (!seg_softmax_args_ok(en, emb, ld, perm, off, cols) || !out || (ldo & 3) || ldo < cols ||
        ((uintptr_t)out & 15))
        return GI_EINVAL;
    const int c4n = (cols + 3) / 4;
    const long long threads = (long long)rows * c4n;
    GiProfScope prof((hipStream_t)stream, GI_PROF_SEGSUM, 0.0);
    hipLaunchKernelGGL(seg_softmax_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256),
                       0, (hipStream_t)stream, en, emb, ld, perm, off, rows, c4n, out, ldo, rows_dev);
    return gi_launch_status();
}

extern "C" int gi_seg_softmax_bwd(const float* en, const float* emb, int ld, const int* perm,
                                  const int* off, int rows, int cols, const float* dagg, int ldd,
                                  float* d_en_e, float* d_emb_e, int lde, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0) return 0;
    if (!seg_softmax_args_ok(en, emb, ld, perm, off, cols) || !dagg || (ldd & 3) || ldd < cols ||
        ((uintptr_t)dagg & 15) || !d_en_e || !d_emb_e || (lde & 3) || lde < cols ||
        ((uintptr_t)d_en_e & 15) || ((uintptr_t)d_emb_e & 15))
        return GI_EINVAL;
    const int c4n = (cols + 3) / 4;
    const long long threads = (long long)rows * c4n;
    GiProfScope prof((hipStream_t)stream, GI_PROF_SEGSUM, 0.0);
    hipLaunchKernelGGL(seg_softmax_bwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256),
                       0, (hipStream_t)stream, en, emb, ld, perm, off, rows, c4n, dagg, ldd, d_en_e,
                       d_emb_e, lde);
    return gi_launch_status();
}

extern "C" int gi_seg_sum_dselu(const float* vals, int ldv, const int* perm, const int* off,
                                int rows, int cols, float* y, int ldy, void* stream) {
    return gi_seg_sum_dselu_f(vals, ldv, perm, off, rows, cols, y, ldy, 0, stream);
}

extern "C" int gi_seg_sum_dselu_f(const float* vals, int ldv, const int* perm, const int* off,
                                  int rows, int cols, float* y, int ldy, long long fshift,
                                  void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0) return 0;
    if (fshift & 3) return GI_EINVAL;
    if (!vals || !perm || !off || !y || cols <= 0 || (ldv & 3) || (ldy & 3) || ldv < cols ||
        ldy < cols)
        return GI_EINVAL;
    if (((uintptr_t)vals & 15) || ((uintptr_t)y & 15)) return GI_EINVAL;
    const int c4n = (cols + 3) / 4;
    const long long threads = (long long)rows * c4n;
    GiProfScope prof((hipStream_t)stream, GI_PROF_SEGSUM, 0.0);
    hipLaunchKernelGGL(seg_sum_dselu_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, vals, ldv, perm, off, rows, c4n, y, ldy, fshift);
    return gi_launch_status();
}

extern "C" int gi_class_sum_dselu(const float* vals0, const float* vals1, int ldv, const int* idx,
                                  const int* off, int rows, int cols, float* y0, float* y1, int ldy,
                                  void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0 || cols <= 0) return 0;
    if (!vals0 || !y0 || !idx || !off || ldv < cols || ldy < cols || (vals1 && !y1)) return GI_EINVAL;
    ClassSumArgs a;
    a.vals[0] = vals0; a.vals[1] = vals1; a.y[0] = y0; a.y[1] = y1;
    a.ldv = ldv; a.ldy = ldy; a.cols = cols; a.idx = idx; a.off = off;
    hipLaunchKernelGGL(class_sum_dselu_kernel, dim3(rows, (cols + 63) / 64, vals1 ? 2 : 1), dim3(1024), 0,
                       (hipStream_t)stream, a);
    return gi_launch_status();
}

extern "C" int gi_slab_sum_dselu(const float* slabs, int nsplit, long long stride, int rows, int cols,
                                 int ld, float* y, int ldy, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0 || cols <= 0) return 0;
    if (!slabs || !y || nsplit < 1 || ld < cols || ldy < cols) return GI_EINVAL;
    const long long outputs = (long long)rows * cols;
    hipLaunchKernelGGL(slab_sum_dselu_kernel, dim3((unsigned)((outputs + 15) / 16)), dim3(256), 0,
                       (hipStream_t)stream, slabs, nsplit, stride, rows, cols, ld, y, ldy);
    return gi_launch_status();
}

extern "C" int gi_slab_epilogue(const float* slabs, int nsplit, long long stride, int rows, int cols,
                                int ld, int flags, const float* bias, const float* act, int ldact,
                                float* out, int ldo, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0 || cols <= 0) return 0;
    if (!slabs || !out || nsplit < 1 || ld < cols || ldo < cols) return GI_EINVAL;
    if ((flags & GI_EPI_BIAS) && !bias) return GI_EINVAL;
    if ((flags & (GI_EPI_DSELU | GI_EPI_MULACT)) && (!act || ldact < cols)) return GI_EINVAL;
    const long long threads = (long long)rows * cols;
    hipLaunchKernelGGL(slab_epilogue_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, slabs, nsplit, stride, rows, cols, ld, flags, bias, act, ldact,
                       out, ldo);
    return gi_launch_status();
}

extern "C" int gi_selu_bwd_rows(const float* dY, int lddy, const int* idx, const float* Y, int ldy,
                                float* out, int ldo, int rows, int cols, void* stream) {
    return gi_selu_bwd_rows_f(dY, lddy, idx, Y, ldy, out, ldo, rows, cols, 0, stream);
}

extern "C" int gi_selu_bwd_rows_f(const float* dY, int lddy, const int* idx, const float* Y, int ldy,
                                  float* out, int ldo, int rows, int cols, long long fshift,
                                  void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0 || cols <= 0) return 0;
    if (!dY || !Y || !out) return GI_EINVAL;
    const long long threads = (long long)rows * cols;
    hipLaunchKernelGGL(selu_bwd_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dY, lddy, idx, Y, ldy, out, ldo, rows, cols, fshift);
    return gi_launch_status();
}

extern "C" int gi_selu_bwd_cols3_f(const float* dY, int lddy, const float* Y, int ldy, long long fshift,
                                   int rows, int n0, float* out0, int ld0, int n1, float* out1, int ld1,
                                   int n2, float* out2, int ld2, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0) return 0;
    if (n0 < 0 || n1 < 0 || n2 < 0 || n0 + n1 + n2 <= 0) return GI_EINVAL;
    if (!dY || !Y || (n0 && !out0) || (n1 && !out1) || (n2 && !out2)) return GI_EINVAL;
    if (lddy < n0 + n1 + n2 || ldy < n0 + n1 + n2 || ld0 < n0 || ld1 < n1 || ld2 < n2) return GI_EINVAL;
    const long long threads = (long long)rows * (n0 + n1 + n2);
    hipLaunchKernelGGL(selu_bwd_cols3_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dY, lddy, Y, ldy, fshift, rows, n0, n1, n2, out0, ld0, out1,
                       ld1, out2, ld2);
    return gi_launch_status();
}

// constants of one AlphaDropout site, rounded the way ATen rounds them (double scalars applied to a
// float tensor: each scalar is cast to float first)
extern "C" int gi_dropout_setup(double p, unsigned long long seed, unsigned id, gi_dropout_params* out) {
    if (!out || !(p >= 0.0) || !(p < 1.0)) return GI_EINVAL;
    const double alpha = 1.7580993408473766;
    const double a = 1.0 / sqrt((alpha * alpha * p + 1.0) * (1.0 - p));
    const float c1 = (float)(alpha * a), c2 = (float)(alpha * a * p);
    out->seed = seed; out->id = id;
    const double t = floor(p * 4294967296.0);
    out->thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    out->a = (float)a;
    out->b_keep = c2;                       // (1 - 1) * c1 + c2
    out->b_drop = -c1 + c2;                 // (0 - 1) * c1 + c2, one fp32 rounding like add_
    return 0;
}

extern "C" int gi_alpha_dropout_fwd(float* y, int ldy, int rows, int cols, long long fshift,
                                    const gi_dropout_params* q, void* stream) {
    (void)hipGetLastError();
    if (rows <= 0 || cols <= 0) return 0;
    if (!y || !q || ldy < cols || fshift == 0) return GI_EINVAL;
    const long long threads = (long long)rows * cols;
    hipLaunchKernelGGL(alpha_dropout_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, y, ldy, rows, cols, fshift, *q);
    return gi_launch_status();
}

extern "C" int gi_dropout_mask(const gi_dropout_params* q, int rows, int cols, unsigned char* keep,
                               int ld, void* stream) {
    (void)hipGetLastError();
    if (rows <= 0 || cols <= 0) return 0;
    if (!q || !keep || ld < cols) return GI_EINVAL;
    const long long threads = (long long)rows * cols;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, *q, rows, cols, keep, ld);
    return gi_launch_status();
}

// ---- launch-count reductions of the training step (GI_FUSE) ------------------------------------------
// Bit mask, read once from the environment (GI_FUSE=<int>; default GI_FUSE_DEFAULT): which of the fused /
// vectorised variants of the small kernels around the GEMMs the model uses (include/graphinvent_amd.h:
// the fusions are bit-identical to the launches they replace, the vector gate kernels agree with the scalar
// ones to rounding; tests/test_kernels_gpu.py).
extern "C" int gi_fuse_flags(void) {
    static const int v = getenv("GI_FUSE") ? atoi(getenv("GI_FUSE")) : GI_FUSE_DEFAULT;
    return v;
}

// 16-byte accesses of the vectorised GRU gate kernels: H and every leading dimension a multiple of 4
// floats, every base pointer 16-byte aligned
static bool gates_v4_ok(int H, int ldg, int ldh, int lddh, const void* a, const void* b, const void* c,
                        const void* d) {
    if ((H & 3) || (ldg & 3) || (ldh & 3) || (lddh & 3) || H < 4) return false;
    return !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15);
}

extern "C" int gi_gru_gates_fwd(float* gi, float* gh, int ldg, const float* hx_prev, float* hx_new,
                                int ldh, const int* seg_off, int rows, int H, int Fn,
                                void* stream) {
    return gi_gru_gates_fwd_n(gi, gh, ldg, hx_prev, hx_new, ldh, seg_off, rows, H, Fn, nullptr, stream);
}
int gi_gru_gates_fwd_n(float* gi, float* gh, int ldg, const float* hx_prev, float* hx_new, int ldh,
                       const int* seg_off, int rows, int H, int Fn, const int* rows_dev, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0) return 0;
    if (!gi || !gh || !hx_prev || !hx_new || !seg_off || ldg < 3 * H || ldh < H + Fn)
        return GI_EINVAL;
    if ((gi_fuse_flags() & GI_FUSE_GATES_V4) && gates_v4_ok(H, ldg, ldh, ldh, gi, gh, hx_prev, hx_new)) {
        const long long threads = (long long)rows * (ldh / 4);
        hipLaunchKernelGGL(gru_gates_fwd_v4_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, gi, gh, ldg, hx_prev, hx_new, ldh, seg_off, rows, H, rows_dev);
        return gi_launch_status();
    }
    const long long threads = (long long)rows * ldh;
    hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, gi, gh, ldg, hx_prev, hx_new, ldh, seg_off, rows, H, rows_dev);
    return gi_launch_status();
}

extern "C" int gi_gru_gates_bwd(float* gi, float* gh, int ldg, const float* hx_prev, int ldh,
                                const float* dh_new, const float* dh_b, const float* dh_c,
                                const float* dh_d, float* dh_prev, int lddh, const int* seg_off,
                                int rows, int H, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0) return 0;
    if (!gi || !gh || !hx_prev || !dh_new || !dh_prev || !seg_off || ldg < 3 * H || lddh < H)
        return GI_EINVAL;
    if (gi_fuse_flags() & GI_FUSE_GATES_V4)
        return gi_gru_gates_bwd_ex(gi, gh, ldg, hx_prev, ldh, dh_new, dh_b, dh_c, dh_d, dh_prev, lddh,
                                   seg_off, rows, H, nullptr, nullptr, 0, nullptr, nullptr, stream);
    const long long threads = (long long)rows * H;
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, gi, gh, ldg, hx_prev, ldh, dh_new, dh_b, dh_c, dh_d,
                       dh_prev, lddh, seg_off, rows, H);
    return gi_launch_status();
}

extern "C" int gi_gru_gates_bwd_ex(float* gi, float* gh, int ldg, const float* hx_prev, int ldh,
                                   const float* dh_new, const float* dh_b, const float* dh_c,
                                   const float* dh_d, float* dh_prev, int lddh, const int* seg_off,
                                   int rows, int H, const float* sc0, const float* sc1, int ldsc,
                                   const int* sc_perm, const int* sc_off, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (rows <= 0) return 0;
    if (!gi || !gh || !hx_prev || !dh_new || !dh_prev || !seg_off || ldg < 3 * H || lddh < H || ldh < H)
        return GI_EINVAL;
    if (sc1 && !sc0) return GI_EINVAL;
    if (sc0 && (!sc_perm || !sc_off || ldsc < H || (ldsc & 3) || ((uintptr_t)sc0 & 15) ||
                ((uintptr_t)sc1 & 15)))
        return GI_EINVAL;
    bool v4 = gates_v4_ok(H, ldg, ldh, lddh, gi, gh, hx_prev, dh_new) && !((uintptr_t)dh_prev & 15) &&
              !(((uintptr_t)dh_b | (uintptr_t)dh_c | (uintptr_t)dh_d) & 15);
    if (!v4) {
        if (sc0) return GI_EINVAL;          // the fused scatter exists in the vector kernel only
        const long long threads = (long long)rows * H;
        hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, gi, gh, ldg, hx_prev, ldh, dh_new, dh_b, dh_c, dh_d,
                           dh_prev, lddh, seg_off, rows, H);
        return gi_launch_status();
    }
    const long long threads = (long long)rows * (H / 4);
    hipLaunchKernelGGL(gru_gates_bwd_v4_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, gi, gh, ldg, hx_prev, ldh, dh_new, dh_b, dh_c, dh_d, dh_prev,
                       lddh, seg_off, rows, H, sc0, sc1, ldsc, sc_perm, sc_off);
    return gi_launch_status();
}

extern "C" int gi_gather_readout_fwd(const float* en, const float* emb, int ld, const int* cidx,
                                     const int* node_mask, int B, int N, int G, float big,
                                     float* out0, int ld0, float* out1, int ld1, float* out2,
                                     int ld2, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (B <= 0) return 0;
    if (!en || !emb || !cidx || !node_mask || N <= 0 || N > GI_MAX_NODES || G <= 0 || ld < G)
        return GI_EINVAL;
    if (N > 112)
        hipLaunchKernelGGL(gather_fwd_kernel<32>, dim3(B, gi_cdiv(G, 32)), dim3(32),
                           sizeof(float) * 2 * N * 32, (hipStream_t)stream, en, emb, ld, cidx, node_mask,
                           N, G, big, out0, ld0, out1, ld1, out2, ld2);
    else
        hipLaunchKernelGGL(gather_fwd_kernel<64>, dim3(B, gi_cdiv(G, 64)), dim3(64),
                           sizeof(float) * 2 * N * 64, (hipStream_t)stream, en, emb, ld, cidx, node_mask,
                           N, G, big, out0, ld0, out1, ld1, out2, ld2);
    return gi_launch_status();
}

extern "C" int gi_gather_readout_bwd(float* en, float* emb, int ld, const int* cidx,
                                     const int* node_mask, int B, int N, int G, int S, float big,
                                     const float* dg0, int ld0, const float* dg1, int ld1,
                                     const float* dg2, int ld2, float* zpart, void* stream) {
    return gi_gather_readout_bwd_f(en, emb, ld, cidx, node_mask, B, N, G, S, big, dg0, ld0, dg1, ld1,
                                   dg2, ld2, zpart, 0, stream);
}

extern "C" int gi_gather_readout_bwd_f(float* en, float* emb, int ld, const int* cidx,
                                       const int* node_mask, int B, int N, int G, int S, float big,
                                       const float* dg0, int ld0, const float* dg1, int ld1,
                                       const float* dg2, int ld2, float* zpart, long long fshift,
                                       void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (B <= 0) return 0;
    if (!en || !emb || !cidx || !node_mask || !zpart || N <= 0 || N > GI_MAX_NODES || G <= 0)
        return GI_EINVAL;
    if (N > 112)
        hipLaunchKernelGGL(gather_bwd_kernel<32>, dim3(B, gi_cdiv(G, 32)), dim3(32),
                           sizeof(float) * 2 * N * 32, (hipStream_t)stream, en, emb, ld, cidx, node_mask,
                           N, G, S, big, dg0, ld0, dg1, ld1, dg2, ld2, zpart, fshift);
    else
        hipLaunchKernelGGL(gather_bwd_kernel<64>, dim3(B, gi_cdiv(G, 64)), dim3(64),
                           sizeof(float) * 2 * N * 64, (hipStream_t)stream, en, emb, ld, cidx, node_mask,
                           N, G, S, big, dg0, ld0, dg1, ld1, dg2, ld2, zpart, fshift);
    return gi_launch_status();
}

extern "C" int gi_expand_slots(const float* t1, int ldt, const int* cidx, int B, int N, int W,
                               float* cat, int ldc, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (B <= 0) return 0;
    if (!t1 || !cidx || !cat || N <= 0 || W <= 0 || ldt < W || ldc < N * W) return GI_EINVAL;
    const long long total = (long long)B * N * W;
    hipLaunchKernelGGL(expand_slots_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, t1, ldt, cidx, N * W, W, cat, ldc, total);
    return gi_launch_status();
}

extern "C" int gi_compress_slots(float* t1, int ldt, const int* cidx, int B, int N, int W, int S,
                                 const float* dcat, int ldc, float* zpart, int ldz, void* stream) {
    return gi_compress_slots_f(t1, ldt, cidx, B, N, W, S, dcat, ldc, zpart, ldz, 0, stream);
}

extern "C" int gi_compress_slots_f(float* t1, int ldt, const int* cidx, int B, int N, int W, int S,
                                   const float* dcat, int ldc, float* zpart, int ldz,
                                   long long fshift, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (B <= 0) return 0;
    if (!t1 || !cidx || !dcat || !zpart || N <= 0 || W <= 0 || ldz < W) return GI_EINVAL;
    if (N > GI_MAX_NODES) return GI_ELIMIT;
    const int parts = (long long)N * W >= 4096 ? 4 : 1;
    hipLaunchKernelGGL(compress_slots_kernel, dim3(B, parts), dim3(256), 0, (hipStream_t)stream, t1, ldt,
                       cidx, N, W, S, dcat, ldc, zpart, ldz, fshift);
    return gi_launch_status();
}

extern "C" int gi_expand_slots2(const float* t1a, int ldta, int Wa, float* cata, int ldca,
                                const float* t1b, int ldtb, int Wb, float* catb, int ldcb,
                                const int* cidx, int B, int N, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (B <= 0) return 0;
    if (!t1a || !t1b || !cata || !catb || !cidx || N <= 0 || Wa <= 0 || Wb <= 0 || ldta < Wa ||
        ldtb < Wb || ldca < N * Wa || ldcb < N * Wb)
        return GI_EINVAL;
    ExpandPair a;
    a.t1[0] = t1a; a.t1[1] = t1b; a.ldt[0] = ldta; a.ldt[1] = ldtb; a.W[0] = Wa; a.W[1] = Wb;
    a.cat[0] = cata; a.cat[1] = catb; a.ldc[0] = ldca; a.ldc[1] = ldcb;
    const long long total_a = (long long)B * N * Wa, total = total_a + (long long)B * N * Wb;
    hipLaunchKernelGGL(expand_slots2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a, cidx, N, total_a, total);
    return gi_launch_status();
}

extern "C" int gi_compress_slots2_f(float* t1a, int ldta, int Wa, const float* dcata, int ldca,
                                    float* zparta, int ldza, float* t1b, int ldtb, int Wb,
                                    const float* dcatb, int ldcb, float* zpartb, int ldzb,
                                    const int* cidx, int B, int N, int S, long long fshift, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (B <= 0) return 0;
    if (!t1a || !t1b || !dcata || !dcatb || !zparta || !zpartb || !cidx || N <= 0 || Wa <= 0 || Wb <= 0 ||
        ldza < Wa || ldzb < Wb)
        return GI_EINVAL;
    if (N > GI_MAX_NODES) return GI_ELIMIT;
    CompressPair a;
    a.t1[0] = t1a; a.t1[1] = t1b; a.ldt[0] = ldta; a.ldt[1] = ldtb; a.W[0] = Wa; a.W[1] = Wb;
    a.dcat[0] = dcata; a.dcat[1] = dcatb; a.ldc[0] = ldca; a.ldc[1] = ldcb;
    a.zpart[0] = zparta; a.zpart[1] = zpartb; a.ldz[0] = ldza; a.ldz[1] = ldzb;
    const int Wmax = Wa > Wb ? Wa : Wb;
    const int parts = (long long)N * Wmax >= 4096 ? 4 : 1;
    hipLaunchKernelGGL(compress_slots2_kernel, dim3(B, parts, 2), dim3(256), 0, (hipStream_t)stream, a,
                       cidx, N, S, fshift);
    return gi_launch_status();
}

extern "C" int gi_colsum(const float* part, int ldp, int rows, int cols, const float* y,
                         float* out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (cols <= 0) return 0;
    if (!part || !out || rows < 0) return GI_EINVAL;
    ColsumTable tab;
    memset(&tab, 0, sizeof(tab));
    tab.d[0] = gi_colsum_desc{part, ldp, rows, cols, y, out};
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, 1), dim3(1024), 0, (hipStream_t)stream,
                       tab);
    return gi_launch_status();
}

extern "C" int gi_colsum_multi(const gi_colsum_desc* descs, int n, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (n <= 0) return 0;
    if (!descs || n > 8) return GI_EINVAL;
    ColsumTable tab;
    memset(&tab, 0, sizeof(tab));
    int maxcols = 0;
    for (int i = 0; i < n; ++i) {
        if (!descs[i].part || !descs[i].out || descs[i].rows < 0 || descs[i].cols < 0) return GI_EINVAL;
        tab.d[i] = descs[i];
        maxcols = descs[i].cols > maxcols ? descs[i].cols : maxcols;
    }
    if (maxcols == 0) return 0;
    hipLaunchKernelGGL(colsum_kernel, dim3((maxcols + 63) / 64, n), dim3(1024), 0,
                       (hipStream_t)stream, tab);
    return gi_launch_status();
}

extern "C" int gi_reduce_slabs(const gi_reduce_desc* descs, int n_desc, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (n_desc <= 0) return 0;
    if (!descs) return GI_EINVAL;
    for (int base = 0; base < n_desc; base += GI_REDUCE_MAX) {
        const int n = (n_desc - base < GI_REDUCE_MAX) ? n_desc - base : GI_REDUCE_MAX;
        ReduceTable tab;
        int total = 0;
        for (int i = 0; i < n; ++i) {
            tab.d[i] = descs[base + i];
            tab.start[i] = total;
            total += (int)(((long long)tab.d[i].N * (tab.d[i].K + 1) + 255) / 256);
        }
        for (int i = n; i < GI_REDUCE_MAX; ++i) tab.d[i] = descs[base];
        tab.start[n] = total;
        tab.n = n;
        if (total == 0) continue;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, tab);
        const int rc = gi_launch_status();
        if (rc) return rc;
    }
    return 0;
}

extern "C" int gi_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr,
                            double beta1, double beta2, double eps, double weight_decay, int step,
                            void* stream) {
    (void)hipGetLastError();
    if (n <= 0) return 0;
    if (!p || !g || !m || !v || step < 1 || (n & 3)) return GI_EINVAL;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return GI_EINVAL;
    // every scalar is formed in double like torch.optim.Adam forms it, then rounded once: 1 - beta2
    // from a float beta2 is 1.3e-5 off, powf(beta, step) ~6e-5 at step 1
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2_sqrt = sqrt(1.0 - pow(beta2, (double)step));
    const long long n4 = n / 4;
    const int blocks = (int)std::min<long long>((n4 + 255) / 256, 4096);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4,
                       (float)(lr / bc1), (float)beta1, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)bc2_sqrt);
    return gi_launch_status();
}

extern "C" int gi_scale_by_scalar(float* x, long long n, const float* scale, void* stream) {
    (void)hipGetLastError();
    if (n <= 0) return 0;
    if (!x || !scale || ((uintptr_t)x & 15)) return GI_EINVAL;
    const long long n4 = n / 4;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)((n4 + 1 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, n4, n, scale);
    return gi_launch_status();
}

extern "C" int gi_kl_loss(const float* out, int ldo, const void* target, int tgt_dtype, int ldt, int B,
                          int width, float* row_loss, float* d_out, int ldd, float* loss_mean,
                          void* stream) {
    (void)hipGetLastError();
    if (B <= 0) return 0;
    if (!out || !target || !row_loss || width <= 0 || ldo < width || ldt < width) return GI_EINVAL;
    const hipStream_t st = (hipStream_t)stream;
    // rows wider than 2048 logits (ZINC / ChEMBL shapes: 4 k - 10 k) get 1024 threads each
#define GI_KL_LAUNCH(T_, NT_)                                                                     \
    hipLaunchKernelGGL((kl_loss_kernel<T_, NT_>), dim3(B), dim3(NT_), 0, st, out, ldo,            \
                       (const T_*)target, ldt, width, 1.f / (float)B, row_loss, d_out, ldd)
    if (tgt_dtype == GI_DTYPE_F32) {
        if (width > 2048) GI_KL_LAUNCH(float, 1024); else GI_KL_LAUNCH(float, 256);
    } else if (tgt_dtype == GI_DTYPE_I8) {
        if (width > 2048) GI_KL_LAUNCH(signed char, 1024); else GI_KL_LAUNCH(signed char, 256);
    }
#undef GI_KL_LAUNCH
    else
        return GI_EINVAL;
    if (loss_mean)       // the batch mean right behind the row kernel, fixed summation order
        hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, st, row_loss, B, 1.f / (float)B,
                           loss_mean);
    return gi_launch_status();
}


// ---- largest magnitude of several tensors (the weights of the fp16x2 GEMM launches, gi_x2.h) -----------------------
#include "gi_x2.h"
namespace {
struct AbsmaxArgs { gi_absmax_desc d[GI_ABSMAX_MAX]; int start[GI_ABSMAX_MAX + 1]; int n; };
// a workgroup = 256 threads x 8 float4 of one tensor (contiguous tensors: rows folded into one run by the host)
__global__ __launch_bounds__(256) void absmax_kernel(const AbsmaxArgs a) {
    int i = 0;
    while (i < a.n - 1 && (int)blockIdx.x >= a.start[i + 1]) ++i;
    const gi_absmax_desc& d = a.d[i];
    const long long n = (long long)d.rows * d.cols;                     // (ld == cols here)
    const long long base = (long long)(blockIdx.x - a.start[i]) * 8192;
    float m = 0.f;
    if (((uintptr_t)d.x & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long e = base + ((long long)k * 256 + threadIdx.x) * 4;
            if (e + 3 < n) {
                const v4f v = *reinterpret_cast<const v4f*>(d.x + e);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            } else {
                for (long long q = e; q < n && q < e + 4; ++q) m = fmaxf(m, fabsf(d.x[q]));
            }
        }
    } else {
        for (long long e = base + threadIdx.x; e < n && e < base + 8192; e += 256) m = fmaxf(m, fabsf(d.x[e]));
    }
    gx_amax_publish(m, d.out);
}
// rows with a pitch (ld > cols; the padding columns are never written: masked): ALL such tensors of a call in one launch,
// a workgroup = a block of whole rows of one tensor, 2 048 float4 slots (round 6: one launch per tensor, one row per
// workgroup pass and scalar loads cost ~10 us per activation tensor — the weight-gradient operands' cells, gi_model.hip)
__global__ __launch_bounds__(256) void absmax_pitched_kernel(const AbsmaxArgs a) {
    int i = 0;
    while (i < a.n - 1 && (int)blockIdx.x >= a.start[i + 1]) ++i;
    const gi_absmax_desc& d = a.d[i];
    const int cols4 = (d.cols + 3) >> 2;
    const int rpw = max(1, 2048 / cols4);                               // rows per workgroup
    const int row0 = ((int)blockIdx.x - a.start[i]) * rpw;
    const int nrows = min(rpw, d.rows - row0);
    const bool vec = (((uintptr_t)d.x & 15) == 0) && (d.ld & 3) == 0;
    float m = 0.f;
    for (int slot = threadIdx.x; slot < nrows * cols4; slot += 256) {
        const int r = slot / cols4, c = 4 * (slot - r * cols4);
        const float* src = d.x + (long long)(row0 + r) * d.ld + c;
        if (vec && c + 3 < d.cols) {
            const v4f v = *reinterpret_cast<const v4f*>(src);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        } else {
            for (int q = 0; q < 4 && c + q < d.cols; ++q) m = fmaxf(m, fabsf(src[q]));
        }
    }
    gx_amax_publish(m, d.out);
}
// ---- bias-gradient column of weight-gradient slabs (GiBiasSlab, gi_common.h) ----------------------------------------
// One workgroup per (problem, 16 output channels, slab): 4 column threads (one float4 each) x 64 row lanes, four rows in
// flight per lane — a column sum is a pure latency problem (11 MB per GRU problem, no reuse), so what counts is bytes in
// flight: ~270 workgroups x 16 KB.  The first version (64 columns x 4 row lanes, two scalar loads in flight: 66
// workgroups x 2 KB) cost the weight-gradient queue MORE than the column of tiles it removed.  Partial sums are
// combined in a fixed order (xor shuffles over the row lanes of a wave, then the four waves through LDS): deterministic.
typedef v4f v4f_unaligned __attribute__((aligned(4)));
constexpr int GI_BIAS_SLAB_MAX = 40;
struct BiasSlabArgs { GiBiasSlab d[GI_BIAS_SLAB_MAX]; int start[GI_BIAS_SLAB_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void bias_slabs_kernel(const BiasSlabArgs a) {
    __shared__ v4f part[4][4];
    int i = 0;
    while (i < a.n - 1 && (int)blockIdx.x >= a.start[i + 1]) ++i;
    const GiBiasSlab& d = a.d[i];
    const int local = blockIdx.x - a.start[i];
    const int s = local % d.nsplit, cb = local / d.nsplit;
    const int lo = d.grp_off ? d.grp_off[d.g] : 0, hi = d.grp_off ? d.grp_off[d.g + 1] : d.rows;
    const int chunk = (hi - lo + d.nsplit - 1) / d.nsplit;
    const int r0 = lo + s * chunk, r1 = min(r0 + chunk, hi);
    const int ct = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int c = cb * 16 + 4 * ct;
    const int cc = min(c, ((d.n_out + 3) & ~3) - 4);             // (a readable float4 of the row: lddz >= r4(n_out))
    const float* src = d.dZ + cc;
    v4f acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int r = r0 + rl;
    for (; r + 192 < r1; r += 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += *reinterpret_cast<const v4f_unaligned*>(src + (long long)(r + 64 * u) * d.lddz);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (r + 64 * u < r1) acc[u] += *reinterpret_cast<const v4f_unaligned*>(src + (long long)(r + 64 * u) * d.lddz);
    v4f sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
        sum.x += __shfl_xor(sum.x, o); sum.y += __shfl_xor(sum.y, o);
        sum.z += __shfl_xor(sum.z, o); sum.w += __shfl_xor(sum.w, o);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane < 4) part[wid][lane] = sum;
    __syncthreads();
    if (threadIdx.x < 4 && c == cc) {
        const v4f t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        float* dst = d.slab + (long long)s * d.stride + d.col;
        const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c + j < d.n_out) dst[(long long)(c + j) * d.ld] = v[j];
    }
}

// fp16x2 dynamic-range guard of the WEIGHTS (gi_x2_weight_guard): rows and columns of a matrix whose largest scaled
// magnitude is below 2^-11 (see gi_gemm_bf3.hip) — an output channel of the forward launch (row of W) or of the dgrad
// launch (column of W) that the per-tensor scale leaves with fewer than ~14 significant bits.  One workgroup per
// (matrix, direction, 32 lines): rows — a wave per row at a time, lanes along the row; columns — 32 adjacent columns
// x 8 row lanes, reduced through LDS.
struct WGuardArgs { gi_absmax_desc d[GI_ABSMAX_MAX]; int start[GI_ABSMAX_MAX + 1]; int n; int* counter; int* host_flag; };
__global__ __launch_bounds__(256) void x2_weight_guard_kernel(const WGuardArgs a) {
    __shared__ float part[8][32];
    int i = 0;
    while (i < a.n - 1 && (int)blockIdx.x >= a.start[i + 1]) ++i;
    const gi_absmax_desc& d = a.d[i];
    int local = blockIdx.x - a.start[i];
    const int row_chunks = (d.rows + 31) / 32;
    float s, inv;
    gx_scale(gx_amax_read(d.out), s, inv);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int n_low = 0;
    if (local < row_chunks) {
        for (int r = local * 32 + wid; r < min(local * 32 + 32, d.rows); r += 4) {
            float m = 0.f;
            for (int c = lane; c < d.cols; c += 64) m = fmaxf(m, fabsf(d.x[(long long)r * d.ld + c]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            n_low += (lane == 0 && m > 0.f && m * s < 0x1p-11f) ? 1 : 0;
        }
    } else {
        local -= row_chunks;
        const int c = local * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
        float m = 0.f;
        if (c < d.cols)
            for (int r = rl; r < d.rows; r += 8) m = fmaxf(m, fabsf(d.x[(long long)r * d.ld + c]));
        part[rl][threadIdx.x & 31] = m;
        __syncthreads();
        if (threadIdx.x < 32) {
#pragma unroll
            for (int k = 1; k < 8; ++k) m = fmaxf(m, part[k][threadIdx.x]);
            n_low += (c < d.cols && m > 0.f && m * s < 0x1p-11f) ? 1 : 0;
        }
    }
    if (n_low) {
        atomicAdd(a.counter, n_low);
        if (a.host_flag) *reinterpret_cast<volatile int*>(a.host_flag) = 1;
    }
}
}  // namespace

int gi_bias_slabs(const GiBiasSlab* descs, int n, hipStream_t st) {
    for (int base = 0; base < n; base += GI_BIAS_SLAB_MAX) {
        BiasSlabArgs a;
        memset(&a, 0, sizeof(a));
        int total = 0, k = 0;
        for (int i = base; i < n && i < base + GI_BIAS_SLAB_MAX; ++i) {
            const GiBiasSlab& d = descs[i];
            if (!d.dZ || !d.slab || d.n_out < 4 || d.nsplit < 1 || d.rows < 0 || d.lddz < ((d.n_out + 3) & ~3) ||
                d.col < 0 || d.col >= d.ld)
                return GI_EINVAL;
            a.d[k] = d;
            a.start[k] = total;
            total += ((d.n_out + 15) / 16) * d.nsplit;
            ++k;
        }
        a.start[k] = total; a.n = k;
        if (total > 0) hipLaunchKernelGGL(bias_slabs_kernel, dim3(total), dim3(256), 0, st, a);
        const int e = gi_launch_status();
        if (e) return e;
    }
    return 0;
}

// rows / columns of up to GI_ABSMAX_MAX weight matrices outside fp16x2's per-tensor range (the cells d.out must hold
// the matrices' amax already: gi_absmax on the same stream first)
extern "C" int gi_x2_weight_guard(const gi_absmax_desc* descs, int n, int* counter, int* host_flag, void* stream) {
    (void)hipGetLastError();
    if (!descs || n < 1 || n > GI_ABSMAX_MAX || !counter) return GI_EINVAL;
    WGuardArgs a;
    memset(&a, 0, sizeof(a));
    int total = 0, k = 0;
    for (int i = 0; i < n; ++i) {
        const gi_absmax_desc& d = descs[i];
        if (!d.x || !d.out || d.rows < 0 || d.cols < 0 || d.ld < d.cols) return GI_EINVAL;
        if (d.rows == 0 || d.cols == 0) continue;
        a.d[k] = d;
        a.start[k] = total;
        total += (d.rows + 31) / 32 + (d.cols + 31) / 32;
        ++k;
    }
    if (k == 0) return 0;
    a.start[k] = total; a.n = k; a.counter = counter; a.host_flag = host_flag;
    hipLaunchKernelGGL(x2_weight_guard_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, a);
    return gi_launch_status();
}

// One int of host memory that kernels can write: the trip flag of the fp16x2 dynamic-range guard (gi_graph.x2_guard_host)
extern "C" int gi_host_flag_create(int** host, int** dev) {
    if (!host || !dev) return GI_EINVAL;
    (void)hipGetLastError();
    void* h = nullptr;
    void* d = nullptr;
    hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return (int)e;
    *reinterpret_cast<volatile int*>(h) = 0;
    e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) { (void)hipHostFree(h); return (int)e; }
    *host = static_cast<int*>(h);
    *dev = static_cast<int*>(d);
    return 0;
}
extern "C" int gi_host_flag_destroy(int* host) {
    if (!host) return 0;
    return (int)hipHostFree(host);
}

extern "C" int gi_absmax(const gi_absmax_desc* descs, int n, void* stream) {
    (void)hipGetLastError();
    if (!descs || n < 1 || n > GI_ABSMAX_MAX) return GI_EINVAL;
    AbsmaxArgs a, ap;
    memset(&a, 0, sizeof(a));
    memset(&ap, 0, sizeof(ap));
    int total = 0, k = 0, ptotal = 0, pk = 0;
    for (int i = 0; i < n; ++i) {
        const gi_absmax_desc& d = descs[i];
        if (!d.x || !d.out || d.rows < 0 || d.cols < 0 || d.ld < d.cols) return GI_EINVAL;
        const long long elems = (long long)d.rows * d.cols;
        if (elems == 0) continue;
        if (d.ld != d.cols && d.rows > 1) {                  // pitched rows: one launch for all of them
            const int rpw = std::max(1, 2048 / ((d.cols + 3) / 4));
            ap.d[pk] = d; ap.start[pk] = ptotal;
            ptotal += gi_cdiv(d.rows, rpw);
            ++pk;
            continue;
        }
        if (elems > 0x7fffffffLL * 64) return GI_ELIMIT;
        a.d[k] = d; a.d[k].rows = 1; a.d[k].cols = (int)std::min<long long>(elems, 0x7fffffff); a.d[k].ld = a.d[k].cols;
        if (elems > 0x7fffffffLL) return GI_ELIMIT;
        a.start[k] = total;
        total += (int)((elems + 8191) / 8192);
        ++k;
    }
    a.start[k] = total; a.n = k;
    ap.start[pk] = ptotal; ap.n = pk;
    if (total > 0) hipLaunchKernelGGL(absmax_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, a);
    if (ptotal > 0) hipLaunchKernelGGL(absmax_pitched_kernel, dim3(ptotal), dim3(256), 0, (hipStream_t)stream, ap);
    return gi_launch_status();
}
