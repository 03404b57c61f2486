// Sampling step of graph generation (gfx950): softmax of the APD logits, one categorical draw per
// graph, action decode and the validity rules — replaces `softmax(self.model(...))` +
// `GraphGenerator.get_actions` / `get_invalid_actions` (GraphGenerator.py:121, 467-657): a
// Multinomial object, a [B, W] one-hot sample, three `nonzero`s, boolean-mask gathers and ~15 small
// index kernels per generation step become ONE launch, one workgroup per graph.
//
// The draw is the inverse CDF of a caller-supplied uniform per graph (torch's Multinomial stream is
// not reproducible by any other implementation; the distribution is the same): the first action
// whose cumulative un-normalised probability exceeds u * total.  Cumulative sums run in a fixed
// order (256 contiguous chunks, sequential inside a chunk), so a (logits, u) pair always gives the
// same action.  HBM-bound: reads B*W*4 bytes once; the row lives in LDS between the passes.
#include "gi_common.h"

namespace {

constexpr int SAMPLE_MAX_W = 15360;        // floats of one APD row kept in LDS (60 KB)

template <typename ET>
__global__ __launch_bounds__(256) void sample_actions_kernel(
    const float* __restrict__ logits, int ldl, const float* __restrict__ uniform,
    const int* __restrict__ n_nodes, const ET* __restrict__ edges, int N, int A, int Fe,
    int* __restrict__ action, float* __restrict__ likelihood, int* __restrict__ flags) {
    __shared__ float e[SAMPLE_MAX_W];
    __shared__ float red[256];
    __shared__ float wtot[4];
    __shared__ int found_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int NA = N * A, NC = N * Fe, W = NA + NC + 1;
    const float* row = logits + (long long)b * ldl;
    // pass 1: row -> LDS, maximum
    float mx = -INFINITY;
    for (int i = tid; i < W; i += 256) {
        const float v = row[i];
        e[i] = v;
        mx = fmaxf(mx, v);
    }
    red[tid] = mx;
    if (tid == 0) found_s = 0x7fffffff;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    // pass 2: un-normalised probabilities; sums of 256 contiguous chunks
    for (int i = tid; i < W; i += 256) e[i] = expf(e[i] - mx);
    __syncthreads();
    const int L = (W + 255) / 256;
    const int lo = min(tid * L, W), hi = min(lo + L, W);
    float csum = 0.f;
    for (int i = lo; i < hi; ++i) csum += e[i];
    // exclusive scan of the chunk sums in chunk order (wave shuffle + 4 wave totals)
    float x = csum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wtot[wid] = x;
    __syncthreads();
    float woff = 0.f;
    for (int w = 0; w < wid; ++w) woff += wtot[w];
    const float excl = woff + x - csum;
    const float total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    const float target = uniform[b] * total;
    // the chunk whose cumulative range contains the target walks its elements
    if (hi > lo && excl <= target) {
        float c = excl;
        for (int i = lo; i < hi; ++i) {
            c += e[i];
            if (c > target) { atomicMin(&found_s, i); break; }
        }
    }
    __syncthreads();
    if (tid != 0) return;
    // every chunk starting at or below the target walks; the one containing it finds c > target.  The
    // smallest index wins (rounding at a chunk edge can give two candidates), and a target that
    // rounding puts past the total falls back to the last action.
    int idx = found_s;
    if (idx >= W) idx = W - 1;
    const int nn = n_nodes[b];
    int kind, node = 0, rem = 0, from = 0, invalid = 0, reset = 0;
    if (idx < NA) {                                  // "add" (GraphGenerator.py:555-557)
        kind = 0; node = idx / A; rem = idx - node * A; from = nn;
        const bool empty = nn == 0;
        if (!empty && node >= nn) invalid = 1;       // :605-608 attach to a non-existing node
        if (empty && node != 0) invalid = 1;         // :611-614 first atom must go to slot 0
        if (from >= N) { invalid = 1; reset = 1; }   // :617 graph is full
        if (empty) reset = 1;                        // :650-654
        if (reset) from = 0;                         // :567
    } else if (idx < NA + NC) {                      // "connect" (:559-561)
        kind = 1;
        const int r = idx - NA;
        node = r / Fe; rem = r - node * Fe; from = nn - 1;
        if (node >= nn) invalid = 1;                 // :620
        if (nn == 0) invalid = 1;                    // :623
        if (node == from) invalid = 1;               // :626 self-loop
        const int fj = from < 0 ? from + N : from;   // torch indexing wraps -1 (:629-633)
        float adj = 0.f;
        const ET* ep = edges + (((long long)b * N + node) * N + fj) * Fe;
        for (int f = 0; f < Fe; ++f) adj += (float)ep[f];
        if (adj == 1.f) invalid = 1;                 // :629-633 bond already there
    } else {
        kind = 2;                                    // "terminate"
    }
    action[4 * b + 0] = kind; action[4 * b + 1] = node; action[4 * b + 2] = rem; action[4 * b + 3] = from;
    likelihood[b] = e[idx] / total;                  // :541 apds[one_hot == 1]
    flags[b] = invalid | (reset << 1);
}

}  // namespace

extern "C" int gi_sample_actions(const float* logits, int ldl, const float* uniform,
                                 const int* n_nodes, const void* edges, int edges_dtype, int B,
                                 int N, int A, int Fe, int* action, float* likelihood, int* flags,
                                 void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (B <= 0) return 0;
    if (!logits || !uniform || !n_nodes || !edges || !action || !likelihood || !flags || N <= 0 ||
        A <= 0 || Fe <= 0)
        return GI_EINVAL;
    const long long W = (long long)N * A + (long long)N * Fe + 1;
    if (W > SAMPLE_MAX_W) return GI_ELIMIT;
    if (ldl < W) return GI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (edges_dtype == GI_DTYPE_F32)
        hipLaunchKernelGGL(sample_actions_kernel<float>, dim3(B), dim3(256), 0, st, logits, ldl,
                           uniform, n_nodes, (const float*)edges, N, A, Fe, action, likelihood, flags);
    else if (edges_dtype == GI_DTYPE_I8)
        hipLaunchKernelGGL(sample_actions_kernel<signed char>, dim3(B), dim3(256), 0, st, logits,
                           ldl, uniform, n_nodes, (const signed char*)edges, N, A, Fe, action,
                           likelihood, flags);
    else
        return GI_EINVAL;
    return gi_launch_status();
}
