// K1 graph_compact: dense one-hot adjacency [B,N,N,Fe] -> compact CSR graph (gfx950).
//
// Replaces gnn/summation_mpnn.py:100-124 (`edges.sum(3)`, `nonzero` x2, the dense [V,E] equality
// matrix, `edges[eb,ei,ej,:]`, the zero-padded hidden state) and :146 (`node_mask`).
// Integer / byte work, HBM-bound (reads B*N*N*Fe*4 + B*N*Fn*4 bytes once); three launches:
//   count : one workgroup per graph; adjacency staged in LDS as int8 bond types; per-slot
//           in-degree, per-type out-degrees, activity flags
//   scan  : one 1024-thread workgroup per scanned array; exclusive scans over the B*N slots ->
//           compact row ids, CSR offsets, message-row ids, totals (S, E, U, U_t, E_t); + finish
//   fill  : one workgroup per graph; dst-CSR, message CSR and source CSR index arrays
//           (deterministic order), initial node rows
// Edge enumeration order = row-major nonzero of the adjacency = the reference's edge order, so
// every destination's edges are one contiguous CSR segment and no atomics are needed anywhere.
//
// MESSAGE ROWS.  The message an edge carries, MLP_type(e)(h_src(e)) (gnn/mpnn.py:284-294), depends
// only on (source node, bond type); a carbon with three single bonds sends the same vector three
// times.  The message MLP therefore runs on one row per distinct (source slot, bond type) pair — U
// rows, 0.55-0.63 E on molecular graphs — ordered bond-type-major (the grouped GEMM's buckets),
// inside a type by source slot.  Aggregation reads message rows through the dst-CSR (`in_perm`);
// its backward sums, per message row, the gradients of the edges that read it (message CSR
// `mu_off / mu_dst / mu_slot`); the input gradient of the message MLP goes back to the nodes
// through the source CSR over message rows (`out_perm`, `src_off`).
//
// PASS-0 ROWS.  At the first message pass h = [x | 0 .. 0] (gnn/summation_mpnn.py:121-126), so a
// message row depends only on (feature row of its source, bond type): D0 distinct rows — atom type x
// formal charge x bond type, a few dozen.  Classes = distinct 0/1 feature rows of the source slots
// (64-bit pattern, LDS hash set + rank sort: deterministic ids), rows = present (bond type, class)
// pairs, bond-type-major.  The pass-0 aggregation becomes agg = cmat . m0 with the [R, D0] matrix of
// edge counts `cmat` (rows owned by the destination's thread: no atomics).  D0 = 0 (feature off)
// when a feature is not 0/1, Fn > 62, or there are more than GI_P0_MAX_CLASSES classes.
#include "gi_common.h"

namespace {

struct Lay {
    int counts, type_off, cidx, node_mask, slot_of, seg_off, src_off, scratch, total;
    int dims;           // device-side launch dimensions of a bounded (host-sync-free) forward, see gi_compact_bound
    // scratch sub-arrays (ints)
    int rowcnt, nmsg, active, seg_start, srcm_start, colcnt_t, cstart_t, mflag_t, mstart_t,
        etype_off, etype;
    // pass-0 rows
    int type_off0, key_lo, key_hi, cls, qkeys, present, rep, dmap, d_slot;
    int d_key;          // per pass-0 row: {bond type, pattern lo, pattern hi, 0} (the key of gi_graph.p0_cache)
};
constexpr int CNT_S = 0, CNT_E = 1, CNT_ERR = 2, CNT_U = 3, CNT_UT = 4, CNT_ET = 12, CNT_D0 = 20,
              CNT_P0BAD = 21, CNT_Q = 22, CNT_NODEDUP = 23, CNT_N = 24;
constexpr int P0Q = GI_P0_MAX_CLASSES;

inline Lay make_layout(int B, int N, int Fe) {
    Lay L;
    const int ns = B * N;
    int o = 0;
    auto take = [&](int n) { int r = o; o += gi_r4(n); return r; };
    L.counts = take(CNT_N);
    L.type_off = take(GI_MAX_GROUPS + 1);
    L.cidx = take(ns);
    L.node_mask = take(ns);
    L.slot_of = take(ns);
    L.seg_off = take(ns + 2);
    L.src_off = take(ns + 2);
    L.type_off0 = take(GI_MAX_GROUPS + 1);
    L.dims = take(GI_DIMS);
    L.scratch = o;
    L.rowcnt = take(ns);
    L.nmsg = take(ns);
    L.active = take(ns);
    L.seg_start = take(ns);
    L.srcm_start = take(ns);
    L.colcnt_t = take(Fe * ns);
    L.cstart_t = take(Fe * ns);
    L.mflag_t = take(Fe * ns);
    L.mstart_t = take(Fe * ns);
    L.etype_off = take(GI_MAX_GROUPS + 1);
    L.key_lo = take(ns); L.key_hi = take(ns); L.cls = take(ns);
    L.qkeys = take(2 * P0Q);
    L.present = take(Fe * P0Q); L.rep = take(Fe * P0Q); L.dmap = take(Fe * P0Q);
    L.d_slot = take(Fe * P0Q);
    L.d_key = take(4 * Fe * P0Q);
    L.etype = take((int)(((long long)B * N * N + 3) / 4));
    L.total = o;
    return L;
}

// ---- count ----------------------------------------------------------------------------------
// T = float (what BlockDatasetLoader.HDFDataset hands the reference model) or signed char (the int8
// the preprocessed HDF actually stores, DataProcesser.py:157-161 — 4x less input traffic, no host cast)
template <typename T>
__global__ __launch_bounds__(256) void compact_count_kernel(
    const T* __restrict__ nodes, const T* __restrict__ edges, int N, int Fn, int Fe,
    int* __restrict__ gfix, Lay L, int nodedup) {
    __shared__ signed char typ[GI_MAX_NODES * GI_MAX_NODES];
    __shared__ int err_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NN = N * N;
    if (tid == 0) err_s = 0;
    __syncthreads();
    const T* eg = edges + (long long)b * NN * Fe;
    signed char* etype_g = reinterpret_cast<signed char*>(gfix + L.etype) + (long long)b * NN;
    // A cell (i, j) holds the SET of bond types between the pair as a bit mask.  The preprocessed data has at most
    // one (a one-hot row), but the reference's generation loop applies actions to its dummy graph 0 for ever without
    // resetting it (GraphGenerator.py:133, 424-427) and so builds cells with several types set; the reference then
    // sums the per-type messages of such a cell (gnn/mpnn.py:286-294: every type's MLP masked by its 0/1 entry),
    // i.e. it behaves like parallel edges.  Each set bit is an edge here; entries other than 0 / 1 are refused.
    for (int idx = tid; idx < NN; idx += 256) {
        int mask = 0, bad = 0;
        for (int f = 0; f < Fe; ++f) {
            const float v = (float)eg[(long long)idx * Fe + f];
            if (v == 1.f) mask |= 1 << f;
            else if (v != 0.f) bad = 1;
        }
        if (bad) err_s |= 1;                     // an edge feature that is not 0 / 1: outside the data contract
        if (mask & (mask - 1)) err_s |= 8;       // several bond types on one pair (legal for GGNN, see above)
        typ[idx] = (signed char)mask;
        etype_g[idx] = (signed char)mask;
    }
    __syncthreads();
    const int ns = gridDim.x * N;
    for (int i = tid; i < N; i += 256) {
        const int slot = b * N + i;
        int rc = 0, cc = 0, nm = 0;
        int cct[GI_MAX_GROUPS];                          // out-degree of slot i per bond type
#pragma unroll
        for (int f = 0; f < GI_MAX_GROUPS; ++f) cct[f] = 0;
        for (int j = 0; j < N; ++j) {
            rc += __popc((unsigned)(unsigned char)typ[i * N + j]);
            const int m = (unsigned char)typ[j * N + i];
            cc += __popc((unsigned)m);
#pragma unroll
            for (int f = 0; f < GI_MAX_GROUPS; ++f) cct[f] += (m >> f) & 1;
        }
        bool nz = false, binary = Fn <= 62;
        unsigned long long key = 0;                      // pattern of the 0/1 feature row
        for (int f = 0; f < Fn; ++f) {
            const float v = (float)nodes[(long long)slot * Fn + f];
            nz |= (v != 0.f);
            if (v == 1.f) key |= 1ull << (f & 63);
            else if (v != 0.f) binary = false;
        }
        if (!binary || nodedup) gfix[L.counts + CNT_P0BAD] = 1;     // benign race: everybody writes 1
        gfix[L.key_lo + slot] = (int)(unsigned)(key & 0xffffffffull);
        gfix[L.key_hi + slot] = (int)(unsigned)(key >> 32);
        gfix[L.rowcnt + slot] = rc;
        gfix[L.active + slot] = (nodedup || nz || rc > 0 || cc > 0) ? 1 : 0;
        gfix[L.node_mask + slot] = rc > 0 ? 1 : 0;                       // :146
#pragma unroll
        for (int f = 0; f < GI_MAX_GROUPS; ++f)
            if (f < Fe) {
                gfix[L.colcnt_t + f * ns + slot] = cct[f];
                // message rows of type f sent by the slot: one per (slot, type) pair, or one per
                // edge without de-duplication
                const int nrow = nodedup ? cct[f] : (cct[f] > 0 ? 1 : 0);
                gfix[L.mflag_t + f * ns + slot] = nrow;
                nm += nrow;
            }
        gfix[L.nmsg + slot] = nm;
    }
    __syncthreads();
    if (tid == 0 && err_s) atomicOr(&gfix[L.counts + CNT_ERR], err_s);
    if (tid == 0 && b == 0) gfix[L.counts + CNT_NODEDUP] = nodedup;
}

// ---- pass-0 classes: distinct feature patterns of the source slots, sorted (one workgroup) ----
__device__ void p0_classes(int ns, int Fe, int* __restrict__ gfix, const Lay& L) {
    constexpr int CAP = 4 * P0Q;
    constexpr unsigned long long EMPTY = ~0ull;
    __shared__ unsigned long long tab[CAP];
    __shared__ unsigned long long list[P0Q];
    __shared__ int n_s, bad_s;
    const int tid = threadIdx.x;
    for (int i = tid; i < CAP; i += 1024) tab[i] = EMPTY;
    for (int i = tid; i < Fe * P0Q; i += 1024) {
        gfix[L.present + i] = 0; gfix[L.rep + i] = 0x7fffffff; gfix[L.dmap + i] = -1;
    }
    if (tid == 0) { n_s = 0; bad_s = 0; }
    __syncthreads();
    for (int slot = tid; slot < ns; slot += 1024) {
        if (gfix[L.nmsg + slot] == 0) continue;          // only slots that send messages
        const unsigned long long key = ((unsigned long long)(unsigned)gfix[L.key_hi + slot] << 32) |
                                       (unsigned)gfix[L.key_lo + slot];
        unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) & (CAP - 1);
        for (int probe = 0; probe < CAP; ++probe) {
            if (*(volatile int*)&bad_s) break;
            // plain read first: thousands of slots share a few dozen patterns, and same-address LDS
            // atomics serialise — only the first arrivals of a pattern pay for one
            unsigned long long old = *(volatile unsigned long long*)&tab[h];
            if (old == key) break;
            if (old == EMPTY) {
                old = atomicCAS(&tab[h], EMPTY, key);
                if (old == key) break;
                if (old == EMPTY) { if (atomicAdd(&n_s, 1) >= P0Q) bad_s = 1; break; }
            }
            h = (h + 1) & (CAP - 1);
        }
    }
    __syncthreads();
    const bool bad = bad_s != 0;
    const int total = bad ? 0 : n_s;
    __syncthreads();
    if (tid == 0) n_s = 0;
    __syncthreads();
    if (!bad)
        for (int i = tid; i < CAP; i += 1024)
            if (tab[i] != EMPTY) list[atomicAdd(&n_s, 1)] = tab[i];     // unordered ...
    __syncthreads();
    if (!bad && tid < total) {                                          // ... rank sort -> ordered
        const unsigned long long k = list[tid];
        int rank = 0;
        for (int j = 0; j < total; ++j) rank += list[j] < k;
        gfix[L.qkeys + 2 * rank] = (int)(unsigned)(k & 0xffffffffull);
        gfix[L.qkeys + 2 * rank + 1] = (int)(unsigned)(k >> 32);
    }
    if (tid == 0) {
        gfix[L.counts + CNT_Q] = total;
        if (bad) gfix[L.counts + CNT_P0BAD] = 1;
    }
}

// ---- scan -----------------------------------------------------------------------------------
// One 1024-thread workgroup per scanned array (3 + 2 Fe arrays, independent), chunks of 1024
// elements: coalesced load, wave-shuffle inclusive scan + 16 wave totals through LDS, running carry,
// coalesced store.  Totals go straight to counts[]; a second small kernel builds the compact-row
// views that depend on several of the scans.
__global__ __launch_bounds__(1024) void compact_scan_kernel(int ns, int Fe, int* __restrict__ gfix,
                                                            Lay L) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (a == 3 + 2 * Fe) {                                // the extra block: pass-0 feature classes
        p0_classes(ns, Fe, gfix, L);
        return;
    }
    const int* src; int* dst; int* total;
    if (a == 0) { src = gfix + L.active; dst = gfix + L.cidx; total = gfix + L.counts + CNT_S; }
    else if (a == 1) { src = gfix + L.rowcnt; dst = gfix + L.seg_start; total = gfix + L.counts + CNT_E; }
    else if (a == 2) { src = gfix + L.nmsg; dst = gfix + L.srcm_start; total = gfix + L.counts + CNT_U; }
    else if (a < 3 + Fe) { const int f = a - 3; src = gfix + L.mflag_t + f * ns;
                           dst = gfix + L.mstart_t + f * ns; total = gfix + L.counts + CNT_UT + f; }
    else { const int f = a - 3 - Fe; src = gfix + L.colcnt_t + f * ns;
           dst = gfix + L.cstart_t + f * ns; total = gfix + L.counts + CNT_ET + f; }
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < ns; base += 1024) {
        const int i = base + tid;
        const int v = (i < ns) ? src[i] : 0;
        int x = v;                                            // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const int carry = carry_s;
        if (i < ns) dst[i] = carry + woff + x - v;            // exclusive
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0 && total) *total = carry_s;
}

// compact-row views: slot_of, seg_off, src_off; inactive slots map to the zero row S; type offsets
__global__ __launch_bounds__(256) void compact_finish_kernel(int ns, int Fe, int* __restrict__ gfix,
                                                             Lay L) {
    const int S = gfix[L.counts + CNT_S], E = gfix[L.counts + CNT_E], U = gfix[L.counts + CNT_U];
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot < ns) {
        if (gfix[L.active + slot]) {
            const int c = gfix[L.cidx + slot];
            gfix[L.slot_of + c] = slot;
            gfix[L.seg_off + c] = gfix[L.seg_start + slot];
            gfix[L.src_off + c] = gfix[L.srcm_start + slot];
        } else {
            gfix[L.cidx + slot] = S;
        }
    }
    // pass-0: class of every sending slot (binary search in the sorted patterns), present (bond
    // type, class) pairs and their lowest slot (the row the class's MLP input is gathered from)
    if (slot < ns && gfix[L.nmsg + slot] > 0 && gfix[L.counts + CNT_P0BAD] == 0) {
        const int Q = gfix[L.counts + CNT_Q];
        const unsigned long long key = ((unsigned long long)(unsigned)gfix[L.key_hi + slot] << 32) |
                                       (unsigned)gfix[L.key_lo + slot];
        int lo = 0, hi = Q;                               // first pattern >= key
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const unsigned long long k = ((unsigned long long)(unsigned)gfix[L.qkeys + 2 * mid + 1] << 32) |
                                         (unsigned)gfix[L.qkeys + 2 * mid];
            if (k < key) lo = mid + 1; else hi = mid;
        }
        gfix[L.cls + slot] = lo;
        for (int f = 0; f < Fe; ++f)
            if (gfix[L.mflag_t + f * ns + slot]) {
                gfix[L.present + f * P0Q + lo] = 1;
                // (the value only ever decreases: a stale read costs one redundant atomic, never a wrong minimum)
                if (*(volatile int*)&gfix[L.rep + f * P0Q + lo] > slot) atomicMin(&gfix[L.rep + f * P0Q + lo], slot);
            }
    }
    if (slot == 0) {
        gfix[L.seg_off + S] = E; gfix[L.seg_off + S + 1] = E;
        gfix[L.src_off + S] = U; gfix[L.src_off + S + 1] = U;
        int run = 0, erun = 0;                          // message-row and edge prefixes per bond type
        for (int f = 0; f < Fe; ++f) {
            gfix[L.type_off + f] = run; run += gfix[L.counts + CNT_UT + f];
            gfix[L.etype_off + f] = erun; erun += gfix[L.counts + CNT_ET + f];
        }
        for (int f = Fe; f <= GI_MAX_GROUPS; ++f) { gfix[L.type_off + f] = run; gfix[L.etype_off + f] = erun; }
    }
}

// pass-0 rows: exclusive scan of the present (bond type, class) table in row order -> row ids,
// per-type offsets, D0.  One workgroup: wave-shuffle scan over chunks of 1024 table entries.
__global__ __launch_bounds__(1024) void compact_p0_kernel(int Fe, int* __restrict__ gfix, Lay L) {
    __shared__ int wsum[16];
    __shared__ int toff[GI_MAX_GROUPS + 1];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, n = Fe * P0Q;
    const bool bad = gfix[L.counts + CNT_P0BAD] != 0 || gfix[L.counts + CNT_E] == 0;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = (i < n && !bad) ? gfix[L.present + i] : 0;
        int x = v;                                            // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const int carry = carry_s;
        const int excl = carry + woff + x - v;
        if (i < n) {
            gfix[L.dmap + i] = v ? excl : -1;
            if (v) {
                gfix[L.d_slot + excl] = gfix[L.rep + i];
                const int q = i % P0Q;
                *reinterpret_cast<int4*>(gfix + L.d_key + 4 * excl) =
                    make_int4(i / P0Q, gfix[L.qkeys + 2 * q], gfix[L.qkeys + 2 * q + 1], 0);
            }
            if (i % P0Q == 0) toff[i / P0Q] = excl;           // first row of bond type i / P0Q
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    const int total = carry_s;
    if (tid == 0) gfix[L.counts + CNT_D0] = total;
    if (tid <= GI_MAX_GROUPS) gfix[L.type_off0 + tid] = tid < Fe ? toff[tid] : total;
}

// ---- fill -----------------------------------------------------------------------------------
// NMAX = compile-time bound of N for the two LDS tables (5 KB at N <= 32 instead of the 80 KB a
// single GI_MAX_NODES-sized instance reserved: that allowed two workgroups per CU on a kernel that
// sits on the critical path of every forward).
template <typename T, int NMAX>
__global__ __launch_bounds__(256) void compact_fill_kernel(
    const T* __restrict__ nodes, int N, int Fn, int Fe, const int* __restrict__ gfix, Lay L,
    int S_in, int E_in, int U_in, int* __restrict__ u_src, int* __restrict__ in_perm,
    int* __restrict__ mu_off, int* __restrict__ mu_dst, int* __restrict__ mu_slot,
    int* __restrict__ out_perm, float* __restrict__ hx0, int ldhx, int H, int D0_in,
    int* __restrict__ d_src, float* __restrict__ cmat, int ldc0, int* __restrict__ e2d) {
    // S_in < 0: bounded mode — the sizes are read from the counts on the device (gi_compact_bound has
    // checked them against the bounds the buffers were sized with; after an overflow they are all 0 and
    // only the zero rows are written)
    const bool bounded = S_in < 0;
    const int S = bounded ? gfix[L.counts + CNT_S] : S_in, E = bounded ? gfix[L.counts + CNT_E] : E_in;
    const int U = bounded ? gfix[L.counts + CNT_U] : U_in, D0 = bounded ? gfix[L.counts + CNT_D0] : D0_in;
    if (bounded && (gfix[L.counts + CNT_ERR] & 6)) {
        if (blockIdx.x == 0) {
            for (int col = threadIdx.x; col < ldhx; col += 256) hx0[col] = 0.f;     // row S = row 0
            if (threadIdx.x == 0) mu_off[0] = 0;
        }
        return;
    }
    __shared__ signed char typ[NMAX * NMAX];
    __shared__ int kpos[NMAX * NMAX];                    // dst-CSR slot of edge (i <- j)
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NN = N * N, ns = gridDim.x * N;
    const signed char* etype_g =
        reinterpret_cast<const signed char*>(gfix + L.etype) + (long long)b * NN;
    // Adjacency as bit masks (N <= 128 = two 64-bit words): rowmask[i] = sources j of edges into i,
    // colmask[t][j] = destinations i of j's type-t edges.  A rank inside a row / column is then one
    // popcount instead of a loop over up to N table entries (N = 88: 161 -> 40 us per launch).
    __shared__ unsigned long long rowmask[GI_MAX_GROUPS][NMAX][2];   // [t][i]: sources j of i's incoming type-t edges
    __shared__ unsigned long long colmask[GI_MAX_GROUPS][NMAX][2];
    // per-slot scalars of this graph, read once (the cell loops below used to fetch them from global
    // memory per cell, one dependent load after another)
    __shared__ int seg_s[NMAX], cidx_s[NMAX], cls_s[NMAX];
    __shared__ int mstart_s[GI_MAX_GROUPS][NMAX], cstart_s[GI_MAX_GROUPS][NMAX];
    __shared__ int toff_s[GI_MAX_GROUPS], etoff_s[GI_MAX_GROUPS];
    for (int i = tid; i < N; i += 256) {
        seg_s[i] = gfix[L.seg_start + b * N + i];
        cidx_s[i] = gfix[L.cidx + b * N + i];
        cls_s[i] = D0 > 0 ? gfix[L.cls + b * N + i] : 0;
    }
    for (int idx = tid; idx < Fe * N; idx += 256) {
        const int t = idx / N, i = idx - t * N;
        mstart_s[t][i] = gfix[L.mstart_t + t * ns + b * N + i];
        cstart_s[t][i] = gfix[L.cstart_t + t * ns + b * N + i];
    }
    if (tid < Fe) { toff_s[tid] = gfix[L.type_off + tid]; etoff_s[tid] = gfix[L.etype_off + tid]; }
    for (int idx = tid; idx < GI_MAX_GROUPS * NMAX * 2; idx += 256) {
        (&rowmask[0][0][0])[idx] = 0ull; (&colmask[0][0][0])[idx] = 0ull;
    }
    for (int idx = tid; idx < NN; idx += 256) typ[idx] = etype_g[idx];
    __syncthreads();
    for (int idx = tid; idx < NN; idx += 256) {          // (a cell = the bit mask of its bond types, see compact_count)
        const int m = (unsigned char)typ[idx];
        if (!m) continue;
        const int i = idx / N, j = idx - i * N;
        for (int t = 0; t < Fe; ++t)
            if ((m >> t) & 1) {
                atomicOr(&rowmask[t][i][j >> 6], 1ull << (j & 63));
                atomicOr(&colmask[t][j][i >> 6], 1ull << (i & 63));
            }
    }
    __syncthreads();
    auto rank128 = [](const unsigned long long* m, int pos) {       // set bits below position pos
        return pos < 64 ? __popcll(m[0] & ((1ull << pos) - 1ull))
                        : __popcll(m[0]) + __popcll(m[1] & ((1ull << (pos - 64)) - 1ull));
    };
    // nd: no de-duplication (AlphaDropout training mode) — one message row per EDGE, ordered bond type,
    // source slot, destination; the message CSR is then the identity (mu_off[u] = u)
    const bool nd = gfix[L.counts + CNT_NODEDUP] != 0;
    // One thread per adjacency cell (i <- j): positions come from ranks inside the LDS type table
    // (<= N reads), so nothing below is a serial per-slot loop over global memory.
    if (D0 > 0) {                                        // pass-0 edge-count matrix: zero first
        if (b == 0) {
            for (int d = tid; d < D0; d += 256) d_src[d] = gfix[L.cidx + gfix[L.d_slot + d]];
            for (int d = tid; d < ldc0; d += 256) cmat[(long long)S * ldc0 + d] = 0.f;
        }
        const int zc = bounded ? min(ldc0, (D0 + 3) & ~3) : ldc0;   // (bounded: ldc0 is sized for the class bound)
        for (int idx = tid; idx < N * zc; idx += 256) {
            const int i = idx / zc;
            if (gfix[L.active + b * N + i])
                cmat[(long long)gfix[L.cidx + b * N + i] * ldc0 + (idx - i * zc)] = 0.f;
        }
    }
    for (int idx = tid; idx < NN; idx += 256) {          // dst-CSR: edges into i, (j, type) ascending
        const int m = (unsigned char)typ[idx];
        if (!m) continue;
        const int i = idx / N, j = idx - i * N;
        int base = seg_s[i];                             // + edges into i from sources below j
        for (int t = 0; t < Fe; ++t) base += rank128(rowmask[t][i], j);
        kpos[idx] = base;                                // the cell's first edge slot; its types follow in order
        int k = 0;
        for (int t = 0; t < Fe; ++t)
            if ((m >> t) & 1) {
                // rank of this edge among j's type-t out-edges
                const int urank = nd ? rank128(colmask[t][j], i) : 0;
                in_perm[base + k] = toff_s[t] + mstart_s[t][j] + urank;
                ++k;
            }
    }
    __syncthreads();                                     // kpos complete, cmat rows zeroed
    for (int idx = tid; idx < NN; idx += 256) {          // message CSR: edges out of j of type t, i ascending
        const int m = (unsigned char)typ[idx];
        if (!m) continue;
        const int i = idx / N, j = idx - i * N;
        const int slot = b * N + j;
        int k = 0;
        for (int t = 0; t < Fe; ++t) {
            if (!((m >> t) & 1)) continue;
            const int ed = kpos[idx] + k;
            ++k;
            const int rank = rank128(colmask[t][j], i);
            const int mo = etoff_s[t] + cstart_s[t][j] + rank;
            mu_dst[mo] = cidx_s[i];
            mu_slot[mo] = ed;
            if (nd) {                                    // this edge's own message row
                const int u = toff_s[t] + mstart_s[t][j] + rank;
                int before = 0;                          // rows of lower bond types sent by the slot
                for (int tt = 0; tt < t; ++tt) before += gfix[L.colcnt_t + tt * ns + slot];
                u_src[u] = cidx_s[j];
                out_perm[gfix[L.srcm_start + slot] + before + rank] = u;
                mu_off[u] = mo;
            }
            if (D0 > 0) {  // counts are small integers: float atomics are exact and order-independent
                const int d = gfix[L.dmap + t * P0Q + cls_s[j]];
                atomicAdd(cmat + (long long)cidx_s[i] * ldc0 + d, 1.f);
                if (e2d) e2d[ed] = d;                    // dst-CSR edge slot -> pass-0 row
            }
        }
    }
    for (int idx = tid; idx < N * Fe; idx += 256) {      // message rows of source slot j, by type
        const int j = idx / Fe, t = idx - j * Fe;
        const int slot = b * N + j;
        if (nd || gfix[L.colcnt_t + t * ns + slot] == 0) continue;
        int rank = 0;
        for (int tt = 0; tt < t; ++tt) rank += gfix[L.colcnt_t + tt * ns + slot] > 0;
        const int u = gfix[L.type_off + t] + gfix[L.mstart_t + t * ns + slot];
        u_src[u] = gfix[L.cidx + slot];
        out_perm[gfix[L.srcm_start + slot] + rank] = u;
        mu_off[u] = gfix[L.etype_off + t] + gfix[L.cstart_t + t * ns + slot];
    }
    if (b == 0 && tid == 0) mu_off[U] = E;
    // initial node rows: hx0[c] = [x, 0 .. 0 | x]  (:121-126 zero-padded hidden state; the copy of
    // the raw features at columns [H, H+Fn) feeds the gather attention MLP, gnn/modules.py:45)
    for (int idx = tid; idx < N * ldhx; idx += 256) {
        const int i = idx / ldhx, col = idx - i * ldhx;
        const int slot = b * N + i;
        if (!gfix[L.active + slot]) continue;
        const int c = gfix[L.cidx + slot];
        float v = 0.f;
        if (col < Fn) v = (float)nodes[(long long)slot * Fn + col];
        else if (col >= H && col < H + Fn) v = (float)nodes[(long long)slot * Fn + col - H];
        hx0[(long long)c * ldhx + col] = v;
    }
    if (b == 0)
        for (int col = tid; col < ldhx; col += 256) hx0[(long long)S * ldhx + col] = 0.f;
}

// ---- pass-0 row -> edge slots CSR (AttentionGGNN's pass-0 backward) --------------------------------
// One workgroup per pass-0 row d scans e2d[0..E) twice: first it counts the slots of the rows below d
// (its offset) and its own, then it writes its own slots in ascending order — a stable counting sort
// without atomics or a grid-wide sync, so the summation order of the backward is fixed.  D0 is a few
// dozen to ~100 rows and E a few 10^4 slots: the scans are L2 hits.
__global__ __launch_bounds__(1024) void class_csr_kernel(const int* __restrict__ e2d, int E, int D0,
                                                         int* __restrict__ cls_off,
                                                         int* __restrict__ cls_edges) {
    __shared__ int wbelow[16], wown[16];
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // (1) slots of the rows below d = this row's offset (coalesced count)
    int below = 0;
    for (int k = tid; k < E; k += 1024) below += e2d[k] < d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
    // (2) every thread owns a contiguous chunk of the slots: its matches, an exclusive scan over the
    // threads (one barrier), then it writes them in ascending order — stable, no per-round barriers
    const int C = (E + 1023) / 1024, lo = min(tid * C, E), hi = min(lo + C, E);
    int own = 0;
    for (int k = lo; k < hi; ++k) own += e2d[k] == d;
    int x = own;                                              // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 0) wbelow[wid] = below;
    if (lane == 63) wown[wid] = x;
    __syncthreads();
    int base = 0, woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        base += wbelow[w];
        woff += (w < wid) ? wown[w] : 0;
        total += wown[w];
    }
    int pos = base + woff + x - own;
    for (int k = lo; k < hi; ++k)
        if (e2d[k] == d) cls_edges[pos++] = k;
    if (tid == 0) {
        cls_off[d] = base;
        if (d == D0 - 1) cls_off[D0] = base + total;
    }
}


// ---- pass-0 row cache (inference loops, gi_graph.p0_cache) --------------------------------------------
// A pass-0 row is a function of (bond type, 0/1 feature pattern of the source node) and the weights; the
// table below keeps every such row an inference loop has computed since the weights last changed (the
// caller zero-fills the buffer then).  Layout (4-byte words): hdr[8] = {hit, rows, forwards, hits, ..},
// idx[PC_DMAX] = table row of each pass-0 row of the CURRENT batch (-1: absent), hash[PC_HCAP] x
// {pattern lo, pattern hi, bond type, row + 1} (open addressing, row + 1 == 0: free), rows[PC_CAP][row_floats].
constexpr int PC_HDR = 8, PC_DMAX = GI_MAX_GROUPS * P0Q, PC_CAP = 2 * PC_DMAX, PC_HCAP = 4 * PC_CAP;
struct P0Cache { int* hdr; int* idx; int* hash; float* rows; };
__host__ __device__ inline P0Cache p0_cache_view(int* c) {
    return {c, c + PC_HDR, c + PC_HDR + PC_DMAX, reinterpret_cast<float*>(c + PC_HDR + PC_DMAX + 4 * PC_HCAP)};
}
__device__ inline unsigned p0_hash(int klo, int khi, int t) {
    const unsigned long long key = ((unsigned long long)(unsigned)khi << 32) | (unsigned)klo;
    return (unsigned)(((key + (unsigned long long)t * 0x632BE59BD9B4E019ull) * 0x9E3779B97F4A7C15ull) >> 40) &
           (PC_HCAP - 1);
}
// (bond type, pattern) of pass-0 row d (written next to the row ids by compact_p0_kernel)
__device__ inline void p0_row_key(const int* __restrict__ gfix, const Lay& L, int Fe, int d, int& t, int& klo,
                                  int& khi) {
    const int4 k = *reinterpret_cast<const int4*>(gfix + L.d_key + 4 * d);
    t = k.x; klo = k.y; khi = k.z;
}
__device__ inline int p0_rows_now(const int* __restrict__ gfix, const Lay& L) {
    return (gfix[L.counts + CNT_ERR] & 6) ? 0 : gfix[L.counts + CNT_D0];   // (a bounded forward that overflowed: none)
}

__global__ __launch_bounds__(256) void p0_lookup_kernel(const int* __restrict__ gfix, Lay L, int Fe,
                                                        int* __restrict__ cache, int nfam,
                                                        float* __restrict__ m0, float* __restrict__ e0, int ldm) {
    const P0Cache c = p0_cache_view(cache);
    __shared__ int miss_s;
    const int tid = threadIdx.x;
    const int D0 = p0_rows_now(gfix, L);
    if (tid == 0) miss_s = 0;
    __syncthreads();
    for (int d = tid; d < D0; d += 256) {
        int t, klo, khi, row = -1;
        p0_row_key(gfix, L, Fe, d, t, klo, khi);
        unsigned h = p0_hash(klo, khi, t);
        for (int probe = 0; probe < PC_HCAP; ++probe) {
            const int4 e = reinterpret_cast<const int4*>(c.hash)[h];
            if (e.w == 0) break;
            if (e.x == klo && e.y == khi && e.z == t) { row = e.w - 1; break; }
            h = (h + 1) & (PC_HCAP - 1);
        }
        c.idx[d] = row;
        if (row < 0) miss_s = 1;
    }
    __syncthreads();
    const int hit = (D0 > 0 && miss_s == 0) ? 1 : 0;
    if (tid == 0) { c.hdr[0] = hit; c.hdr[2] += 1; c.hdr[3] += hit; }
    if (!hit) return;
    const int q = ldm >> 2, rowq = nfam * q;                      // float4 per family row / per table row
    for (int i = tid; i < D0 * rowq; i += 256) {
        const int d = i / rowq, j = i - d * rowq;
        const float4 v = reinterpret_cast<const float4*>(c.rows)[(long long)c.idx[d] * rowq + j];
        float* dst = (j < q ? m0 : e0) + (long long)d * ldm;
        reinterpret_cast<float4*>(dst)[j < q ? j : j - q] = v;
    }
}

__global__ __launch_bounds__(256) void p0_insert_kernel(const int* __restrict__ gfix, Lay L, int Fe,
                                                        int* __restrict__ cache, int nfam,
                                                        const float* __restrict__ m0,
                                                        const float* __restrict__ e0, int ldm) {
    const P0Cache c = p0_cache_view(cache);
    const int tid = threadIdx.x;
    const int D0 = p0_rows_now(gfix, L);
    if (c.hdr[0] != 0 || D0 == 0) return;                        // served from the table / nothing computed
    const bool clear = c.hdr[1] + D0 > PC_CAP;                    // would not fit: start over with this batch
    __syncthreads();                                              // (everybody has read hdr[1])
    if (clear) {
        for (int i = tid; i < 4 * PC_HCAP; i += 256) c.hash[i] = 0;
        if (tid == 0) c.hdr[1] = 0;
        __threadfence();
        __syncthreads();
    }
    for (int d = tid; d < D0; d += 256) {
        if (!clear && c.idx[d] >= 0) continue;
        int t, klo, khi;
        p0_row_key(gfix, L, Fe, d, t, klo, khi);
        const int row = atomicAdd(&c.hdr[1], 1);
        unsigned h = p0_hash(klo, khi, t);
        for (;;) {                                                // (bond type, pattern) pairs of a batch are distinct
            if (atomicCAS(&c.hash[4 * h + 3], 0, row + 1) == 0) {
                c.hash[4 * h] = klo; c.hash[4 * h + 1] = khi; c.hash[4 * h + 2] = t;
                break;
            }
            h = (h + 1) & (PC_HCAP - 1);
        }
        c.idx[d] = row;
    }
    __syncthreads();
    const int q = ldm >> 2, rowq = nfam * q;
    for (int i = tid; i < D0 * rowq; i += 256) {                  // (rows already in the table: same values again)
        const int d = i / rowq, j = i - d * rowq;
        const float* src = (j < q ? m0 : e0) + (long long)d * ldm;
        reinterpret_cast<float4*>(c.rows)[(long long)c.idx[d] * rowq + j] =
            reinterpret_cast<const float4*>(src)[j < q ? j : j - q];
    }
}

}  // namespace

long long gi_p0_cache_words_for(int row_floats) {
    if (row_floats <= 0 || (row_floats & 3)) return GI_EINVAL;
    return (long long)PC_HDR + PC_DMAX + 4LL * PC_HCAP + (long long)PC_CAP * row_floats;
}

static int p0_cache_args(const int* gfix, int B, int N, int Fe, const int* cache, int nfam, const float* m0,
                         const float* e0, int ldm, Lay& L) {
    if (!gfix || !cache || !m0 || (nfam == 2 && !e0) || nfam < 1 || nfam > 2 || ldm <= 0 || (ldm & 3))
        return GI_EINVAL;
    if (B <= 0 || N <= 0 || Fe <= 0 || N > GI_MAX_NODES || Fe > GI_MAX_GROUPS) return GI_EINVAL;
    L = make_layout(B, N, Fe);
    return 0;
}

int gi_p0_cache_lookup(const int* gfix, int B, int N, int Fe, int* cache, int nfam, float* m0, float* e0, int ldm,
                       void* stream) {
    Lay L;
    if (int rc = p0_cache_args(gfix, B, N, Fe, cache, nfam, m0, e0, ldm, L)) return rc;
    hipLaunchKernelGGL(p0_lookup_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gfix, L, Fe, cache, nfam, m0,
                       e0, ldm);
    return gi_launch_status();
}

int gi_p0_cache_insert(const int* gfix, int B, int N, int Fe, int* cache, int nfam, const float* m0,
                       const float* e0, int ldm, void* stream) {
    Lay L;
    if (int rc = p0_cache_args(gfix, B, N, Fe, cache, nfam, m0, e0, ldm, L)) return rc;
    hipLaunchKernelGGL(p0_insert_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gfix, L, Fe, cache, nfam, m0,
                       e0, ldm);
    return gi_launch_status();
}

extern "C" int gi_compact_class_csr(const int* e2d, int E, int D0, int* cls_off, int* cls_edges,
                                    void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (D0 <= 0 || E <= 0) return 0;
    if (!e2d || !cls_off || !cls_edges) return GI_EINVAL;
    hipLaunchKernelGGL(class_csr_kernel, dim3(D0), dim3(1024), 0, (hipStream_t)stream, e2d, E, D0,
                       cls_off, cls_edges);
    return gi_launch_status();
}

extern "C" int gi_abi_version(void) { return GI_ABI_VERSION; }

extern "C" int gi_compact_layout(int B, int N, int Fe, gi_compact_layout_t* out) {
    if (!out || B <= 0 || N <= 0 || Fe <= 0) return GI_EINVAL;
    if (N > GI_MAX_NODES || Fe > GI_MAX_GROUPS) return GI_ELIMIT;
    if ((long long)B * N * N > 0x7fffffffLL) return GI_ELIMIT;
    const Lay L = make_layout(B, N, Fe);
    out->total_ints = L.total;
    out->counts = L.counts; out->type_off = L.type_off; out->cidx = L.cidx;
    out->node_mask = L.node_mask; out->slot_of = L.slot_of; out->seg_off = L.seg_off;
    out->src_off = L.src_off; out->type_off0 = L.type_off0; out->scratch = L.scratch;
    out->dims = L.dims;
    return 0;
}

extern "C" int gi_compact_count(const void* nodes, const void* edges, int in_dtype, int B, int N,
                                int Fn, int Fe, int* gfix, void* stream) {
    return gi_compact_count_ex(nodes, edges, in_dtype, B, N, Fn, Fe, gfix, 0, stream);
}

extern "C" int gi_compact_count_ex(const void* nodes, const void* edges, int in_dtype, int B, int N,
                                   int Fn, int Fe, int* gfix, int nodedup, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (!nodes || !edges || !gfix || B <= 0 || N <= 0 || Fn <= 0 || Fe <= 0) return GI_EINVAL;
    if (in_dtype != GI_DTYPE_F32 && in_dtype != GI_DTYPE_I8) return GI_EINVAL;
    if (N > GI_MAX_NODES || Fe > GI_MAX_GROUPS) return GI_ELIMIT;
    const Lay L = make_layout(B, N, Fe);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(gfix + L.counts, 0, CNT_N * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    if (in_dtype == GI_DTYPE_F32)
        hipLaunchKernelGGL(compact_count_kernel<float>, dim3(B), dim3(256), 0, st,
                           (const float*)nodes, (const float*)edges, N, Fn, Fe, gfix, L,
                           nodedup ? 1 : 0);
    else
        hipLaunchKernelGGL(compact_count_kernel<signed char>, dim3(B), dim3(256), 0, st,
                           (const signed char*)nodes, (const signed char*)edges, N, Fn, Fe, gfix, L,
                           nodedup ? 1 : 0);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(3 + 2 * Fe + 1), dim3(1024), 0, st, B * N, Fe, gfix,
                       L);
    hipLaunchKernelGGL(compact_finish_kernel, dim3(gi_cdiv(B * N, 256)), dim3(256), 0, st, B * N, Fe,
                       gfix, L);
    hipLaunchKernelGGL(compact_p0_kernel, dim3(1), dim3(1024), 0, st, Fe, gfix, L);
    return gi_launch_status();
}

// ---- bounded (host-sync-free) forward: sizes stay on the device -----------------------------------------
// One thread checks the counts against the bounds the caller sized its buffers with and derives the launch
// dimensions the forward's kernels read from the device (gfix + layout.dims):
//   dims[0] = R = S + 1 (compact node rows incl. the zero row), dims[1] = row-block height of the message
//   chains (32..36: the height that saves a round of workgroups on `ncu` CUs, like the host-side choice),
//   dims[2] = 32 (row-block height of the pass-0 chains).
// Overflow (E > e_bound or D0 > d0_bound: counts[2] |= 2) and a batch whose pass-0 shortcut is unavailable
// although it has edges (non-0/1 node features or > GI_P0_MAX_CLASSES classes: counts[2] |= 4) cannot run
// bounded: every size is set to 0, the forward then touches nothing beyond its buffers and returns garbage
// logits; the caller reads counts[2] when it next synchronises.
__global__ void compact_bound_kernel(int Fe, int* __restrict__ gfix, Lay L, int e_bound, int d0_bound, int ncu,
                                     int* __restrict__ sticky_err) {
    if (threadIdx.x || blockIdx.x) return;
    int* c = gfix + L.counts;
    int* d = gfix + L.dims;
    const bool over = c[CNT_E] > e_bound || c[CNT_D0] > d0_bound;
    const bool p0bad = c[CNT_P0BAD] != 0 && c[CNT_E] > 0;
    if (over || p0bad) {
        c[CNT_ERR] |= over ? 2 : 4;
        c[CNT_S] = c[CNT_E] = c[CNT_U] = c[CNT_D0] = 0;
        for (int t = 0; t < GI_MAX_GROUPS; ++t) { c[CNT_UT + t] = 0; c[CNT_ET + t] = 0; }
        for (int t = 0; t <= GI_MAX_GROUPS; ++t) { gfix[L.type_off + t] = 0; gfix[L.type_off0 + t] = 0; }
        // the one compact row that is left (R = 1: the zero row) has no incoming edges and sends nothing —
        // the edge index arrays are NOT filled after an overflow, nothing may be read through them
        gfix[L.seg_off] = gfix[L.seg_off + 1] = 0;
        gfix[L.src_off] = gfix[L.src_off + 1] = 0;
    }
    // the error word of THIS forward's gfix dies with it; a caller that checks once after a whole generation loop
    // (GraphGenerator.build_graphs: hundreds of forwards) needs every round's bits: OR them into its accumulator
    if (sticky_err && c[CNT_ERR]) atomicOr(sticky_err, c[CNT_ERR]);
    d[0] = c[CNT_S] + 1;
    auto blocks = [&](int h) { int n = 0; for (int t = 0; t < Fe; ++t) n += (c[CNT_UT + t] + h - 1) / h; return n; };
    int h = 32, rounds = (blocks(32) + ncu - 1) / ncu;
    if (rounds > 1)
        for (int hh = 33; hh <= 36; ++hh)
            if ((blocks(hh) + ncu - 1) / ncu < rounds) { h = hh; break; }
    d[1] = h;
    d[2] = 32;                                   // row-block height of the pass-0 chains (a few dozen rows)
    for (int i = 3; i < GI_DIMS; ++i) d[i] = 0;
}

extern "C" int gi_compact_bound(int* gfix, int B, int N, int Fe, int e_bound, int d0_bound, int* sticky_err,
                                void* stream) {
    (void)hipGetLastError();
    if (!gfix || B <= 0 || N <= 0 || Fe <= 0 || e_bound < 0 || d0_bound < 0) return GI_EINVAL;
    if (N > GI_MAX_NODES || Fe > GI_MAX_GROUPS) return GI_ELIMIT;
    const Lay L = make_layout(B, N, Fe);
    static const int ncu = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    hipLaunchKernelGGL(compact_bound_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, Fe, gfix, L, e_bound,
                       d0_bound, ncu, sticky_err);
    return gi_launch_status();
}

extern "C" int gi_compact_fill(const void* nodes, int in_dtype, int B, int N, int Fn, int Fe,
                               const int* gfix, int S, int E, int U, int* u_src, int* in_perm,
                               int* mu_off, int* mu_dst, int* mu_slot, int* out_perm, float* hx0,
                               int ldhx, int H, int D0, int* d_src, float* cmat, int ldc0, int* e2d,
                               void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (!nodes || !gfix || !hx0 || !mu_off || B <= 0 || N <= 0) return GI_EINVAL;
    const bool bounded = S < 0;            // sizes on the device; E, U, D0 are then the BOUNDS of the buffers
    if (!bounded && (E < 0 || U < 0 || U > E)) return GI_EINVAL;
    if (bounded && (E < 0 || U < 0)) return GI_EINVAL;
    if (E > 0 && (!u_src || !in_perm || !mu_dst || !mu_slot || !out_perm)) return GI_EINVAL;
    if (D0 < 0 || D0 > GI_MAX_GROUPS * GI_P0_MAX_CLASSES) return GI_EINVAL;
    if (D0 > 0 && (!d_src || !cmat || ldc0 < D0 || (ldc0 & 3))) return GI_EINVAL;
    if (N > GI_MAX_NODES || Fe > GI_MAX_GROUPS) return GI_ELIMIT;
    if (ldhx < H + Fn || Fn > H) return GI_EINVAL;
    const Lay L = make_layout(B, N, Fe);
#define GI_FILL(T_, NMAX_)                                                                         \
    hipLaunchKernelGGL((compact_fill_kernel<T_, NMAX_>), dim3(B), dim3(256), 0, (hipStream_t)stream, \
                       (const T_*)nodes, N, Fn, Fe, gfix, L, bounded ? -1 : S, E, U, u_src, in_perm,  \
                       mu_off, mu_dst, mu_slot, out_perm, hx0, ldhx, H, D0, d_src, cmat, ldc0, e2d)
#define GI_FILL_N(T_)                                                                              \
    do {                                                                                           \
        if (N <= 32) GI_FILL(T_, 32);                                                              \
        else if (N <= 64) GI_FILL(T_, 64);                                                         \
        else GI_FILL(T_, GI_MAX_NODES);                                                            \
    } while (0)
    if (in_dtype == GI_DTYPE_F32) GI_FILL_N(float);
    else if (in_dtype == GI_DTYPE_I8) GI_FILL_N(signed char);
    else
        return GI_EINVAL;
#undef GI_FILL_N
#undef GI_FILL
    return gi_launch_status();
}
