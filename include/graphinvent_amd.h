/*
 * graphinvent_amd.h — C ABI of libgraphinvent_amd.so (gfx950 / MI355X only).
 *
 * Drop-in boundary.  The reference has NO native layer: its GGNN hot path
 * (graphinvent/gnn/summation_mpnn.py:80-149, gnn/mpnn.py:229-303, gnn/modules.py:12-52,111-281) is
 * stock PyTorch called as `model(nodes, edges)` from Workflow.py:785,826, GraphGenerator.py:121,
 * GraphGeneratorRL.py:131-132 and Analyzer.py:759.  The Python boundary therefore stays the
 * reference's own — `gnn.mpnn.GGNN(constants).forward(nodes, edges)` — and this header is what
 * that class binds underneath (ctypes, graphinvent_amd/lib.py).  Each entry point names the
 * reference statement(s) it replaces.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller (torch tensors passed by data_ptr());
 *     nothing here allocates or frees; workspaces are caller provided;
 *   - matrices are fp32 row-major with an explicit leading dimension in floats; index arrays
 *     are int32;
 *   - `stream` is a hipStream_t; all work is enqueued asynchronously on it;
 *   - return value: 0 = success, >0 = hipError_t from a launch, <0 = argument error (GI_E*);
 *   - no exceptions cross the ABI; one process per GPU, calls come from one host thread at a time.
 *     Process-wide state is limited to: the optional per-launch timing log (gi_prof_*), a pool of
 *     timing-disabled hipEvents used to order the backward's two streams, the test / measurement hooks
 *     gi_gemm_config and gi_mlp_chain_config, and switches read once from the environment:
 *       GI_FUSE=<mask>          launch-count reductions (gi_fuse_flags, default 15)
 *       GI_CHAIN=0              per-bond-type stacks layer by layer through gi_gemm instead of gi_mlp_chain
 *       GI_BF3=0                every GEMM and chain on the fp32 MFMA (default 1: the node-level readout layers >= 192
 *                               wide and the message stacks' chains, forward and dZ, run as splits on the 16-bit MFMA
 *                               pipe, GI_GEMM_BF3 — same result to ~3e-7; gi_bf3_enable)
 *       GI_X2=0                 those launches as three bf16 planes (six products) instead of two scaled fp16 planes
 *                               (three products, GI_GEMM_X2; gi_x2_enable); also puts the chains back on fp32
 *       GI_CHAIN_X2=0           only the dZ chains back on the fp32 chain kernel (gi_chain_params.x2_wamax unused)
 *       GI_CHAIN_FWD_X2=0       only the FORWARD chains back on the fp32 chain kernel (default: the row-independent
 *                               fp16x2 kernel, gi_chain_params.x2_rows32)
 *       GI_WGRAD_X2_ALL=1       (round 6; measured slower, off by default) EVERY weight gradient of at least 32 x 32 outputs
 *                               over >= 512 rows as an fp16x2 launch; operands nobody publishes a maximum of get their amax
 *                               cell from gi_absmax in front of the launch.  Default: the round-4 set + the message / energy
 *                               stacks' hidden layers (cells published by the fp16x2 chain kernels)
 *                               GI_WGRAD_T128=0 / 1 / 2: none (default) / the round-6 set / all fp16x2 weight gradients on the
 *                               128 x 128-tile kernel of gi_gemm_b3v.hip (GI_GEMM_T128) instead of the pipelined 128 x 256 one
 *       GI_MSG_WGRAD_X2=0       the fp16x2 chain kernels' amax cells are not used for the message / energy stacks' weight
 *                               gradients (their first, gathered layer then stays on the fp32 MFMA);
 *                               GI_MSG_SLAB_ROWS=<n>: reduction rows per split-K slab of the new launches (default 460)
 *       GI_GRU_PAD_LDS=0        (measurement aid) the fused GRU launch without the unused dynamic LDS that keeps it at one
 *                               workgroup per CU
 *       GI_GRU_FUSED=0          the forward's GRU update as a two-projection GEMM launch + the gate kernel (default, round 6:
 *                               one fused launch, csrc/gi_gru.hip)
 *       GI_CHAIN_PACK_FUSED=0   the fp16x2 chain image as memset + gi_absmax + pack launches (default, round 6: one launch that
 *                               also writes the max |W| cells)
 *       GI_P0_GRU_MAIN=0        pass 0's GRU weight gradients go to the weight-gradient queue with everything else (default,
 *                               round 6: they ride in the main queue's last launch, so that both queues end together)
 *       GI_GEMM_LOG=<file>      one line per GEMM launch (tools/gemm_launch_report.py)
 *     and measurement aids that pick between kernels / schedules that compute the same thing (the A/B files under
 *     profiles/r04 name them): GI_B3P, GI_B3V, GI_B3P_ALL, GI_B3P_STREAM, GI_B3V_GROUPED (which 16-bit-pipe kernel),
 *     GI_B3W_MSG, GI_B3W_G (message-stack / graph-level weight gradients on the 16-bit pipe below their size
 *     thresholds), GI_P0_LAYERWISE (pass 0 without the chain kernel), GI_CHAIN_BWD64 (64-row fp32 chain blocks in
 *     the backward), GI_CHAIN_XCD (0: the chain kernels' row blocks in dispatch order instead of the XCD-aware one),
 *     GI_CHAIN_X2R_DUAL (0 / 1: the row-independent fp16x2 chain never / always as two workgroups per CU; default: when a
 *     launch has more row blocks than CUs), GI_CHAIN_BWD_X2R (1: the dZ chains through that kernel too),
 *     GI_WGRAD_BIAS (1: weight gradients whose input width is a multiple of 64 get their bias gradient from a separate
 *     launch instead of a "ones" column that costs a column of tiles — measured a tie, off by default), GI_WGRAD_TN /
 *     GI_WGRAD_WGS (tile class / workgroups per problem of the fp32-MFMA weight gradients), GI_SEGSUM_U (outputs per
 *     thread of the aggregation kernel: 1, 2 or 4), GI_HOLD_KICKS (0 / 1: weight-gradient launches beside the
 *     node-level dgrad launches always / never; default: held back up to 9 000 node rows), GI_KICK_N (weight-gradient problems per hand-over to the second
 *     queue, 1..8), GI_SIDE_PRIO (0: gi_side_stream_create with a middle instead of the lowest priority), GI_CHAIN_RING3_SMALL (chain launches
 *     of at most that many row blocks on the three-slot weight ring), GI_DBG_X2 (only in a library built with -DGI_CHAIN_X2_LAB; results WRONG: parts of the fp16x2 chain kernel switched
 *     off for timing),
 *     GI_B3V_X2_FWD (1: the fp16x2 forward / dgrad launches of the node-level stacks on the 32-deep-tile kernel of gi_gemm_b3v.hip),
 *     GI_B3P_WGRAD_REMAP / GI_WGRAD_SLAB_ORDER
 *     (0: the 16-bit-pipe / fp32 weight-gradient tiles in dispatch order / per-slab order instead of slab-per-XCD); graphinvent_amd/gnn/mpnn.py reads GI_PREPACK (0: gi_ggnn_forward_ex without a side stream and without
 *     GI_RUN_PREPACK_BWD: the round-4 schedule).
 */
#ifndef GRAPHINVENT_AMD_H
#define GRAPHINVENT_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define GI_ABI_VERSION 18
#define GI_MAX_GROUPS 8       /* max bond types (n_edge_features) */
#define GI_MAX_NODES 128      /* max max_n_nodes */
#define GI_P0_MAX_CLASSES 256 /* max distinct node feature rows for the pass-0 shortcut */

#define GI_DTYPE_F32 0        /* element type of model INPUTS (nodes, edges, APD targets): fp32 ... */
#define GI_DTYPE_I8  1        /* ... or the int8 the preprocessed HDF stores (DataProcesser.py:157-161) */

#define GI_EINVAL   (-1)      /* bad dims / null pointer */
#define GI_ELIMIT   (-2)      /* exceeds a compiled-in limit */

int gi_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * K1 graph_compact — replaces the dense->sparse bookkeeping of gnn/summation_mpnn.py:100-124
 * (`edges.sum(3)`, two `nonzero`s, the [V,E] 0/1 summation matrix, `edges[eb,ei,ej,:]`, the
 * zero-padded `hidden_nodes`) and `node_mask` (:146).
 *
 * MESSAGE ROWS.  The message an edge carries, MLP_type(e)(h_src(e)) (gnn/mpnn.py:284-294), depends
 * only on (source node, bond type); it is computed once per distinct pair: U <= E rows (0.55-0.63 E
 * on molecular graphs), bond-type-major, inside a type by source slot.
 *
 * Layout of the fixed-size int32 index buffer `gfix` (offsets in ints, from gi_compact_layout):
 *   counts[GI_COUNTS]  [0]=S active node slots, [1]=E directed edges, [2]=error flag (an edge whose
 *                feature vector is not one-hot), [3]=U message rows, [4+t]=message rows of bond
 *                type t, [12+t]=edges of bond type t, [20]=D0 pass-0 rows (0 = shortcut off)
 *   type_off[GI_MAX_GROUPS+1] (message rows per bond type, prefix), cidx[B*N] (slot -> compact row,
 *   S for inactive slots), node_mask[B*N] (1 if the slot has >=1 incoming edge), slot_of[B*N],
 *   seg_off[B*N+2] (dst-CSR over compact rows: row c's incoming edges are slots
 *   [seg_off[c], seg_off[c+1]); row S empty), src_off[B*N+2] (source CSR over MESSAGE rows: row c
 *   sends out_perm[src_off[c] .. src_off[c+1])), then scratch.
 * Variable-size arrays (caller-allocated once S, E, U are known, written by gi_compact_fill):
 *   u_src[U]    message row -> source compact row (the message MLP's gather index)
 *   in_perm[E]  dst-CSR slot -> message row (edges in the reference's order: row-major nonzero)
 *   mu_off[U+1], mu_dst[E], mu_slot[E]   message row -> its edges: destination compact row and
 *               dst-CSR slot of each, ascending destination (backward of the aggregation)
 *   out_perm[U] message rows grouped by source compact row, inside a row by bond type
 * PASS-0 ROWS.  At the first message pass h = [x | 0..0], so a message row depends only on (feature
 * row of its source, bond type): D0 rows (atom type x charge x bond type; counts[20]), bond-type-
 * major with offsets type_off0[GI_MAX_GROUPS+1] in gfix.  d_src[D0] = a compact row of each class
 * (the MLP's gather index); cmat[S+1, ldc0] fp32 = number of edges into each compact row from each
 * pass-0 row, so that the pass-0 aggregation is cmat . m0 and its backward cmat^T . d agg.
 * ------------------------------------------------------------------------------------------ */
#define GI_COUNTS 24
#define GI_DIMS 8             /* device-side launch dimensions of a bounded forward (gi_compact_bound) */
typedef struct gi_compact_layout_t {
    int total_ints;
    int counts, type_off, cidx, node_mask, slot_of, seg_off, src_off, type_off0, scratch;
    int dims;
} gi_compact_layout_t;

int gi_compact_layout(int B, int N, int Fe, gi_compact_layout_t* out);

/* phase 1: per-graph counting + global scans; fills everything in gfix. */
int gi_compact_count(const void* nodes, const void* edges, int in_dtype, int B, int N, int Fn,
                     int Fe, int* gfix, void* stream);
/* nodedup != 0: NO row sharing — every slot of the [B, N] grid is a compact row of its own (S = B*N;
 * padded slots too), every edge its own message row (U = E, ordered bond type, source slot,
 * destination), no pass-0 rows (D0 = 0).  The layout AlphaDropout's training mode needs (an
 * independent mask per edge and per padded slot, gnn/modules.py:130-142); counts[23] records the
 * mode for gi_compact_fill.  nodedup == 0 is gi_compact_count. */
int gi_compact_count_ex(const void* nodes, const void* edges, int in_dtype, int B, int N, int Fn,
                        int Fe, int* gfix, int nodedup, void* stream);
/* BOUNDED (host-sync-free) forward, between phase 1 and phase 2: instead of reading S, E, U, D0 back the
 * caller sizes every buffer for bounds (S <= B*N; E, U <= e_bound; D0 <= d0_bound) and this one-thread launch
 * checks the counted sizes against them and derives the launch dimensions the forward's kernels read on the
 * device: gfix[layout.dims + 0] = R = S + 1, + 1 = row-block height of the message chains, + 2 = 32.  On a
 * violation — more edges / pass-0 rows than the bound (counts[2] |= 2), or a batch with edges whose pass-0
 * shortcut is unavailable (node features not 0/1, counts[2] |= 4) — every size is set to 0: the forward then
 * stays inside its buffers and returns meaningless logits; read counts[2] at the next synchronisation.
 * `sticky_err` (may be NULL): a caller-owned device int that accumulates (atomic OR) the error bits of every
 * bounded forward — counts[2] lives in that forward's own gfix, so a loop that checks once at its end would
 * otherwise see only the last round (bit 0 = an edge's feature vector is not one-hot, from phase 1).
 * This is what a caller that mutates `nodes` / `edges` in place between forwards
 * (GraphGenerator.build_graphs, GraphGenerator.py:118-157) needs: no device -> host wait per forward. */
int gi_compact_bound(int* gfix, int B, int N, int Fe, int e_bound, int d0_bound, int* sticky_err, void* stream);
/* phase 2 (after the host has read S, E, U from counts): the variable-size index arrays and the
 * initial node rows hx0[S+1, ldhx] = [x | 0.. | x] with the input features in columns [0,Fn) and
 * again in [H, H+Fn) (row S = 0).  S < 0: bounded mode — E, U, D0 (and the buffers) are the bounds handed to
 * gi_compact_bound, the real sizes are read from gfix on the device. */
int gi_compact_fill(const void* nodes, int in_dtype, int B, int N, int Fn, int Fe, const int* gfix,
                    int S, int E, int U, int* u_src, int* in_perm, int* mu_off, int* mu_dst,
                    int* mu_slot, int* out_perm, float* hx0, int ldhx, int H, int D0, int* d_src,
                    float* cmat, int ldc0, int* e2d, void* stream);
/* AttentionGGNN's pass 0 on the D0 class rows: e2d[E] (written by gi_compact_fill when non-NULL and
 * D0 > 0) = dst-CSR edge slot -> pass-0 row, the index the segment softmax reads its rows through;
 * gi_compact_class_csr turns it into the CSR pass-0 row -> its edge slots (cls_off[D0+1],
 * cls_edges[E], slots ascending: fixed summation order) over which the softmax backward is summed. */
int gi_compact_class_csr(const int* e2d, int E, int D0, int* cls_off, int* cls_edges, void* stream);

/* The compacted graph as the fused model calls take it. */
typedef struct gi_graph {
    int S, E, U;
    const int* gfix;
    const int* u_src; const int* in_perm; const int* mu_off; const int* mu_dst; const int* mu_slot;
    const int* out_perm;
    const int* Ut;            /* HOST array [Fe]: message rows per bond type (counts[4..4+Fe)) */
    int D0, ldc0;             /* pass-0 rows (0 = shortcut off) and leading dimension of cmat */
    const int* d_src;         /* [D0] */
    const float* cmat;        /* [S+1, ldc0] */
    const int* e2d;           /* [E]    AttentionGGNN pass 0 (NULL: that pass runs on message rows) */
    const int* cls_off;       /* [D0+1] */
    const int* cls_edges;     /* [E] */
    int bounded;              /* != 0: host-sync-free forward — S, E, U, D0, Ut[] above are upper BOUNDS (the
                                 sizes every buffer was allocated for); the real sizes stay on the device in
                                 gfix (gi_compact_bound) and every kernel of gi_ggnn_forward reads them there.
                                 Forward only (gi_ggnn_backward needs host sizes), no dropout mode. */
    void* p0_cache;           /* NULL, or gi_p0_cache_words() 4-byte words of device memory (16-byte aligned,
                                 zero-filled = empty) that OUTLIVE the call: the pass-0 row cache of an
                                 inference loop, see gi_p0_cache_words.  Forward only (the hidden activations
                                 of the pass-0 stacks are not produced on a hit). */
    int* x2_guard;            /* NULL, or GI_X2_GUARD_WORDS device ints that OUTLIVE the call: sticky counters of the
                                 fp16x2 dynamic-range guard (see GI_GEMM_X2 / gi_gemm_params.x2_guard):
                                 [0] rows of a FORWARD launch's activation operand whose largest magnitude lies more
                                     than 2^24 below the tensor's (such a row keeps fewer than ~14 bits in fp16x2),
                                 [1] rows / columns of a weight matrix of those layers in the same position,
                                 [2] the same for the dZ operands of the dgrad launches (informational: everything the
                                     backward outputs is a sum over rows, which an absolute error floor of 2^-38 of the
                                     tensor's maximum does not move),
                                 [3] reserved (stays 0). */
    int* x2_guard_host;       /* NULL, or ONE int of host memory mapped into the device (gi_host_flag_create): set to 1
                                 by the launch that increments counter [0] or [1], so that the host learns of a trip
                                 without a read-back and can run the following calls with GI_RUN_NO_X2 */
    float* wcache;            /* NULL, or gi_ggnn_wcache_floats() floats of device memory (16-byte aligned) that OUTLIVE the
                                 call: everything a FORWARD derives from the weights alone — the fp16x2 forward chains'
                                 two-plane weight image and their max |W| cells, max |W| of the node-level fp16x2 layers —
                                 kept across forwards of an inference / generation loop (round 6).  Forward only
                                 (gi_ggnn_backward does not read it: pass a graph without it to a forward that needs a tape). */
    int wcache_valid;         /* with wcache: 0 = (re)derive everything into it now (first forward, or the weights /
                                 the arithmetic switches changed since: the CALLER keeps that key), != 0 = its contents
                                 match the weights: no pack, no amax pass, no weight guard launches */
} gi_graph;
#define GI_X2_GUARD_WORDS 4
/* One int of pinned host memory that kernels can write (hipHostMalloc mapped): *host = the caller's view,
 * *dev = the pointer to hand to kernels (gi_graph.x2_guard_host).  Zeroed on creation. */
int gi_host_flag_create(int** host, int** dev);
int gi_host_flag_destroy(int* host);

/* ------------------------------------------------------------------------------------------
 * Dense GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32) with fused prologue/epilogue.  One kernel
 * family serves
 *   forward  Y = selu(X[a_idx] W^T + b)      torch.nn.Linear + SELU, gnn/modules.py:130-142,166-170
 *   dgrad    dX = (dZ W) * selu'(Xact)        autograd of the same
 *   wgrad    [dW | db] = dZ^T [X[b_idx] | 1]  autograd of the same (split over rows, slabs)
 * grouped per bond type (gnn/mpnn.py:284-294 evaluates all Fe MLPs on all edges and masks; here
 * every edge row only meets its own type's weights).
 * ------------------------------------------------------------------------------------------ */
#define GI_EPI_BIAS    1   /* v += bias[col]                                 */
#define GI_EPI_SELU    2   /* v = selu(v)                                    */
#define GI_EPI_DSELU   4   /* v *= selu'(act[row,col]) (act = selu output)   */
#define GI_EPI_ACCUM   8   /* v += C[row,col]                                */
#define GI_GEMM_SPLITK 16  /* reduction range partitioned by groups/splits; C is a slab set */
#define GI_EPI_MULACT  64  /* v *= act[row,col] (a stored factor: AlphaDropout training mode) */
#define GI_GEMM_BF3    128 /* B is a pre-split bf16 image (gi_bf3_pack); the launch runs on the bf16 MFMA pipe with
                              fp32 operands split three ways (six bf16 products per fp32 product, fp32 accumulate:
                              the same result to ~4e-7 relative, the fp32 MFMA chain's own distance from the fp64 product).  A contig fp32, no groups / split-K / a_idx / b_idx
                              (32-bit operand offsets: a gathered row is not bounded by M), K < 4 needs lda / ldb >= 4;
                              every problem of a batched launch or none */
#define GI_GEMM_BF3B_F32 512 /* with GI_GEMM_BF3: B is the plain fp32 matrix [N][ldb] (a forward weight as stored), split while
                              it is staged like A: no image, 4 bytes per element through L2 instead of 6 */
#define GI_GEMM_BF3A   256 /* with GI_GEMM_BF3: A is a pre-split bf16 image too ([3][M][Kp], gi_bf3_pack of an [M, K]
                              matrix or the `planes` output of a producing launch); no a_idx */
#define GI_GEMM_T128  2048 /* with GI_GEMM_BF3 | GI_GEMM_X2, weight-gradient layout: prefer the 128 x 128-tile kernel (256 threads,
                              32 KB of LDS, several workgroups per CU: gi_gemm_b3v.hip) to the software-pipelined
                              128 x 256-tile one (one 512-thread workgroup per CU) — launches of many small problems */
#define GI_GEMM_X2    1024 /* with GI_GEMM_BF3 and plain fp32 operands: split every operand into TWO scaled fp16 values
                              instead of three bf16 (csrc/gi_x2.h): three f16 MFMA products per fp32 product instead of
                              six, the same ~3e-7 distance from the fp64 product.  Needs the largest magnitude of both
                              operand tensors on the device (`a_amax`, `b_amax`: written by the `c_amax` of the launch
                              that produced the tensor, or by gi_absmax) */

typedef struct gi_gemm_params {
    const float* A; const float* B; float* C;
    const float* bias; const float* act;
    const int* a_idx; const int* b_idx;   /* gather on the STORED rows of A / B, or NULL */
    const int* grp_off;                   /* device [ngroups+1] or NULL */
    int M, N, K;                          /* output M x N, reduction length K */
    int lda, ldb, ldc, ldact;
    int flags;                            /* GI_EPI_* | GI_GEMM_SPLITK | GI_GEMM_BF3 */
    int a_major, b_major;                 /* operand stored [reduction][rows] instead of [rows][reduction] */
    int tm, tn;                           /* block tile = 64*tm x 64*tn, (tm,tn) in {(1,1),(1,2),(2,2)} */
    int ngroups, nsplit;                  /* ngroups 0 = ungrouped; nsplit >= 1 */
    int max_group_rows;                   /* host upper bound of rows per group (grid sizing) */
    int ones_col;                         /* B stored column that reads as 1.0 (bias-grad column), -1 = none */
    long long c_split_stride;             /* floats between split slabs */
    const float* Bg[GI_MAX_GROUPS]; const float* biasg[GI_MAX_GROUPS]; float* Cg[GI_MAX_GROUPS];
    int gsplit[GI_MAX_GROUPS];            /* grouped split-K: slabs of group g (>= 1 each); nsplit ignored */
    /* Bounded (host-sync-free) launches: M (and K) are then UPPER BOUNDS that size the grid, the real
     * extents are read on the device — rows = min(M, *m_dev), reduction length = min(K, *k_dev); output
     * tiles beyond the real rows exit at once.  NULL: M / K as given.  Not with groups or split-K. */
    const int* m_dev; const int* k_dev;
    /* GI_GEMM_X2: the "amax cells" (GI_AMAX_WORDS floats each, see below) holding max |A|, max |B|, read when the kernel
     * starts.  c_amax (any launch, may be NULL): the cell that receives max |C| over everything this launch stores —
     * the caller zeroes the cell before the producer runs. */
    const float* a_amax; const float* b_amax;
    float* c_amax;
    /* GI_GEMM_X2 launches with a row-major A (forward / dgrad layouts), optional: dynamic-range guard.  The launch counts
     * in x2_guard[0] the rows of A whose largest magnitude is non-zero and more than 2^24 below max |A| (a row the
     * per-TENSOR scale leaves with fewer than ~14 significant bits; an exactly-zero row is exact; every row is counted
     * once per launch) and, when it counted a row and x2_guard_host != NULL, stores 1 there. */
    int* x2_guard; int* x2_guard_host;
} gi_gemm_params;

int gi_gemm(const gi_gemm_params* p, void* stream);
/* An amax cell: GI_AMAX_WORDS floats, the tensor's largest magnitude = the maximum over the cell (writers spread their
 * atomic max over 64 slots one 128-byte line apart, csrc/gi_x2.h; all values >= 0, a zeroed cell = "nothing yet"). */
#define GI_AMAX_WORDS 2048
/* max |x| of up to GI_ABSMAX_MAX matrices [rows][cols] (pitch ld) in one launch, maxed into the amax cell `out` (zero it
 * first) — the weights' side of GI_GEMM_X2 (their other operands get theirs from the producing launch's c_amax). */
#define GI_ABSMAX_MAX 16
typedef struct { const float* x; int rows, cols, ld; float* out; } gi_absmax_desc;
int gi_absmax(const gi_absmax_desc* descs, int n, void* stream);
/* fp16x2 dynamic-range guard of weight matrices (descs[i].out = the matrix's amax cell, already filled by gi_absmax on
 * this stream): counts in *counter the ROWS and COLUMNS whose largest magnitude is non-zero and more than 2^24 below
 * the matrix's — an output channel of the forward (row of W) or of the dgrad (column of W) launch that the per-tensor
 * scale leaves with fewer than ~14 significant bits — and stores 1 to *host_flag (may be NULL) when it counted one. */
int gi_x2_weight_guard(const gi_absmax_desc* descs, int n, int* counter, int* host_flag, void* stream);
/* n (<= 8) independent problems of the same tile shape and operand layouts in ONE launch.
 * Every launch is a tile loop: launches with more output tiles than the device holds workgroups at once
 * run as a persistent grid (each workgroup walks tiles id, id + grid, ...; the next tile's first operand
 * loads are issued in front of the current tile's epilogue).  Operands are addressed with 32-bit byte
 * offsets from their base pointers: every matrix must span less than 4 GB (GI_ELIMIT otherwise). */
int gi_gemm_batch(const gi_gemm_params* problems, int n, void* stream);
/* Measurement / test hook (process-wide): persist_tenths >= 0 sets the persistent-grid threshold in tenths of
 * a full round of resident workgroups (default 0 = one workgroup per tile: the persistent grid measured a tie alone
 * and a loss beside the weight-gradient stream);
 * grid_cap > 0 caps every launch at that many workgroups (0 = no cap). */
int gi_gemm_config(int persist_tenths, int grid_cap);   /* returns 0 */

/* ------------------------------------------------------------------------------------------
 * Resident-activation MLP chain — a whole `MLP.forward` (gnn/modules.py:166-170: Linear -> SELU for
 * every layer including the last) of the per-bond-type message / attention stacks
 * (`GGNN.message_terms` gnn/mpnn.py:284-294, `AttentionGGNN.aggregate_message` gnn/mpnn.py:370-389),
 * or the whole dZ chain of its backward, in ONE launch: each workgroup carries 32 rows through all
 * layers with the activation tile resident in LDS; every layer's output is also written to `out`.
 *   forward : layer l: out_l[rows, N] = selu(in_l W_l^T + bias_l),   W_l = [N][K] row-major
 *   backward: layer l: out_l[rows, N] = (in_l W_l) * selu'(act_l),   W_l = [K][N] row-major
 *             (act_l NULL: no SELU factor — the input gradient of the stack's first Linear)
 * in_0 = X rows (optionally gathered by x_idx), in_{l+1} = out_l.  Rows are grouped like gi_gemm's
 * grouped problems (group t uses W[t] / bias[t]; `grp_off` on the device, `group_rows` = host upper
 * bounds for the grid).  Limits: nlayers <= GI_CHAIN_MAXL, 4 <= K, N <= GI_CHAIN_MAXW (GI_ELIMIT
 * otherwise — callers then run the stack layer by layer through gi_gemm).
 * ------------------------------------------------------------------------------------------ */
#define GI_CHAIN_MAXL 8
#define GI_CHAIN_MAXW 256
typedef struct {
    const float* W[GI_MAX_GROUPS];
    const float* bias[GI_MAX_GROUPS];     /* forward only */
    float* out; int ldo;
    const float* act; int ldact;          /* backward only */
    int K, N;
    float* out_amax;                      /* NULL, or an amax cell (GI_AMAX_WORDS floats, csrc/gi_x2.h): the fp16x2 chain kernels
                                             (x2_wamax != NULL) publish max |out| of this layer over ALL groups into it — what
                                             the fp16x2 weight-gradient launches of the stack scale their operands with
                                             (activations: forward chain, dZ: backward chain).  Ignored by the fp32 chain. */
} gi_chain_layer;

typedef struct {
    gi_chain_layer layer[GI_CHAIN_MAXL];
    int nlayers;
    const float* X; int ldx;              /* ldx >= round-up-to-4(layer[0].K) */
    const int* x_idx;                     /* row gather for X, or NULL */
    const int* grp_off;                   /* device: ngroups + 1 row offsets (NULL: one group, rows [0, rows)) */
    int ngroups;
    int group_rows[GI_MAX_GROUPS];        /* host upper bound of rows per group */
    int rows;                             /* total rows */
    int backward;
    float* image;                         /* packed weight image, gi_mlp_chain_image_floats() floats,
                                             16-byte aligned; filled by gi_mlp_chain_pack */
    long long image_stride;               /* floats per group in the image; 0 = that of THESE layers.
                                             A chain that runs only the first layers of a packed
                                             stack passes the stride of the full pack. */
    const int* skip_flag;                 /* NULL, or a device int read at kernel start: != 0 -> the launch does
                                             nothing (gi_ggnn_forward: pass-0 rows served from gi_graph.p0_cache) */
    const int* tile_rows_dev;             /* bounded (host-sync-free) launch, else NULL: device int holding the
                                             row-block height (32..36).  `rows` is then an upper BOUND of the
                                             rows of all groups together (sizes the grid), the real row ranges
                                             come from grp_off on the device (required), group_rows is unused */
    float* x2_wamax;                      /* NULL, or nlayers x ngroups amax cells (GI_AMAX_WORDS floats each, [layer][group]):
                                             the fp16x2 chain (csrc/gi_x2.h) — gi_mlp_chain_pack computes max |W| of
                                             every layer and group into them and writes the image as two scaled fp16
                                             planes; gi_mlp_chain then runs 64-row blocks on the f16 MFMA pipe (three
                                             products per fp32 product, activations scaled per 64-ROW BLOCK and layer by
                                             a power of two from the block's largest magnitude: a row's last bits
                                             depend on which rows share its block — NOT bitwise row-independent like
                                             the fp32 chain: gi_ggnn_backward's dZ chains run this way, gi_ggnn_forward
                                             uses x2_rows32 below).  Pack and launch must agree; a
                                             bounded launch walks 64-row blocks (the value behind tile_rows_dev is unused) */
    float* x_amax;                        /* NULL, or an amax cell for max |X rows read| (after the x_idx gather; all groups) —
                                             published by the fp16x2 chain kernels like layer[].out_amax */
    int x2_rows32;                        /* with x2_wamax: != 0 = the ROW-INDEPENDENT fp16x2 chain — 32-row blocks, every row
                                             of the activation tile scaled by its OWN power of two (the product is formed
                                             transposed, so that a lane holds one row: its maximum is an in-register
                                             reduction), outputs staged through LDS and stored as whole rows.  A row's
                                             result depends on the row and the weights only, bit for bit, like the fp32
                                             chain's: what gi_ggnn_forward uses for the message rows.  Same packed image as
                                             the 64-row variant.  Launches of more row blocks than the chip has CUs run
                                             two workgroups per CU (a build of the kernel with half the LDS). */
} gi_chain_params;

/* The kernel streams the weights as a pre-packed image (one linear stream of 32 KB LDS tile images per
 * group, zero padded, all layers back to back): gi_mlp_chain_pack writes it from layer[].W (needs
 * nlayers, ngroups, backward, K/N/W of every layer and `image`); it stays valid until the weights
 * change, for any rows / X / out of the same stack. */
/* ------------------------------------------------------------------------------------------
 * Operand image for GI_GEMM_BF3 launches: the three bf16 planes [rows][Kp] (Kp = cols rounded up to 32, zero
 * padded) of a weight matrix, one plane after the other — B[n][k] = W[n * ld + k], or W[k * ld + n] with
 * `transpose` (dgrad of a torch.nn.Linear weight [out, in]: rows = in, cols = out, ld = in).  Valid until the
 * weights change.  `image`: gi_bf3_image_elems(rows, cols) 2-byte elements, 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
#define GI_BF3_PACK_MAX 16
#define GI_BF3_DEFAULT 1    /* gi_ggnn_forward / backward: node-level readout layers >= 192 wide on GI_GEMM_BF3 launches
                              (environment GI_BF3=0 / 1 overrides) */
typedef struct {
    const float* W; int rows, cols, ld; int transpose;
    unsigned short* image;
    int as_f32;               /* != 0: write B as plain fp32 [rows][(cols + 3) & ~3] instead of the three planes (for
                                 GI_GEMM_BF3B_F32 launches: a transposed copy of a weight, 4 bytes per element through L2
                                 instead of 6); fits the same buffer */
} gi_bf3_pack_desc;
/* Which kernel runs a GI_GEMM_BF3 launch whose operands are plain fp32 (measurement aids; on = 1 / 0 sets, on < 0
 * queries; both return the previous setting; environment GI_B3P / GI_B3V set the initial values, default 1):
 *   gi_b3p_enable: the software-pipelined 128 x 256 kernel (gi_gemm_b3p.hip) — weight-gradient launches
 *                  (a_major + b_major + split-K slabs; autograd of gnn/modules.py:166-170) always, forward / dgrad
 *                  launches from 2.5 tiles per CU on;
 *   gi_b3v_enable: the 32-deep-tile 128 x 128 kernel (gi_gemm_b3v.hip) for what is left (dgrad with W as stored);
 * everything else — and everything when both are off — runs on the round-3 kernel (gi_gemm_bf3.hip). */
int gi_b3p_enable(int on);
int gi_b3v_enable(int on);
/* gi_ggnn_forward / backward: their GI_GEMM_BF3 launches as fp16x2 (GI_GEMM_X2: two scaled fp16 planes per operand,
 * three products; environment GI_X2, default 1) or as bf16x3 (0).  on < 0 queries; returns the previous setting. */
int gi_x2_enable(int on);
/* Process-wide switch of gi_ggnn_forward / backward's use of GI_GEMM_BF3 launches (initial value: environment
 * GI_BF3, else GI_BF3_DEFAULT): on = 1 / 0 sets it, on < 0 only queries; returns the previous setting.  The
 * workspace size does not depend on it. */
int gi_bf3_enable(int on);
long long gi_bf3_image_elems(int rows, int cols);
int gi_bf3_pack(const gi_bf3_pack_desc* descs, int n, void* stream);

long long gi_mlp_chain_image_floats(const gi_chain_params* p);
int gi_mlp_chain_pack(const gi_chain_params* chains, int nchains, void* stream);
/* nchains (1 or 2, same direction) independent chains in one launch; `image` must be packed. */
int gi_mlp_chain(const gi_chain_params* chains, int nchains, void* stream);
/* Test / measurement hook (process-wide; the launcher's own choices are tile_rows = 0, rows64 = -1, ring = 2,
 * trace = NULL): tile_rows in 32..36 forces the row-block height (32 MFMA rows + extra rows on the VALU);
 * rows64 = 1 / 0 forces / forbids the 64-row variant (-1: where it pays); ring = 3 streams the weights of
 * 32-row blocks through a three-slot ring (146 KB of LDS; 2: two slots, 114 KB, a GEMM workgroup fits beside it);
 * trace = device buffer of 16 int64 per workgroup for per-phase timestamps (tools/trace_chain.py). */
int gi_mlp_chain_config(int tile_rows, int rows64, int ring, void* trace);

/* ------------------------------------------------------------------------------------------
 * Graph / pointwise kernels
 * ------------------------------------------------------------------------------------------ */
/* K4 seg_sum — replaces `torch.matmul(message_summation_matrix, message_terms)`
 * (gnn/summation_mpnn.py:141) and, in backward, the scatter of d(gathered rows):
 *   out[c, 0:cols] (+)= sum_{k in [off[c], off[c+1])} vals[perm[k], 0:cols],  c < rows */
int gi_seg_sum(const float* vals, int ldv, const int* perm, const int* off, int rows, int cols,
               float* out, int ldo, int accumulate, void* stream);

/* Backward of the aggregation onto MESSAGE rows, fused with the SELU backward of the message MLP's
 * last layer, in place:  y[u, c] = selu'(y[u, c]) * sum_{k in [off[u], off[u+1])} vals[perm[k], c] */
int gi_seg_sum_dselu(const float* vals, int ldv, const int* perm, const int* off, int rows, int cols,
                     float* y, int ldy, void* stream);

/* y[r, c] = selu'(y[r, c]) * sum_{s < nsplit} slabs[s * stride + r * ld + c]: sums the split-K slabs
 * of the pass-0 aggregation backward (cmat^T . d agg) with the SELU backward folded in. */
/* y_i[d, c] = selu'(y_i[d, c]) * sum_{k in [off[d], off[d+1])} vals_i[idx[k], c] for i = 0 (and 1 when
 * vals1 != NULL): the per-row sums of AttentionGGNN's pass-0 softmax backward (long segments). */
int gi_class_sum_dselu(const float* vals0, const float* vals1, int ldv, const int* idx, const int* off,
                       int rows, int cols, float* y0, float* y1, int ldy, void* stream);
int gi_slab_sum_dselu(const float* slabs, int nsplit, long long stride, int rows, int cols, int ld,
                      float* y, int ldy, void* stream);

/* out[r, c] = epilogue(sum_{s < nsplit} slabs[s * stride + r * ld + c]), epilogue = GI_EPI_* flags as in
 * gi_gemm (BIAS, SELU, DSELU / MULACT through act, ACCUM): finishes a GI_GEMM_SPLITK forward / dgrad
 * problem — the skinny, long-reduction layers of the graph-level stacks (B rows, K = N*A + G: 9 252 at the
 * ChEMBL shape) run split-K so that more than a handful of workgroups share the reduction. */
int gi_slab_epilogue(const float* slabs, int nsplit, long long stride, int rows, int cols, int ld,
                     int flags, const float* bias, const float* act, int ldact, float* out, int ldo,
                     void* stream);

/* Attention aggregation of AttentionGGNN — replaces `aggregate_message` (gnn/mpnn.py:370-389:
 * mask, Softmax(dim=1) over the padded neighbour axis, weighted sum) on the destination CSR:
 *   att[k, f] = softmax over k in [off[c], off[c+1]) of en[perm[k], f];
 *   out[c, f] = sum_k att[k, f] * emb[perm[k], f]        (0 for an empty segment),  c < rows.
 * en / emb: [U, ld] post-SELU outputs of the per-bond-type energy / message MLPs on message rows;
 * perm = in_perm (dst-CSR slot -> message row). */
int gi_seg_softmax_fwd(const float* en, const float* emb, int ld, const int* perm, const int* off,
                       int rows, int cols, float* out, int ldo, void* stream);
/* Its backward, per edge: with dagg = d loss / d out,
 *   d_emb_e[k, f] = att[k, f] * dagg[c, f],   d_en_e[k, f] = att[k, f] * (emb[perm[k], f] * dagg[c, f] - <att, emb*dagg>_c)
 * written in dst-CSR slot order [E, lde] (the softmax is recomputed from en).  The message-row
 * gradients are gi_seg_sum_dselu of these over the message CSR (perm = mu_slot, off = mu_off). */
int gi_seg_softmax_bwd(const float* en, const float* emb, int ld, const int* perm, const int* off,
                       int rows, int cols, const float* dagg, int ldd, float* d_en_e,
                       float* d_emb_e, int lde, void* stream);

/* out[r, c] = dY[idx ? idx[r] : r, c] * selu'(Y[r, c])  (may run in place on Y) */
int gi_selu_bwd_rows(const float* dY, int lddy, const int* idx, const float* Y, int ldy,
                     float* out, int ldo, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------
 * AlphaDropout, training mode with p > 0 — `torch.nn.AlphaDropout(dropout_p)` behind every
 * Linear + SELU of gnn/modules.py:130-142 (MLP._linear_block), including the last one.
 * ------------------------------------------------------------------------------------------
 * One site = one MLP layer's output buffer.  keep(row, col) is a counter-based hash of (seed, id,
 * row, col) compared with thresh = p * 2^32; nothing is stored except, `fshift` floats behind every
 * activation, the factor d y / d z = keep * a * selu'(z) the backward multiplies by
 * (GI_EPI_MULACT; the *_f variants below).  The arithmetic is ATen's _dropout_impl:
 *   y = fl(fl(s * (keep ? a : 0)) + (keep ? b_keep : b_drop)),  a = ((alpha'^2 p + 1)(1 - p))^-1/2,
 *   b = (keep - 1) * alpha' a + alpha' a p,  alpha' = 1.7580993408473766  (every scalar cast to fp32). */
typedef struct gi_dropout_params {
    unsigned long long seed;
    unsigned id, thresh;
    float a, b_keep, b_drop;
} gi_dropout_params;
/* fills `out` for probability p in [0, 1) (p == 0: keep everything, y == s, factor == selu') */
int gi_dropout_setup(double p, unsigned long long seed, unsigned id, gi_dropout_params* out);
/* in place on y[rows, 0:cols] (SELU outputs) -> AlphaDropout outputs; factors to y + fshift */
int gi_alpha_dropout_fwd(float* y, int ldy, int rows, int cols, long long fshift,
                         const gi_dropout_params* q, void* stream);
/* test hook: the keep bits of one site as bytes [rows, ld] */
int gi_dropout_mask(const gi_dropout_params* q, int rows, int cols, unsigned char* keep, int ld,
                    void* stream);
/* The SELU-backward sites with the factor read from `fshift` floats behind the activation instead of
 * derived from it (fshift == 0: exactly the functions without the suffix; fshift % 4 == 0). */
int gi_seg_sum_dselu_f(const float* vals, int ldv, const int* perm, const int* off, int rows,
                       int cols, float* y, int ldy, long long fshift, void* stream);
int gi_selu_bwd_rows_f(const float* dY, int lddy, const int* idx, const float* Y, int ldy,
                       float* out, int ldo, int rows, int cols, long long fshift, void* stream);
int gi_gather_readout_bwd_f(float* en, float* emb, int ld, const int* cidx, const int* node_mask,
                            int B, int N, int G, int S, float big,
                            const float* dg0, int ld0, const float* dg1, int ld1,
                            const float* dg2, int ld2, float* zpart, long long fshift, void* stream);
int gi_compress_slots_f(float* t1, int ldt, const int* cidx, int B, int N, int W, int S,
                        const float* dcat, int ldc, float* zpart, int ldz, long long fshift,
                        void* stream);

/* GRU gates — torch.nn.GRUCell as used at gnn/mpnn.py:249-253,296-297, applied only to rows
 * with >=1 incoming edge (gnn/summation_mpnn.py:107,124,143-144 update only those nodes).
 * In: gi,gh [rows,3H] (+bias already added), hx_prev [rows, ldh]; out: hx_new (cols [0,H) and
 * the feature tail [H,H+Fn) copied); gi is overwritten with (r|z|n), gh keeps W_hn h + b_hn. */
/* The whole GRU update of a message pass (torch.nn.GRUCell as called at gnn/mpnn.py:296-297, with the node mask of
 * gnn/summation_mpnn.py:146) in ONE launch: gi = agg W_ih^T + b_ih, gh = h W_hh^T + b_hh on the fp32 MFMA and the gate
 * arithmetic in registers.  agg [rows, lda] (M columns), hx [rows, ldh] = [h (H) | features | padding], W_ih [3H, M],
 * W_hh [3H, H] row-major.  Writes h' and the copied tail into hx_new [rows, ldh], and what gi_gru_gates_bwd reads: the
 * gates r, z, n into gi[:, 0:3H], gh_n into gh[:, 2H:3H] (gi / gh rows of nodes without incoming edges, and gh[:, 0:2H],
 * are left untouched).  H % 4 == M % 4 == lda % 4 == ldh % 4 == 0, 16-byte aligned agg / hx / hx_new; GI_EINVAL otherwise
 * (callers then use gi_gemm_batch + gi_gru_gates_fwd). */
int gi_gru_forward(const float* agg, int lda, const float* hx, int ldh, const float* Wih, const float* Whh,
                   const float* bih, const float* bhh, float* gi, float* gh, int ldg, float* hx_new,
                   const int* seg_off, int rows, int H, int M, void* stream);
int gi_gru_gates_fwd(float* gi, float* gh, int ldg, const float* hx_prev, float* hx_new, int ldh,
                     const int* seg_off, int rows, int H, int Fn, void* stream);
/* In: dh_new (+ up to three more partial gradients dh_b/c/d or NULL, all [rows, lddh]);
 * gi=(r|z|n), gh=(..|..|hn) from forward are overwritten with d gi, d gh;
 * dh_prev = direct part of the gradient to h_prev. */
int gi_gru_gates_bwd(float* gi, float* gh, int ldg, const float* hx_prev, int ldh,
                     const float* dh_new, const float* dh_b, const float* dh_c, const float* dh_d,
                     float* dh_prev, int lddh, const int* seg_off, int rows, int H, void* stream);

/* Launch-count reductions of the training step.  gi_fuse_flags() = the bit mask (environment GI_FUSE,
 * default GI_FUSE_DEFAULT) of the fused / vectorised variants gi_ggnn_forward / gi_ggnn_backward use.
 * DH_SCATTER, TIER2_DSELU and SLOTS reproduce the launches they replace bit for bit (same
 * operations, same summation orders; tests/test_kernels_gpu.py compares them with torch.equal); GATES_V4
 * evaluates the same formulas four hidden units at a time and agrees with the scalar kernels to rounding
 * (hipcc contracts the multiply-adds of the vector code differently: <= 1e-6 relative, same test file).
 *   GATES_V4     gi_gru_gates_fwd / _bwd on 16-byte vectors (4 hidden units per thread) when H % 4 == 0
 *   DH_SCATTER   the scatter of the message stacks' input gradients to their source nodes
 *                (gi_seg_sum over the source CSR, accumulating into d h; backward of
 *                `nodes[edge_batch_nghb_idc]`, gnn/summation_mpnn.py:131-133) folded into the
 *                gi_gru_gates_bwd_ex launch of the next (earlier) message pass
 *   TIER2_DSELU  the SELU backward of the three logit column ranges in one launch (gi_selu_bwd_cols3_f)
 *   SLOTS        fAddNet1 / fConnNet1 glue in one launch each way (gi_expand_slots2, gi_compress_slots2_f) */
#define GI_FUSE_GATES_V4    1
#define GI_FUSE_DH_SCATTER  2
#define GI_FUSE_TIER2_DSELU 4
#define GI_FUSE_SLOTS       8
/* Measured on the headline step (round 2, two A/B pairs): 15 -> 2.319 / 2.325 ms against 2.359 / 2.353 with
 * everything off. */
#define GI_FUSE_DEFAULT     15
int gi_fuse_flags(void);
/* gi_gru_gates_bwd with d h = dh_new + sum over the source-CSR segment [sc_off[r], sc_off[r+1]) of the rows
 * sc0[sc_perm[k]] (+ the same over sc1 when non-NULL; both [*, ldsc >= H]) — what gi_seg_sum(sc, sc_perm,
 * sc_off, ..., dh_new, accumulate) launches in front of gi_gru_gates_bwd would have left in dh_new, bit for
 * bit.  sc0 == NULL: plain gi_gru_gates_bwd on the vector kernel (scalar kernel when H % 4 != 0 or a
 * pointer / leading dimension is not 16-byte aligned; with sc0 that case is GI_EINVAL). */
int gi_gru_gates_bwd_ex(float* gi, float* gh, int ldg, const float* hx_prev, int ldh,
                        const float* dh_new, const float* dh_b, const float* dh_c, const float* dh_d,
                        float* dh_prev, int lddh, const int* seg_off, int rows, int H,
                        const float* sc0, const float* sc1, int ldsc, const int* sc_perm,
                        const int* sc_off, void* stream);
/* out_k[r, c] = dY[r, s_k + c] * selu'(Y[r, s_k + c]) for the three consecutive column ranges of widths
 * n0, n1, n2 (s_0 = 0, s_1 = n0, s_2 = n0 + n1): three gi_selu_bwd_rows_f calls in one launch
 * (backward entry of the tier-2 stacks, gnn/modules.py:265-279).  fshift as in gi_selu_bwd_rows_f. */
int gi_selu_bwd_cols3_f(const float* dY, int lddy, const float* Y, int ldy, long long fshift, int rows,
                        int n0, float* out0, int ld0, int n1, float* out1, int ld1,
                        int n2, float* out2, int ld2, void* stream);
/* gi_expand_slots for two tier-1 outputs (a, b) sharing cidx in one launch; gi_compress_slots_f likewise */
int gi_expand_slots2(const float* t1a, int ldta, int Wa, float* cata, int ldca,
                     const float* t1b, int ldtb, int Wb, float* catb, int ldcb,
                     const int* cidx, int B, int N, void* stream);
int gi_compress_slots2_f(float* t1a, int ldta, int Wa, const float* dcata, int ldca, float* zparta, int ldza,
                         float* t1b, int ldtb, int Wb, const float* dcatb, int ldcb, float* zpartb, int ldzb,
                         const int* cidx, int B, int N, int S, long long fshift, void* stream);

/* K7 gather readout — gnn/modules.py:44-52: g[b,:] = sum_n softmax_n(en[cidx[b,n]] - big*[mask==0]) * emb[cidx[b,n]],
 * written to up to three destinations (tier-2 concat inputs). */
int gi_gather_readout_fwd(const float* en, const float* emb, int ld, const int* cidx,
                          const int* node_mask, int B, int N, int G, float big,
                          float* out0, int ld0, float* out1, int ld1, float* out2, int ld2,
                          void* stream);
/* backward: dg = dg0+dg1+dg2 per graph; writes dZ of the last att/emb layers IN PLACE over
 * en/emb rows < S, and per-graph partial sums for the zero row into zpart[B, 2*G]. */
int gi_gather_readout_bwd(float* en, float* emb, int ld, const int* cidx, const int* node_mask,
                          int B, int N, int G, int S, float big,
                          const float* dg0, int ld0, const float* dg1, int ld1,
                          const float* dg2, int ld2, float* zpart, void* stream);

/* tier-1 -> tier-2 glue (gnn/modules.py:256-262 `view` + `cat`):
 *   cat[b, n*W + w] = t1[cidx[b,n], w] */
int gi_expand_slots(const float* t1, int ldt, const int* cidx, int B, int N, int W,
                    float* cat, int ldc, void* stream);
/* backward: t1 rows < S are overwritten in place with dcat * selu'(t1); inactive slots are
 * summed per graph into zpart[B, W] (raw, no selu'). */
int gi_compress_slots(float* t1, int ldt, const int* cidx, int B, int N, int W, int S,
                      const float* dcat, int ldc, float* zpart, int ldz, void* stream);
/* out[c] = (sum_b part[b, c]) * (y ? selu'(y[c]) : 1), deterministic */
int gi_colsum(const float* part, int ldp, int rows, int cols, const float* y, float* out,
              void* stream);
/* n (<= 8) independent column sums in one launch */
typedef struct gi_colsum_desc {
    const float* part; int ldp, rows, cols; const float* y; float* out;
} gi_colsum_desc;
int gi_colsum_multi(const gi_colsum_desc* descs, int n, void* stream);

/* sums wgrad slabs into the parameter gradients: for each descriptor,
 *   dW[n,k] = sum_s slab[s][n*ld + k] (k < K),  db[n] = sum_s slab[s][n*ld + K] */
typedef struct gi_reduce_desc {
    const float* slabs; float* dW; float* db;
    long long slab_stride; int n_slabs, N, K, ld;
} gi_reduce_desc;
int gi_reduce_slabs(const gi_reduce_desc* descs, int n_desc, void* stream);

/* torch.optim.Adam step (no amsgrad) over ONE flat fp32 bucket of n floats (n % 4 == 0, 16-byte
 * aligned): the optimizer the reference builds at Workflow.py:219-263, as a single launch. */
int gi_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1,
                 double beta2, double eps, double weight_decay, int step, void* stream);

/* Workflow.loss (Workflow.py:833-860) forward + gradient in one pass over the logits:
 * row_loss[b] = KL(target_b / sum(target_b) || softmax(out_b)); loss = mean_b row_loss[b];
 * d_out (may be NULL) = d loss / d out; loss_mean (may be NULL) = the scalar batch mean, summed in a
 * fixed order.  All-zero target rows give NaN like the reference. */
int gi_kl_loss(const float* out, int ldo, const void* target, int tgt_dtype, int ldt, int B,
               int width, float* row_loss, float* d_out, int ldd, float* loss_mean, void* stream);

/* x[0:n] *= *scale with the scalar read on the device (the upstream gradient autograd hands to the
 * loss node: `loss.backward()` passes ones, RL-style callers pass a weight) — no host read-back. */
int gi_scale_by_scalar(float* x, long long n, const float* scale, void* stream);

/* Sampling step of graph generation — replaces `softmax(model(nodes, edges))` +
 * GraphGenerator.get_actions / get_invalid_actions (GraphGenerator.py:121, 467-657) with one launch:
 * per graph b, softmax of logits[b, 0:W] (W = N*A + N*Fe + 1), ONE categorical draw = the first
 * action whose cumulative probability exceeds uniform[b] (inverse CDF; uniform in [0,1)), decode and
 * validity rules.  n_nodes[B] int32 = atoms currently in each graph; edges [B,N,N,Fe] (GI_DTYPE).
 *   action[b] = {kind (0 add, 1 connect, 2 terminate), node_to, rem, from}: rem = index inside the
 *               node's block (add: ravelled (atom type, charge, ..., bond type); connect: bond type);
 *               from = n_nodes (add; 0 where the reference resets it, :567) or n_nodes-1 (connect)
 *   likelihood[b] = probability of the drawn action (:541)
 *   flags[b]  bit 0: invalid action (:573-646); bit 1: the add's "connect to" index was reset (:650-654) */
int gi_sample_actions(const float* logits, int ldl, const float* uniform, const int* n_nodes,
                      const void* edges, int edges_dtype, int B, int N, int A, int Fe, int* action,
                      float* likelihood, int* flags, void* stream);

/* Optional per-launch timing for the benchmark's roofline leg: when enabled, every gi_gemm and
 * gi_seg_sum launch is bracketed by hipEvents on its stream.  gi_prof_collect blocks until the
 * recorded work finished and returns, per kernel family k (0 = GEMM, 1 = seg_sum): summed elapsed
 * ms, busy ms = length of the UNION of the launches' [start, stop] intervals (launches on the
 * backward's two streams overlap; the sum double-counts that time), summed work (GEMM: useful flops
 * 2*M*N*K; seg_sum: 0, the caller knows the bytes) and the number of launches; it clears the log. */
int gi_prof_enable(int on);
int gi_prof_collect(double* ms, double* busy_ms, double* work, int* launches);
/* The GEMM family of the LAST gi_prof_collect split by the matrix pipe the launch ran on — [0] fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), [1] bf16 MFMA (GI_GEMM_BF3: six bf16 products per fp32 product), [2] f16 MFMA (GI_GEMM_X2:
 * three f16 products): summed elapsed ms, useful flops 2*M*N*K and launches, three entries each. */
int gi_prof_pipes(double* ms, double* work, int* launches);

/* ------------------------------------------------------------------------------------------
 * Whole-model entry points: one call enqueues the complete GGNN forward (or backward) —
 * `SummationMPNN.forward` message passes + `GGNN.readout` (gnn/summation_mpnn.py:128-149,
 * gnn/mpnn.py:284-303).  Parameters are passed as a pointer table in state_dict order.
 * ------------------------------------------------------------------------------------------ */
typedef struct gi_ggnn_dims {
    int B, N, Fn, Fe, H, M, G, A, C, passes;
    int enn_depth, enn_hidden, att_depth, att_hidden, emb_depth, emb_hidden;
    int mlp1_depth, mlp1_hidden, mlp2_depth, mlp2_hidden;
    float big_positive;
    /* model kind: GI_KIND_GGNN — sum aggregation (gnn/mpnn.py:229-303);  GI_KIND_ATTGGNN —
     * `AttentionGGNN` (gnn/mpnn.py:306-398): a second per-bond-type MLP (eatt_depth/eatt_hidden =
     * constants.att_depth / att_hidden_dim; enn_* then carry msg_depth / msg_hidden_dim) gives
     * attention energies, aggregated with gi_seg_softmax_*.  Parameter table order: msg_nns of all
     * bond types, [att_nns of all bond types,] gru, gather, APDReadout (state_dict order). */
    int kind, eatt_depth, eatt_hidden;
    /* AlphaDropout training mode (gnn/modules.py:130-142 with dropout_p > 0; eval and p == 0: all
     * zero).  dropout != 0: every MLP layer output goes through torch.nn.AlphaDropout with its
     * stack's probability (enn/msg stacks, AttentionGGNN's energy stacks, gather att_nn / emb_nn,
     * fAddNet1+fConnNet1, the three tier-2 stacks) and masks drawn from drop_seed.  Requires a graph
     * compacted WITHOUT row sharing (gi_compact_count_ex, nodedup = 1); the workspace doubles (every
     * activation gets a factor twin); the logits buffer `out` must have 2 B rows (rows [B, 2B) take
     * the factors of the logits) and the same buffer must be handed to the backward as y_out. */
    int dropout;
    float drop_enn, drop_eatt, drop_att, drop_emb, drop_mlp1, drop_mlp2;
    unsigned long long drop_seed;
} gi_ggnn_dims;
#define GI_KIND_GGNN 0
#define GI_KIND_ATTGGNN 1

/* Creates / destroys a lowest-priority, non-blocking stream on the current device for
 * gi_ggnn_backward's `side_stream` argument (caller-owned handle; any other stream of the same
 * device works too). */
int gi_side_stream_create(void** stream);
int gi_side_stream_destroy(void* stream);

int gi_ggnn_num_params(const gi_ggnn_dims* d);
/* S active slots, E directed edges, U message rows, D0 pass-0 rows (counts[0], [1], [3], [20]) */
long long gi_ggnn_workspace_floats(const gi_ggnn_dims* d, int S, int E, int U, int D0);
long long gi_ggnn_slab_floats(const gi_ggnn_dims* d, int S, int U, const int* Ut);
/* ws must hold hx0 (from gi_compact_fill) at offset gi_ggnn_hx0_offset(); forward keeps every
 * activation in ws for backward. */
long long gi_ggnn_hx0_offset(const gi_ggnn_dims* d, int S, int E, int U, int D0);
int gi_ggnn_ldhx(const gi_ggnn_dims* d);
/* test/debug hook: offset (floats) and leading dimension of a named workspace buffer
 * ("hx" i=pass, "eact" i=pass j=layer, "m","agg","gi","gh" i=pass, "att_act" j=layer, "en", ...) */
int gi_ggnn_ws_query(const gi_ggnn_dims* d, int S, int E, int U, int D0, const char* name, int i,
                     int j, long long* off, int* ld);
int gi_ggnn_forward(const gi_ggnn_dims* d, const float* const* params, const gi_graph* g,
                    float* ws, float* out, int ldout, void* stream);
/* The same with a second stream and run flags.  side_stream (may be NULL; gi_side_stream_create): everything that
 * depends on the WEIGHTS only is enqueued there instead of in front of the kernels that wait for it — the amax cells
 * of the fp16x2 layers' weights and their dynamic-range check (needed by the readout, ~0.4 ms into the forward), and,
 * with GI_RUN_PREPACK_BWD, what gi_ggnn_backward would otherwise pack at its start: the W^T images of the 16-bit-pipe
 * layers and the dZ chains' weight image (then call the backward with GI_BWD_PREPACKED: it waits for that work
 * instead of redoing it).  `stream` waits for the side stream where it needs the results, and at the end of the call
 * for the packs into `ws`: everything the caller enqueues on `stream` afterwards (the backward, a reuse of the
 * workspace's memory) is ordered behind the side stream's work on `ws`.
 * GI_RUN_NO_X2: this call's 16-bit-pipe launches as bf16x3 instead of fp16x2 (what GI_X2=0 does process-wide) — what a
 * caller switches to after gi_graph.x2_guard_host tripped.  Forward and backward of one tape must agree. */
#define GI_RUN_PREPACK_BWD 1
#define GI_RUN_NO_X2 2
int gi_ggnn_forward_ex(const gi_ggnn_dims* d, const float* const* params, const gi_graph* g,
                       float* ws, float* out, int ldout, void* stream, void* side_stream, int flags);
/* Pass-0 row cache for inference loops (generation: GraphGenerator.py:118-157 calls the model thousands of
 * times with the same weights).  In the first message pass h = [x | 0], so a message row (and AttentionGGNN's
 * energy row) depends only on (bond type, 0/1 feature pattern of the source node) and the weights: a few dozen
 * distinct rows per batch, the same ones forward after forward.  With gi_graph.p0_cache set, gi_ggnn_forward
 * looks every pass-0 row of the batch up in a device-side table keyed by (bond type, pattern); when all are
 * there, the rows are copied from the table and the pass-0 stack launch exits at once (gi_chain_params.
 * skip_flag), otherwise the stack runs as usual and its rows are added to the table (the table is emptied when
 * they would not fit).  No host involvement: works inside a bounded (host-sync-free) forward and a captured
 * hipGraph.  The CALLER zero-fills the buffer whenever a weight of the message / energy stacks changes
 * (gnn/mpnn.py does, from the parameters' versions).  words[0] = hit flag of the latest forward, [1] = rows
 * in the table, [2] / [3] = forwards / hits so far.  Returns the buffer size in 4-byte words, < 0 on error. */
long long gi_p0_cache_words(const gi_ggnn_dims* d);
/* floats of gi_graph.wcache for this model (independent of the batch) */
long long gi_ggnn_wcache_floats(const gi_ggnn_dims* d);
/* consumes (overwrites) the activations in ws; y_out = the logits forward returned; grads[i]
 * receives the gradient of params[i]; slabs = gi_ggnn_slab_floats() floats of scratch.
 * side_stream (may be NULL): a second hipStream_t of the same device.  When given, the weight-
 * gradient GEMMs run there, ordered after their operands by events, concurrently with the dZ chain
 * on `stream`; `stream` waits for them before the final slab reduction, so on return every piece of
 * work is ordered before whatever the caller enqueues next on `stream`. */
int gi_ggnn_backward(const gi_ggnn_dims* d, const float* const* params, const gi_graph* g,
                     float* ws, float* slabs, const float* y_out, int ldout, const float* d_out,
                     int lddout, float* const* grads, void* stream, void* side_stream);
/* The same in two calls, for overlapping the data-parallel gradient exchange with the backward:
 * GI_BWD_READOUT differentiates the readout (gather + APDReadout: ~86 % of the parameters) and
 * completes the gradients of params[gi_ggnn_first_readout_param() ..) — on `side_stream` when given
 * (record an event there to know when), else on `stream`; GI_BWD_PASSES then differentiates the
 * message passes and completes the rest.  Same arguments to both calls; GI_BWD_ALL = gi_ggnn_backward. */
#define GI_BWD_ALL 0
#define GI_BWD_READOUT 1
#define GI_BWD_PASSES 2
/* OR-ed into `phase`: GI_BWD_PREPACKED — the forward ran as gi_ggnn_forward_ex(.., GI_RUN_PREPACK_BWD) with the same
 * side stream: the weight images of this backward are (being) written there, the call waits for them instead of packing;
 * GI_BWD_NO_X2 — bf16x3 instead of fp16x2, like GI_RUN_NO_X2 (the forward of the same tape must have run that way). */
#define GI_BWD_PREPACKED 0x100
#define GI_BWD_NO_X2 0x200
int gi_ggnn_backward_phase(const gi_ggnn_dims* d, const float* const* params, const gi_graph* g,
                           float* ws, float* slabs, const float* y_out, int ldout,
                           const float* d_out, int lddout, float* const* grads, void* stream,
                           void* side_stream, int phase);
int gi_ggnn_first_readout_param(const gi_ggnn_dims* d);

#ifdef __cplusplus
}
#endif
#endif
