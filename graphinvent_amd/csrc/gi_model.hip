// Host-side orchestration of the whole GGNN / AttentionGGNN forward and backward (gfx950): the
// forward on one HIP stream, the backward on two (dZ chain; weight gradients + their reductions).
//
// One call from Python enqueues every kernel of `SummationMPNN.forward`'s message passes and
// `GGNN.readout` (gnn/summation_mpnn.py:128-149, gnn/mpnn.py:284-303) — no Python between
// launches, no host synchronisation, no allocation: all buffers are carved out of one caller
// provided fp32 workspace by `make_ws`, which forward and backward evaluate identically.
//
// Data layout in HBM (all fp32 row-major, leading dimensions rounded up to 4 floats = 16 B):
//   node level  : R = S+1 compact rows (row S = shared zero row);  hx[p] = [h (H) | x (Fn)] per pass
//   message lvl : U rows, one per distinct (source node, bond type) pair, bond-type-major (rows of
//                 type t = [type_off[t], type_off[t+1])); see gi_compact.hip
//   graph level : B rows
//   pass 0      : D0 rows, one per (feature class, bond type) pair, when the shortcut applies
// Backward: the dZ of a stack's LAST layer overwrites that layer's output in place; the dZ of hidden
// layers go to their own buffers, so that every weight-gradient GEMM (which needs dZ_l and the
// activation below it) can be deferred, batched and run on the side stream.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "gi_common.h"

namespace {

constexpr int MAXL = 12;   // max Linear layers per MLP (depth + 1)
constexpr int MAXP = 16;   // max message passes

struct Mlp {
    int base, in, hidden, depth, out;
    float drop_p = 0.f;      // AlphaDropout probability of the stack (training mode only)
    int layers() const { return depth + 1; }
    int fan_in(int l) const { return l == 0 ? in : hidden; }
    int fan_out(int l) const { return l == depth ? out : hidden; }
    int w(int l) const { return base + 2 * l; }
    int b(int l) const { return base + 2 * l + 1; }
};

struct Model {
    gi_ggnn_dims d;
    Mlp msg[GI_MAX_GROUPS], eatt[GI_MAX_GROUPS], att, emb, add1, conn1, add2, conn2, term2;
    int gru_wih, gru_whh, gru_bih, gru_bhh, nparams, NA, NC;
};

int build_model(const gi_ggnn_dims* dp, Model& m) {
    if (!dp) return GI_EINVAL;
    const gi_ggnn_dims& d = *dp;
    if (d.B <= 0 || d.N <= 0 || d.Fn <= 0 || d.Fe <= 0 || d.H <= 0 || d.M <= 0 || d.G <= 0 ||
        d.A <= 0 || d.C <= 0 || d.passes < 0 || d.Fn > d.H)
        return GI_EINVAL;
    if (d.N > GI_MAX_NODES || d.Fe > GI_MAX_GROUPS || d.passes > MAXP) return GI_ELIMIT;
    if (d.kind != GI_KIND_GGNN && d.kind != GI_KIND_ATTGGNN) return GI_EINVAL;
    const bool attn = d.kind == GI_KIND_ATTGGNN;
    if (attn && d.eatt_hidden <= 0 && d.eatt_depth > 0) return GI_EINVAL;
    const int depths[] = {d.enn_depth, d.att_depth, d.emb_depth, d.mlp1_depth, d.mlp2_depth,
                          attn ? d.eatt_depth : 0};
    for (int x : depths)
        if (x < 0 || x + 1 > MAXL) return GI_ELIMIT;
    m.d = d;
    m.NA = d.N * d.A;
    m.NC = d.N * d.C;
    int idx = 0;
    const float drops[] = {d.drop_enn, d.drop_eatt, d.drop_att, d.drop_emb, d.drop_mlp1, d.drop_mlp2};
    for (float x : drops)
        if (!(x >= 0.f) || !(x < 1.f)) return GI_EINVAL;
    auto mk = [&](int in, int hidden, int depth, int out, float drop_p) {
        Mlp r{idx, in, hidden, depth, out, d.dropout ? drop_p : 0.f};
        idx += 2 * (depth + 1);
        return r;
    };
    for (int t = 0; t < d.Fe; ++t) m.msg[t] = mk(d.H, d.enn_hidden, d.enn_depth, d.M, d.drop_enn);
    if (attn)   // AttentionGGNN registers msg_nns before att_nns (gnn/mpnn.py:316-317)
        for (int t = 0; t < d.Fe; ++t)
            m.eatt[t] = mk(d.H, d.eatt_hidden, d.eatt_depth, d.M, d.drop_eatt);
    m.gru_wih = idx++; m.gru_whh = idx++; m.gru_bih = idx++; m.gru_bhh = idx++;
    m.att = mk(d.H + d.Fn, d.att_hidden, d.att_depth, d.G, d.drop_att);
    m.emb = mk(d.H, d.emb_hidden, d.emb_depth, d.G, d.drop_emb);
    m.add1 = mk(d.H, d.mlp1_hidden, d.mlp1_depth, d.A, d.drop_mlp1);
    m.conn1 = mk(d.H, d.mlp1_hidden, d.mlp1_depth, d.C, d.drop_mlp1);
    m.add2 = mk(m.NA + d.G, d.mlp2_hidden, d.mlp2_depth, m.NA, d.drop_mlp2);
    m.conn2 = mk(m.NC + d.G, d.mlp2_hidden, d.mlp2_depth, m.NC, d.drop_mlp2);
    m.term2 = mk(d.G, d.mlp2_hidden, d.mlp2_depth, 1, d.drop_mlp2);
    m.nparams = idx;
    return 0;
}

// ---- workspace --------------------------------------------------------------------------------
struct Ws {
    int R, E, U, D0, B;
    long long p0slab; int p0split;   // pass-0 shortcut: slabs of cmat^T . d agg
    int ldhx, ldH, ldM, ld3H, ldG, ldA, ldC, ldEh, ldAtt, ldEmb, ldM1, ldM2, ldNA, ldNC, ldCA,
        ldCC, ldZG, ldEa;
    long long hx[MAXP + 1];
    // AttGGNN only: hidden activations / dZ of the per-bond-type energy MLP, its output (edge
    // energies), and its first-layer input gradient
    long long aact[MAXP][MAXL], een[MAXP], adz[MAXP][MAXL], dxa, tmp_en, tmp_emb;
    long long eact[MAXP][MAXL], m[MAXP], agg[MAXP], gi[MAXP], gh[MAXP];
    long long att_act[MAXL], en, emb_act[MAXL], embo, add1_act[MAXL], add1o, conn1_act[MAXL],
        conn1o;
    long long cat_add, cat_conn, gemb, add2_act[MAXL], conn2_act[MAXL], term2_act[MAXL];
    long long dzA, dzC, dzT, dcat_add, dcat_conn, dgemb, zpart_g, zpart_a, zpart_c, dh, dh2, dxe;
    long long dhb, dhc, dhd;           // per-sibling input gradients of the node-level readout stacks
    // dZ of the hidden layers (same shapes as the activations): kept separate from the activations
    // so that every weight-gradient GEMM can be deferred and batched off the critical path
    long long edz[MAXP][MAXL], att_dz[MAXL], emb_dz[MAXL], add1_dz[MAXL], conn1_dz[MAXL],
        add2_dz[MAXL], conn2_dz[MAXL], term2_dz[MAXL], dagg[MAXP];
    // packed weight images of the resident-activation chains (gi_chain.hip): [msg, energy] stacks,
    // forward and backward layouts; 0 floats when the stack does not fit the chain kernel
    long long img_f[2], img_b[2], img_f_n[2], img_b_n[2], img_b_stride[2];
    long long chain_amax[2];            // fp16x2 chains: amax cells of the stack's weights, [layer][bond type] (gi_chain_params.x2_wamax)
    long long img_fx[2], chain_amax_f[2];   // the same for the FORWARD chain of the message rows (gi_chain_params.x2_rows32)
    // fp16x2 weight gradients of the message / energy stacks (round 6): amax cells of their operands, published by the
    // fp16x2 chain kernels — [pass][0: layer inputs X_l, 1: dZ_l][layer] (msg_cells)
    long long msg_amax[2];
    long long amax_pool;                // cells for weight-gradient operands nobody publishes a maximum of (AmaxPool)
    // AlphaDropout training mode: the workspace is allocated twice; float i of the second half holds
    // the backward factor d y / d z of activation i of the first (0: mode off)
    long long fshift;
    // split-K slabs of the skinny, long-reduction layers of the graph-level stacks (skinny_splits)
    long long skinny, skinny_floats;
    long long bf3, bf3_floats;          // bf16x3 operand images of the node-level stacks' wide layers (gi_gemm_bf3.hip)
    long long amax;                     // fp16x2 launches: max |.| of their operand tensors, 3 (of 4) amax cells per layer (gi_x2.h)
    long long total;
};

bool chain_fits(const Mlp& q, int dx_cols);
long long chain_image_floats(const Mlp& q, int groups, bool backward, long long* stride);
// Round 6: EVERY weight-gradient problem of a backward can run as an fp16x2 launch.  Operands whose largest magnitude
// no kernel publishes (GRU gate gradients, aggregated messages, h, the stacks' first / last layers, the graph-level
// stacks at small batches) get it from one gi_absmax launch per hand-over, on the stream of the weight-gradient launches
// and in front of them: a cell per distinct tensor from this pool, zeroed once per backward call.
constexpr int AMAX_POOL_CELLS = 160;
// amax cells of one pass of one stack family: [0: layer inputs, 1: dZ][layer]
constexpr long long MSG_CELLS_PER_PASS = 2LL * GI_CHAIN_MAXL * GI_AMAX_WORDS;
inline float* msg_cell(float* pass_base, int which, int layer) {
    return pass_base ? pass_base + ((long long)which * GI_CHAIN_MAXL + layer) * GI_AMAX_WORDS : nullptr;
}

// Layers of the node-level readout stacks that run on the bf16 MFMA pipe (gi_gemm_bf3.hip: fp32 operands split
// three ways, fp32 accumulate — the same result to ~3e-7): wide enough for 128 x 128 tiles in both directions.
// GI_BF3=0 keeps every GEMM on v_mfma_f32_32x32x2_f32.
// (rows: measured crossover against the fp32 kernel ~2 300 — 2 048 rows 35.0 vs 33.5 us, 3 000 rows 43.2 vs 53.5)
constexpr int BF3_MIN_WIDTH = 192, BF3_MIN_ROWS = 2560;
static int g_bf3 = -1;                  // -1: not read yet
static int g_x2 = -1;                   // fp16x2 instead of bf16x3 on those launches (environment GI_X2, default 1)
static bool x2_enabled() {
    if (g_x2 < 0) g_x2 = getenv("GI_X2") ? (atoi(getenv("GI_X2")) != 0) : 1;
    return g_x2 != 0;
}
bool bf3_enabled() {
    if (g_bf3 < 0) g_bf3 = getenv("GI_BF3") ? (atoi(getenv("GI_BF3")) != 0) : (GI_BF3_DEFAULT != 0);
    return g_bf3 != 0;
}
// The message / energy stacks' chain launches as fp16x2 (64-row blocks on the f16 pipe, gi_chain.hip): with the other
// 16-bit-pipe launches (GI_BF3, GI_X2) — the dZ chains of the BACKWARD only.  The forward keeps the fp32 chain: its
// rows are bit-independent of one another, which the pass-0 row cache, the equality of blocking and host-sync-free
// forwards and of forwards with and without a tape rely on (the fp16x2 chain scales activations per 64-row block, so a
// row's last bits depend on its block).  GI_CHAIN_X2=0: the fp32-MFMA chain in the backward too.
static bool chain_x2_enabled(bool call_x2) {
    static const int v = getenv("GI_CHAIN_X2") ? atoi(getenv("GI_CHAIN_X2")) : 1;
    return v != 0 && call_x2 && x2_enabled() && bf3_enabled();
}
// ... and, round 5, the FORWARD chains through the row-independent kernel (gi_chain_x2r_kernel: every row scaled by
// itself, so the properties listed above hold bit for bit — tests/test_kernels_gpu.py).  GI_CHAIN_FWD_X2=0: fp32.
static bool chain_fwd_x2_enabled(bool call_x2) {
    static const int v = getenv("GI_CHAIN_FWD_X2") ? atoi(getenv("GI_CHAIN_FWD_X2")) : 1;
    return v != 0 && chain_x2_enabled(call_x2);
}
bool bf3_wide(const Mlp& q, int l) { return q.fan_in(l) >= BF3_MIN_WIDTH && q.fan_out(l) >= BF3_MIN_WIDTH; }
bool bf3_layer_ok(const Mlp& q, int l) { return bf3_enabled() && bf3_wide(q, l); }

// A forward / dgrad problem with few output tiles and a long reduction (the first layer of fAddNet2 at
// the ChEMBL shape: 250 x 500 outputs, K = N*A + G = 9 252 — 32 workgroups walking 290 k tiles each,
// 165 us) runs split-K into slabs and gets its epilogue from gi_slab_epilogue: 16 x 32 workgroups of
// 18 k tiles.  Returns the number of splits (1: leave the problem alone).
int skinny_splits(int rows, int cols, int K) {
    const int tiles = gi_cdiv(rows, 64) * gi_cdiv(cols, 64);
    if (tiles >= 256 || K < 2048) return 1;
    return std::min(16, std::max(2, std::min(K / 512, 768 / tiles)));
}
long long skinny_floats(int rows, int cols, int K) {
    const int n = skinny_splits(rows, cols, K);
    return n > 1 ? (long long)n * gi_r4l((long long)rows * gi_r4(cols)) : 0;
}

void make_ws(const Model& m, int S, int E, int U, int D0, Ws& w) {
    const gi_ggnn_dims& d = m.d;
    memset(&w, 0, sizeof(w));
    w.R = S + 1; w.E = E; w.U = U; w.B = d.B;
    w.D0 = d.passes > 0 ? D0 : 0;       // pass-0 class rows (both models)
    w.ldhx = gi_r4(d.H + d.Fn); w.ldH = gi_r4(d.H); w.ldM = gi_r4(d.M); w.ld3H = gi_r4(3 * d.H);
    w.ldG = gi_r4(d.G); w.ldA = gi_r4(d.A); w.ldC = gi_r4(d.C); w.ldEh = gi_r4(d.enn_hidden);
    w.ldAtt = gi_r4(d.att_hidden); w.ldEmb = gi_r4(d.emb_hidden); w.ldM1 = gi_r4(d.mlp1_hidden);
    w.ldM2 = gi_r4(d.mlp2_hidden); w.ldNA = gi_r4(m.NA); w.ldNC = gi_r4(m.NC);
    w.ldCA = gi_r4(m.NA + d.G); w.ldCC = gi_r4(m.NC + d.G); w.ldZG = gi_r4(2 * d.G);
    const bool attn = d.kind == GI_KIND_ATTGGNN;
    w.ldEa = attn ? gi_r4(d.eatt_hidden) : 4;
    long long o = 0;
    auto take = [&](long long rows, int ld) { long long r = o; o += gi_r4l(rows * ld); return r; };
    const long long R = w.R, B = d.B, Er = std::max(U, 1);   // message-row buffers
    for (int p = 0; p <= d.passes; ++p) w.hx[p] = take(R, w.ldhx);
    const long long D0r = std::max(w.D0, 1);
    for (int p = 0; p < d.passes; ++p) {
        const long long Ep = (p == 0 && w.D0 > 0) ? D0r : Er;       // pass 0 runs on the D0 class rows
        for (int l = 0; l < d.enn_depth; ++l) w.eact[p][l] = take(Ep, w.ldEh);
        w.m[p] = take(Ep, w.ldM);
        if (attn) {
            for (int l = 0; l < d.eatt_depth; ++l) w.aact[p][l] = take(Ep, w.ldEa);
            w.een[p] = take(Ep, w.ldM);
        }
        w.agg[p] = take(R, w.ldM);
        w.gi[p] = take(R, w.ld3H);
        w.gh[p] = take(R, w.ld3H);
    }
    for (int l = 0; l < d.att_depth; ++l) w.att_act[l] = take(R, w.ldAtt);
    w.en = take(R, w.ldG);
    for (int l = 0; l < d.emb_depth; ++l) w.emb_act[l] = take(R, w.ldEmb);
    w.embo = take(R, w.ldG);
    for (int l = 0; l < d.mlp1_depth; ++l) w.add1_act[l] = take(R, w.ldM1);
    w.add1o = take(R, w.ldA);
    for (int l = 0; l < d.mlp1_depth; ++l) w.conn1_act[l] = take(R, w.ldM1);
    w.conn1o = take(R, w.ldC);
    w.cat_add = take(B, w.ldCA);
    w.cat_conn = take(B, w.ldCC);
    w.gemb = take(B, w.ldG);
    for (int l = 0; l < d.mlp2_depth; ++l) w.add2_act[l] = take(B, w.ldM2);
    for (int l = 0; l < d.mlp2_depth; ++l) w.conn2_act[l] = take(B, w.ldM2);
    for (int l = 0; l < d.mlp2_depth; ++l) w.term2_act[l] = take(B, w.ldM2);
    // backward scratch
    w.dzA = take(B, w.ldNA); w.dzC = take(B, w.ldNC); w.dzT = take(B, 4);
    w.dcat_add = take(B, w.ldCA); w.dcat_conn = take(B, w.ldCC); w.dgemb = take(B, w.ldG);
    w.zpart_g = take(B, w.ldZG); w.zpart_a = take(B, w.ldA); w.zpart_c = take(B, w.ldC);
    w.dh = take(R, w.ldH); w.dh2 = take(R, w.ldH); w.dxe = take(Er, w.ldH);
    w.dhb = take(R, w.ldH); w.dhc = take(R, w.ldH); w.dhd = take(R, w.ldH);
    for (int p = 0; p < d.passes; ++p) {
        const long long Ep = (p == 0 && w.D0 > 0) ? D0r : Er;
        for (int l = 0; l < d.enn_depth; ++l) w.edz[p][l] = take(Ep, w.ldEh);
        if (attn)
            for (int l = 0; l < d.eatt_depth; ++l) w.adz[p][l] = take(Ep, w.ldEa);
        w.dagg[p] = take(R, w.ldM);
    }
    if (w.D0 > 0 && !attn) { // split-K slabs of dm0 = cmat^T . dagg0: ~256 workgroups over the R rows
        const int tiles = gi_cdiv(w.D0, 64) * gi_cdiv(d.M, 64);
        const int kt = gi_cdiv((int)R, 32);
        w.p0split = std::min(std::max(256 / tiles, 1), std::max(1, kt / 2));
        w.p0slab = take((long long)w.p0split * gi_r4l(D0r * w.ldM), 1);
    }
    if (attn) {
        w.dxa = take(Er, w.ldH);
        w.tmp_en = take(std::max(E, 1), w.ldM);      // per-edge softmax-backward contributions
        w.tmp_emb = take(std::max(E, 1), w.ldM);
    }
    for (int l = 0; l < d.att_depth; ++l) w.att_dz[l] = take(R, w.ldAtt);
    for (int l = 0; l < d.emb_depth; ++l) w.emb_dz[l] = take(R, w.ldEmb);
    for (int l = 0; l < d.mlp1_depth; ++l) { w.add1_dz[l] = take(R, w.ldM1); w.conn1_dz[l] = take(R, w.ldM1); }
    for (int l = 0; l < d.mlp2_depth; ++l) {
        w.add2_dz[l] = take(B, w.ldM2); w.conn2_dz[l] = take(B, w.ldM2); w.term2_dz[l] = take(B, w.ldM2);
    }
    {   // graph-level stacks: forward layers (K = fan_in) and dgrad layers (K = fan_out), B rows each;
        // one launch holds at most one problem per stack
        const Mlp* t2[3] = {&m.add2, &m.conn2, &m.term2};
        long long need = 0;
        for (const Mlp* q : t2) {
            long long worst = 0;
            for (int l = 0; l < q->layers(); ++l) {
                worst = std::max(worst, skinny_floats(d.B, q->fan_out(l), q->fan_in(l)));
                worst = std::max(worst, skinny_floats(d.B, q->fan_in(l), q->fan_out(l)));
            }
            need += worst;
        }
        w.skinny_floats = need;
        w.skinny = take(std::max(need, 4LL), 1);
    }
    {   // images of either direction share the region (the backward re-packs)
        const Mlp* t1[4] = {&m.att, &m.emb, &m.add1, &m.conn1};
        long long elems = 0;
        for (const Mlp* q : t1)
            for (int l = 0; l < q->layers(); ++l)
                if (bf3_wide(*q, l))          // (sized whether or not the switch is on)
                    elems += std::max(gi_bf3_image_elems(q->fan_out(l), q->fan_in(l)),
                                      gi_bf3_image_elems(q->fan_in(l), q->fan_out(l)));
        w.bf3_floats = elems / 2;
        w.bf3 = take(std::max(w.bf3_floats, 4LL), 1);
        w.amax = take(4LL * GI_AMAX_WORDS * GI_BF3_PACK_MAX, 1);
    }
    for (int k = 0; k < (attn ? 2 : 1); ++k) {
        const Mlp& q = k ? m.eatt[0] : m.msg[0];
        if (d.passes > 0 && !d.dropout && chain_fits(q, d.H)) {   // (dropout: layer by layer)
            w.img_f_n[k] = chain_image_floats(q, d.Fe, false, nullptr);
            w.img_b_n[k] = chain_image_floats(q, d.Fe, true, &w.img_b_stride[k]);
            w.img_f[k] = take(w.img_f_n[k], 1);
            w.img_b[k] = take(w.img_b_n[k], 1);
            w.chain_amax[k] = take((long long)GI_AMAX_WORDS * GI_CHAIN_MAXL * GI_MAX_GROUPS, 1);
            w.img_fx[k] = take(w.img_f_n[k], 1);               // fp16x2 forward image (every forward chain, pass 0 included)
            w.chain_amax_f[k] = take((long long)GI_AMAX_WORDS * GI_CHAIN_MAXL * GI_MAX_GROUPS, 1);
            w.msg_amax[k] = take(MSG_CELLS_PER_PASS * (long long)d.passes, 1);
        }
    }
    w.amax_pool = take((long long)AMAX_POOL_CELLS * GI_AMAX_WORDS, 1);
    w.fshift = d.dropout ? gi_r4l(o) : 0;
    w.total = d.dropout ? 2 * gi_r4l(o) : o;
}

// ---- gi_graph.wcache: what a forward derives from the weights alone, kept across forwards -----------------------
// [per stack family: fp16x2 forward image | its max |W| cells [layer][bond type]] [max |W| cell of every node-level
// 16-bit-pipe layer].  Sizes depend on the model only (not on the batch).
struct Wc { long long img_fx[2], chain_amax_f[2], bf3_wamax, total; };
void wcache_layout(const Model& m, Wc& c) {
    memset(&c, 0, sizeof(c));
    const gi_ggnn_dims& d = m.d;
    long long o = 0;
    auto take = [&](long long n) { long long r = o; o += gi_r4l(n); return r; };
    for (int k = 0; k < (d.kind == GI_KIND_ATTGGNN ? 2 : 1); ++k) {
        const Mlp& q = k ? m.eatt[0] : m.msg[0];
        if (d.passes > 0 && chain_fits(q, d.H)) {
            c.img_fx[k] = take(chain_image_floats(q, d.Fe, false, nullptr));
            c.chain_amax_f[k] = take((long long)GI_AMAX_WORDS * GI_CHAIN_MAXL * GI_MAX_GROUPS);
        } else {
            c.img_fx[k] = c.chain_amax_f[k] = -1;
        }
    }
    c.bf3_wamax = take((long long)GI_AMAX_WORDS * GI_BF3_PACK_MAX);
    c.total = o;
}

// ---- wgrad slab plan ----------------------------------------------------------------------------
struct SlabEntry { long long off, stride; int nsplit, calls, done, n_out, n_in, ld, tn, bidx, launched, reduced, bf3, single_last, sep, x2all; };   // x2all: fp16x2 launch when the call has (or can make) the operands' amax cells (plan_slabs)   // sep: bias column by gi_bias_slabs, reduced with the last batch   // single_last: the last call wrote ONE slab (the pass-0 rows, defer_wgrad)
struct SlabPlan {
    SlabEntry e[160];
    long long total;
};

// wgrad launch shape: 64x64 output tiles.  Weight-gradient GEMMs are deferred and launched in
// batches of up to 8 problems, so ONE problem only needs ~256 workgroups (x its share of a
// type-grouped launch); fewer splits = fewer slabs to write and reduce.
// Weight gradients on the bf16 pipe (gi_gemm_b3p.hip, weight-gradient layout): hidden-layer problems of the node-level
// readout stacks — both dimensions >= BF3_MIN_WIDTH, reduction over >= BF3_MIN_ROWS node rows.  Tiles are 128 x 256
// and one workgroup owns a CU, so the slab count is chosen for equal tiles of ~29 k steps (920 rows): at the
// headline batch 8 slabs -> a 500 x 501 problem is 64 workgroups, a 250 x 251 one 16, and four big problems (or two
// big + six small) make one full round of the device (launch_wgrad_batches packs them that way).
// The bond-type-grouped message stacks (reduction over a type's message rows; decided by the rows of all types
// together) and the graph-level stacks (reduction over the B graphs) join from BF3_WGRAD_SMALL_MIN rows on: their
// launches are a fraction of a round of whole-CU workgroups beside the message passes' dZ chains, which measured a
// LOSS at the headline batch (8.4 k message rows, B = 1000: step 2.09 -> 2.16 ms) and a gain at 26 k message rows
// (ZINC shape 4.38 -> 4.34 ms).
constexpr int BF3_WGRAD_SLAB_ROWS = 920, BF3_WGRAD_MIN_ROWS = 768, BF3_WGRAD_SHORT_SLAB_ROWS = 336;
constexpr int BF3_WGRAD_SMALL_MIN = 16000;
bool bf3_wgrad_ok(int n_out, int n_in, int red_rows, int min_width = BF3_MIN_WIDTH) {
    return bf3_enabled() && gi_b3p_enable(-1) && n_out >= min_width && n_in >= min_width &&
           red_rows >= BF3_WGRAD_MIN_ROWS;
}
int bf3_wgrad_nsplit(int red_rows, int decide_rows, int slab_rows = 0) {
    const int L = slab_rows > 0 ? slab_rows : (decide_rows >= BF3_MIN_ROWS ? BF3_WGRAD_SLAB_ROWS : BF3_WGRAD_SHORT_SLAB_ROWS);
    return std::max(1, (red_rows + L / 2) / L);
}
// Round 6: the message / energy stacks' weight gradients as fp16x2 launches of gi_b3p_kernel.  What they lacked was the
// largest magnitude of their operands (the stacks' activations and dZ, written by the chain kernels): the fp16x2 chain
// kernels now publish them (gi_chain_layer.out_amax, gi_chain_params.x_amax) whenever BOTH directions of a stack run
// on those kernels — 410 us of the 594-us weight-gradient queue that bounds the backward of the message passes
// (profiles/r05) were these problems on the fp32 MFMA.  And every OTHER weight gradient too (wgrad_x2_all_possible):
// operands without a publishing producer get their cell from gi_absmax (Run::AmaxPool).
// Their launches are a few dozen workgroups each (250 x 251 outputs = two 128 x 256 tiles per slab), i.e. latency-bound:
// SHORT slabs (GI_MSG_SLAB_ROWS, default below) trade slab traffic for workgroups.  GI_MSG_WGRAD_X2=0 / GI_WGRAD_X2_ALL=0:
// as before.
constexpr int X2ALL_MIN_WIDTH = 32, X2ALL_MIN_ROWS = 512;
// GI_P0_GRU_MAIN (default 1): pass 0's GRU weight gradients run in the main queue's last launch (gi_ggnn_backward_phase)
bool p0_gru_on_main() {
    static const int v = getenv("GI_P0_GRU_MAIN") ? atoi(getenv("GI_P0_GRU_MAIN")) : 1;
    return v != 0;
}
int msg_slab_rows() {
    static const int v = getenv("GI_MSG_SLAB_ROWS") ? std::max(64, atoi(getenv("GI_MSG_SLAB_ROWS"))) : 460;
    return v;
}
int x2all_slab_rows(int decide_rows) { return decide_rows >= BF3_MIN_ROWS ? msg_slab_rows() : BF3_WGRAD_SHORT_SLAB_ROWS; }
bool msg_wgrad_x2_possible(const Model& m, bool call_x2);
// MEASURED (round 6, profiles/r06/ab_wgrad_x2_all.txt) AND NOT THE DEFAULT: with every weight gradient on the 16-bit pipe the
// headline step is 2.02 ms against 1.886 (the 128 x 128-tile kernel; 2.04-2.06 through the pipelined one): a launch
// of either kernel is >= 30 us whatever its size (per workgroup ~2 us per 32-deep k tile: the fp32 -> 2 x fp16 split is
// VALU work the small problems cannot hide), and the operands' gi_absmax passes add 15-40 us per hand-over, while the
// fp32 kernel runs the same small problems as 1 500 light workgroups.  What IS on by default: the stacks' hidden layers
// through the chain kernels' cells (msg_wgrad_x2_possible, -0.8 %), and the pool for round-4-rule problems that lack a
// producer's cell (a wide first / last layer: fp16x2 instead of the spilling bf16x3 instantiation).
bool wgrad_x2_all_enabled() {
    static const int v = getenv("GI_WGRAD_X2_ALL") ? atoi(getenv("GI_WGRAD_X2_ALL")) : 0;
    return v != 0;
}
bool wgrad_x2_pool_possible(const Model& m, bool call_x2) {
    return !m.d.dropout && call_x2 && x2_enabled() && bf3_enabled() && gi_b3p_enable(-1);
}
bool wgrad_x2_all_possible(const Model& m, bool call_x2) { return wgrad_x2_all_enabled() && wgrad_x2_pool_possible(m, call_x2); }
bool msg_wgrad_x2_possible(const Model& m, bool call_x2) {
    static const int v = getenv("GI_MSG_WGRAD_X2") ? atoi(getenv("GI_MSG_WGRAD_X2")) : 1;
    return v != 0 && !m.d.dropout && m.d.passes > 0 && chain_fwd_x2_enabled(call_x2) && gi_b3p_enable(-1) &&
           chain_fits(m.msg[0], m.d.H) && (m.d.kind != GI_KIND_ATTGGNN || chain_fits(m.eatt[0], m.d.H));
}

// The bias gradient as its own launch (GiBiasSlab, gi_common.h) when the "ones" column would start a new column of
// 64-wide tiles: n_in % 64 == 0 (GRU projections and first layers at H = 128: 129 columns = 3 tiles for 2 tiles' worth
// of work).  fp32-MFMA weight gradients only (the 16-bit-pipe kernel's 256-wide tiles hold 501 columns either way).
// The bias columns of ALL such problems of a backward are written by ONE launch (two in the two-call backward) on the
// weight-gradient queue, and their parameters are reduced with the last batch.
// MEASURED AND NOT ADOPTED (default off; GI_WGRAD_BIAS=1 switches it on; tools/experiments/README.md): the tiles it
// removes are worth ~60 us of weight-gradient kernel time per step, but one bias launch per hand-over (five per step,
// ~10 us each beside the GEMMs) lost 30 us (1.955 against 1.923 ms), and the single launch ties (1.952-1.961 against
// 1.945-1.964 ms; ZINC shape 4.06-4.09 against 4.09-4.11): more, smaller split-K slabs to write and reduce eat the rest.
bool wgrad_sep_bias(int n_in) {
    static const bool on = getenv("GI_WGRAD_BIAS") && atoi(getenv("GI_WGRAD_BIAS")) != 0;
    return on && n_in >= 64 && (n_in & 63) == 0;
}

void wgrad_shape(int n_out, int n_in, int red_rows, double share, int& tn, int& nsplit) {
    // 64x64 output tiles; 128x128 tiles for the big square weight gradients (a quarter of the slabs)
    // measured slower: 2.73 against 2.64 ms per step (tools/experiments/README.md)
    // (measurement aids, re-run in round 5: GI_WGRAD_TN=2 -> 128 x 128 tiles for problems of at least 192 x 192,
    // GI_WGRAD_WGS=<n> -> workgroups per problem)
    static const int env_tn = getenv("GI_WGRAD_TN") ? atoi(getenv("GI_WGRAD_TN")) : 1;
    static const int env_wgs = getenv("GI_WGRAD_WGS") ? atoi(getenv("GI_WGRAD_WGS")) : 0;
    tn = (env_tn == 2 && n_out >= 192 && n_in >= 192) ? 2 : 1;
    if (env_tn == 12) tn = 12;                          // 64 x 128 tiles (tm = 1, tn = 2) for every problem
    const int tiles = tn == 12 ? gi_cdiv(n_out, 64) * gi_cdiv(wgrad_sep_bias(n_in) ? n_in : n_in + 1, 128)
                               : gi_cdiv(n_out, 64 * tn) * gi_cdiv(wgrad_sep_bias(n_in) ? n_in : n_in + 1, 64 * tn);
    const int kt = gi_cdiv(std::max(red_rows, 1), 32);
    // workgroups per problem, measured in round 2: 96 -> 2.44-2.51 ms per step, 128 -> 2.39-2.40,
    // 192 -> 2.34-2.35, 256 -> 2.37, 384 -> 2.40-2.41
    const double wgs = env_wgs > 0 ? (double)env_wgs : 192.0;
    const int want = (int)(wgs * share / tiles + 0.5);
    nsplit = std::min(std::max(want, 1), std::max(1, kt / 2));
}

void plan_slabs(const Model& m, int S, int E, const int* Et, SlabPlan& sp) {   // E, Et: message rows
    const gi_ggnn_dims& d = m.d;
    memset(&sp, 0, sizeof(sp));
    long long o = 0;
    int maxEt = 0;
    for (int t = 0; t < d.Fe; ++t) maxEt = std::max(maxEt, Et ? Et[t] : E);
    // x2all (round 6): a problem the round-4 rule leaves on the fp32 MFMA becomes an fp16x2 launch of gi_b3p_kernel when
    // the call can give both operands an amax cell (defer_wgrad; without them it runs on the fp32 MFMA with THIS plan's
    // slab count).  Such launches are a few dozen 128 x 256-tile workgroups, i.e. latency-bound: SHORT slabs.
    const bool allx = wgrad_x2_all_possible(m, x2_enabled());
    auto add = [&](int widx, int bidx, int n_out, int n_in, int red, int calls, double share, bool bf3 = false,
                   int decide_rows = -1, bool gathered = false, bool chain_cells = false) {
        SlabEntry& e = sp.e[widx];
        e.bidx = bidx; e.launched = 0; e.reduced = 0;
        e.n_out = n_out; e.n_in = n_in; e.ld = gi_r4(n_in + 1); e.calls = calls; e.done = 0;
        wgrad_shape(n_out, n_in, red, share, e.tn, e.nsplit);
        if (decide_rows < 0) decide_rows = red;
        e.bf3 = bf3 && !d.dropout && bf3_wgrad_ok(n_out, n_in, decide_rows);
        if (e.bf3) e.nsplit = bf3_wgrad_nsplit(red, decide_rows);
        e.x2all = 0;
        if ((allx || chain_cells) && !e.bf3 && n_out >= X2ALL_MIN_WIDTH && n_in >= X2ALL_MIN_WIDTH && decide_rows >= X2ALL_MIN_ROWS &&
            (!gathered || (allx && msg_wgrad_x2_possible(m, x2_enabled())))) {
            e.x2all = 1;
            e.nsplit = bf3_wgrad_nsplit(red, decide_rows, x2all_slab_rows(decide_rows));
        }
        e.stride = gi_r4l((long long)n_out * e.ld);
        e.off = o;
        o += e.stride * e.nsplit * calls;
    };
    // stack: a message / energy stack — first layer gathered, the others' operand maxima published by the fp16x2 chains
    const bool msg_cells = msg_wgrad_x2_possible(m, x2_enabled());
    auto add_mlp = [&](const Mlp& q, int red, int calls, double share = 1.0, bool bf3 = false, int decide_rows = -1,
                       bool stack = false) {
        for (int l = 0; l < q.layers(); ++l)
            add(q.w(l), q.b(l), q.fan_out(l), q.fan_in(l), red, calls, share, bf3, decide_rows, stack && l == 0,
                stack && l > 0 && msg_cells);
    };
    const int R = S + 1;
    for (int t = 0; t < d.Fe; ++t) {
        const int et = Et ? Et[t] : E / d.Fe;
        static const int force_msg = getenv("GI_B3W_MSG") ? atoi(getenv("GI_B3W_MSG")) : -1;   // (measurement aid)
        const bool m3 = force_msg >= 0 ? force_msg != 0 : E >= BF3_WGRAD_SMALL_MIN;
        add_mlp(m.msg[t], et, d.passes, E > 0 ? 1.5 * (double)et / E : 1.0, m3, E, true);
        if (d.kind == GI_KIND_ATTGGNN)
            add_mlp(m.eatt[t], et, d.passes, E > 0 ? 1.5 * (double)et / E : 1.0, m3, E, true);
    }
    add(m.gru_wih, m.gru_bih, 3 * d.H, d.M, R, d.passes, 1.0);
    add(m.gru_whh, m.gru_bhh, 3 * d.H, d.H, R, d.passes, 1.0);
    add_mlp(m.att, R, 1, 1.0, true); add_mlp(m.emb, R, 1, 1.0, true); add_mlp(m.add1, R, 1, 1.0, true);
    add_mlp(m.conn1, R, 1, 1.0, true);
    static const int force_g = getenv("GI_B3W_G") ? atoi(getenv("GI_B3W_G")) : -1;             // (measurement aid)
    const bool g3 = force_g >= 0 ? force_g != 0 : d.B >= BF3_WGRAD_SMALL_MIN / 4;
    add_mlp(m.add2, d.B, 1, 1.0, g3); add_mlp(m.conn2, d.B, 1, 1.0, g3); add_mlp(m.term2, d.B, 1, 1.0, g3);
    sp.total = o;
}

// ---- launch helpers -----------------------------------------------------------------------------
struct Grp { int n; const int* off; int max_rows; const int* host_rows = nullptr; int dim_slot = 1; };   // dim_slot: which device dim holds the row-block height of a bounded chain launch (1: message rows, 2: pass-0 rows)   // n == 0: ungrouped; host_rows: per-group upper bounds (null: max_rows)

struct SideStream;

struct Run {
    hipStream_t st;
    const float* const* P;
    int rc;
    SideStream* side = nullptr;    // optional second stream for the weight-gradient GEMMs
    float* img_f[2] = {nullptr, nullptr};   // packed chain weight images [msg, energy stack]; null: the
    float* img_b[2] = {nullptr, nullptr};   // stack runs layer by layer
    float* chain_amax[2] = {nullptr, nullptr};   // != null: the stack's chains run as fp16x2 (gi_chain.hip), images packed that way
    float* img_fx[2] = {nullptr, nullptr};       // != null: EVERY forward chain of the stack (message rows, pass-0 rows, the row-cache
    float* chain_amax_f[2] = {nullptr, nullptr}; // insert) runs as row-independent fp16x2 (gi_chain_params.x2_rows32) from this image /
                                                 // these cells; img_f then aliases it (the fp32 image is not packed)
    long long img_b_stride[2] = {0, 0};
    // amax cells of the CURRENT pass's message-row launches, per stack family (null: pass-0 class rows, fp32 chains,
    // dropout): the fp16x2 chain kernels publish into them, defer_stack_wgrads hands them to the fp16x2
    // weight-gradient launches (msg_cell)
    float* msg_cells[2] = {nullptr, nullptr};
    // weight-gradient operands without a published maximum (AMAX_POOL_CELLS): pool of cells in ws, the tensors that
    // already have one in this call, the stream the pool lives on (zeroed there at first use)
    struct AmaxPool {
        float* base = nullptr; int used = 0; hipStream_t st = nullptr; bool zeroed = false;
        struct Key { const float* x; int rows, cols, ld; float* cell; } have[AMAX_POOL_CELLS];
    } pool;
    bool wgrad_x2 = false;                  // this call: weight gradients without cells may take fp16x2 through the pool
    const struct Mlp* eatt0 = nullptr;      // identifies the energy stacks (second image)
    bool hold_kicks = false;                // no weight-gradient launches on the side stream for now
    bool p0_on_main = false;                // everything queued went to the side stream before the pass-0 dZ chain; the
                                            // pass-0 rows' own (one-slab) problems follow their chain on the main stream
    // bounded (host-sync-free) forward: the sizes of the graph are on the device (gi_compact_bound) —
    // dims[0] = R, dims[1] / dims[2] = row-block heights of the message / pass-0 chains; d0_dev = D0.
    // Host-side S, E, U, D0 are then BOUNDS that size buffers and grids only.
    const int* dims = nullptr;
    const int* d0_dev = nullptr;
    int R_bound = 0;
    const int* skip = nullptr;              // set around the pass-0 stack launch: hit flag of gi_graph.p0_cache
    // layers that run as bf16x3 launches in this call (bf3_prepare): weight -> W^T image (backward; the forward
    // reads the fp32 weight as stored, GI_GEMM_BF3B_F32, img = NULL)
    // amax (fp16x2, gi_x2.h; NULL: the launch stays bf16x3): [0] max |W|, [1] max |layer input|, [2] max |dZ of the
    // layer's output| — written by gi_absmax / by the c_amax of the launch that produces the tensor
    // in_ok / dz_ok: the cell has a PRODUCER in this model — the forward launch of the layer below (c_amax of add_fwd)
    // and the dgrad launch of the layer above (c_amax of add_dgrad).  A stack's first layer reads h (GRU gate kernel)
    // and its last layer's dZ comes from the loss / gather backward: nobody measures those, so fp16x2 launches that
    // would need them stay bf16x3 (round-4 advisor finding: a zeroed cell reads as scale 1).
    struct Bf3 { const float* W; const unsigned short* img; float* amax; float* wamax; bool in_ok, dz_ok; } bf3[GI_BF3_PACK_MAX];   // wamax: the max |W| cell (amax[0], or its place in gi_graph.wcache)
    float* wc_bf3 = nullptr;                // gi_graph.wcache: the node-level layers' max |W| cells ...
    bool wc_valid = false;                  // ... and whether the cache's contents match the weights (nothing to derive)
    int nbf3 = 0;
    const Bf3* bf3_layer(const float* W, int rows) const {
        if (rows < BF3_MIN_ROWS) return nullptr;
        for (int i = 0; i < nbf3; ++i)
            if (bf3[i].W == W) return &bf3[i];
        return nullptr;
    }
    // AlphaDropout training mode (gnn/modules.py:130-142 with p > 0)
    bool drop = false;
    unsigned long long seed = 0;
    long long fshift = 0;                   // activation -> its stored backward factor (floats)
    int pass = 0;                           // message pass being evaluated (part of the site id)
    // scratch for split-K slabs of skinny problems inside one batched launch (reset by flush_batch)
    float* skinny = nullptr;
    long long skinny_floats = 0, skinny_used = 0;
    bool x2 = true;                         // fp16x2 on this call's 16-bit-pipe launches (GI_X2 and not GI_RUN_NO_X2 / GI_BWD_NO_X2)
    int* guard = nullptr;                   // gi_graph.x2_guard / x2_guard_host: the fp16x2 dynamic-range guard
    int* guard_host = nullptr;
    struct SlabPlan* sp = nullptr; // backward only: where the wgrad slabs go and what they reduce to
    float* slabs = nullptr;
    float* const* grads = nullptr;
    bool ok() const { return rc == 0; }
    void chk(int r) { if (rc == 0 && r != 0) rc = r; }
};

// forward / dgrad tile: at this path's sizes (10^3..10^4 rows, 100..700 columns, K <= 700) the
// 64x64 tile wins everywhere measured (tools/bench_gemm.py): it keeps 4 blocks per CU resident
// and more than one block per CU in flight; 64x128 only pays beyond a few thousand tiles.
void pick_tile(int rows, int ncols, int& tm, int& tn) {
    tm = 1;
    const long long b11 = (long long)gi_cdiv(rows, 64) * gi_cdiv(ncols, 64);
    tn = (ncols > 64 && b11 > 4096) ? 2 : 1;
}

void gemm_defaults(gi_gemm_params& p) {
    memset(&p, 0, sizeof(p));
    p.nsplit = 1;
    p.ones_col = -1;
}

// AlphaDropout behind layer l of a stack, in place on its SELU outputs y[rows, fan_out(l)]; the site id
// (weight index of the layer, message pass) selects the mask stream.  No-op outside dropout mode.
void drop_site(Run& r, const Mlp& q, int l, int pass, float* y, int ld, int rows, long long fshift) {
    if (!r.drop || !r.ok() || rows <= 0) return;
    gi_dropout_params dp;
    r.chk(gi_dropout_setup(q.drop_p, r.seed, (unsigned)(q.w(l) * MAXP + pass), &dp));
    if (r.ok()) r.chk(gi_alpha_dropout_fwd(y, ld, rows, q.fan_out(l), fshift, &dp, r.st));
}

// dgrad epilogue factor: selu'(through the stored activation), or the stored factor in dropout mode
void dact(const Run& r, gi_gemm_params& p, const float* act, int ldact, bool accumulate) {
    p.act = act ? act + r.fshift : nullptr; p.ldact = ldact;
    p.flags = (act ? (r.drop ? GI_EPI_MULACT : GI_EPI_DSELU) : 0) | (accumulate ? GI_EPI_ACCUM : 0);
}

// Y[rows, out] = (selu)(X[a_idx][rows, in] W^T + b); group t uses mlps[t]'s layer l
void linear_fwd(Run& r, const Mlp* mlps, int l, const Grp& g, const float* X, int ldx,
                const int* a_idx, int rows, float* Y, int ldy) {
    if (!r.ok() || rows <= 0) return;
    gi_gemm_params p;
    gemm_defaults(p);
    const Mlp& q = mlps[0];
    p.A = X; p.lda = ldx; p.a_idx = a_idx;
    p.C = Y; p.ldc = ldy;
    p.M = rows; p.N = q.fan_out(l); p.K = q.fan_in(l); p.ldb = q.fan_in(l);
    p.flags = GI_EPI_BIAS | GI_EPI_SELU;
    if (g.n) {
        p.ngroups = g.n; p.grp_off = g.off; p.max_group_rows = g.max_rows;
        for (int t = 0; t < g.n; ++t) { p.Bg[t] = r.P[mlps[t].w(l)]; p.biasg[t] = r.P[mlps[t].b(l)]; }
    } else {
        p.B = r.P[q.w(l)]; p.bias = r.P[q.b(l)];
    }
    pick_tile(g.n ? g.max_rows : rows, p.N, p.tm, p.tn);
    r.chk(gi_gemm(&p, r.st));
}

// dX[rows, ncols] (+)= (dZ[rows, n_out] W[n_out, n_in][:, :ncols]) (* selu'(act)); W itself is the B operand
// (reduction-major: the GEMM's "major" layout)
const float* dgrad_operand(const Run& r, int widx, int n_out, int n_in, gi_gemm_params& p) {
    (void)n_out;
    p.ldb = n_in; p.b_major = 1;
    return r.P[widx];
}

void linear_dgrad(Run& r, const int* widx, const Grp& g, int n_out,
                  int n_in, int ncols, const float* dZ, int lddz, int rows, float* dX, int lddx,
                  const float* act, int ldact, bool accumulate) {
    if (!r.ok() || rows <= 0) return;
    gi_gemm_params p;
    gemm_defaults(p);
    p.A = dZ; p.lda = lddz; p.C = dX; p.ldc = lddx;
    p.M = rows; p.N = ncols; p.K = n_out;
    dact(r, p, act, ldact, accumulate);
    if (g.n) {
        p.ngroups = g.n; p.grp_off = g.off; p.max_group_rows = g.max_rows;
        for (int t = 0; t < g.n; ++t) p.Bg[t] = dgrad_operand(r, widx[t], n_out, n_in, p);
    } else {
        p.B = dgrad_operand(r, widx[0], n_out, n_in, p);
    }
    pick_tile(g.n ? g.max_rows : rows, ncols, p.tm, p.tn);
    r.chk(gi_gemm(&p, r.st));
}

bool chain_fits(const Mlp& q, int dx_cols);
void chain_fwd_params(gi_chain_params& c, const Run& r, float* ws, const Mlp* mlps, const Grp& g,
                      const float* X, int ldx, const int* idx, int rows, const long long* acts,
                      int ldh, float* final_dst, int ld_final);
// Pass 0 has a few hundred distinct message rows (one per atom kind and bond type): its chain launch is a dozen
// workgroups that each stream their bond type's whole ~1 MB weight image, 57 us forward and 125 us backward on the
// critical path for next to no arithmetic (profiles/r04/x2/critical_path_amax_cells.txt).  Layer by layer the same rows
// are 5 (4) grouped launches of a dozen 64 x 64 tiles: measured a tie in the forward (50 us), and in the backward the
// four dgrad launches end 30-40 us before the chain would (they run beside pass 1's weight gradients either way).
// GI_P0_LAYERWISE: bit 0 forward, bit 1 backward.  Default 0: with the pass-0 weight gradients in one slab the chain won the A/B at all three shapes (profiles/r04/p0).
bool p0_layerwise(const Run& r, const Grp& g, int rows, bool backward) {
    static const int v = getenv("GI_P0_LAYERWISE") ? atoi(getenv("GI_P0_LAYERWISE")) : 0;     // bit 0: forward, bit 1: backward
    return (v & (backward ? 2 : 1)) && g.dim_slot == 2 && !r.dims && rows <= 2048;
}

void mlp_forward(Run& r, float* ws, const Mlp* mlps, const Grp& g, const float* X, int ldx,
                 const int* a_idx, int rows, const long long* acts, int ldh, float* final_dst,
                 int ld_final) {
    const int L = mlps[0].layers();
    if (g.n && r.ok() && rows > 0 && r.img_f[mlps == r.eatt0 ? 1 : 0] && ldx >= gi_r4(mlps[0].in) &&
        !p0_layerwise(r, g, rows, false)) {
        gi_chain_params c;                      // the whole stack in one resident-activation launch
        chain_fwd_params(c, r, ws, mlps, g, X, ldx, a_idx, rows, acts, ldh, final_dst, ld_final);
        r.chk(gi_mlp_chain(&c, 1, r.st));
        return;
    }
    for (int l = 0; l < L; ++l) {
        const float* src = (l == 0) ? X : ws + acts[l - 1];
        float* dst = (l == L - 1) ? final_dst : ws + acts[l];
        linear_fwd(r, mlps, l, g, src, l == 0 ? ldx : ldh, l == 0 ? a_idx : nullptr, rows, dst,
                   l == L - 1 ? ld_final : ldh);
        drop_site(r, mlps[0], l, r.pass, dst, l == L - 1 ? ld_final : ldh, rows, r.fshift);
    }
}

// ---- batched ("horizontally fused") execution of sibling MLPs -----------------------------------
// The readout's four node-level stacks (att, emb, fAddNet1, fConnNet1), its three graph-level
// stacks (fAddNet2, fConnNet2, fTermNet2) and the two GRU projections are mutually independent;
// layer l of every sibling goes into ONE gi_gemm_batch launch (and in backward one wgrad launch +
// one dgrad launch per layer level).
struct MlpJob {
    const Mlp* mlp;
    const float* X; int ldx; int rows;           // input (first `fan_in(0)` columns are used)
    const long long* acts; int ldh;              // hidden activation buffers (ws offsets)
    float* out; int ldout;                       // forward: last layer's destination
    const long long* dzs;                        // backward: dZ buffers of the hidden layers
    const float* Zlast; int ldz;                 // backward: dZ of the last layer
    float* dX; int lddx; int dx_cols; bool accumulate;   // backward: first-layer input gradient (or null)
    long long out_fshift = 0;                    // dropout mode: factor twin of `out` when it is not in ws
};

struct SlabEpilogue {               // gi_slab_epilogue arguments of a problem that went split-K
    const float* slabs; int nsplit; long long stride; int rows, cols, ld, flags;
    const float* bias; const float* act; int ldact; float* out; int ldo;
};

struct Batch {
    gi_gemm_params p[8];
    SlabEpilogue post[8];
    int n = 0, npost = 0;
    gi_gemm_params& next() { gemm_defaults(p[n]); return p[n++]; }
};

// Turns the forward / dgrad problem p (output p.C[M, N], epilogue p.flags) into a split-K problem
// writing slabs when it is skinny and long (skinny_splits) and scratch is left; the epilogue moves to
// a gi_slab_epilogue launch behind the batch.
void maybe_split_k(Batch& b, Run& r, gi_gemm_params& p) {
    const int ns = skinny_splits(p.M, p.N, p.K);
    if (ns <= 1 || p.ngroups || p.a_idx || b.npost >= 8) return;
    if (p.c_amax) return;       // (the slabs' epilogue publishes no amax, and the GEMM's own would be that of partial sums)
    const int ld = gi_r4(p.N);
    const long long stride = gi_r4l((long long)p.M * ld), need = stride * ns;
    if (!r.skinny || r.skinny_used + need > r.skinny_floats) return;
    SlabEpilogue& e = b.post[b.npost++];
    e = SlabEpilogue{r.skinny + r.skinny_used, ns, stride, p.M, p.N, ld, p.flags, p.bias, p.act, p.ldact,
                     p.C, p.ldc};
    p.C = r.skinny + r.skinny_used; p.ldc = ld;
    p.flags = GI_GEMM_SPLITK; p.bias = nullptr; p.act = nullptr;
    p.nsplit = ns; p.c_split_stride = stride;
    r.skinny_used += need;
}

// Weight-gradient GEMMs only feed the final slab reduction, so they are collected here while the
// dZ chain (the critical path) runs and launched afterwards in batches of 8 problems.
struct Deferred {
    gi_gemm_params p[96];
    int widx[96][GI_MAX_GROUPS];     // weight indices each problem's slabs belong to
    int nw[96];
    unsigned char sep_bias[96];      // the problem's bias-gradient column is written by gi_bias_slabs, not by a ones column
    unsigned char want_amax[96];     // fp16x2 problem whose a_amax (bit 0) / b_amax (bit 1) cell comes from the pool at launch time
    GiBiasSlab bias[160];            // ... collected over the whole backward, launched once (flush_bias)
    int nbias = 0;
    int n = 0;
    gi_gemm_params& next() { gemm_defaults(p[n]); nw[n] = 0; sep_bias[n] = 0; want_amax[n] = 0; return p[n++]; }
};

void flush_batch(Run& r, Batch& b, bool wgrad) {
    if (!r.ok() || b.n == 0) { b.n = 0; b.npost = 0; r.skinny_used = 0; return; }
    // a launch is ONE arithmetic: fp16x2 problems, bf16x3 problems and fp32-MFMA problems each get their own
    {
        gi_gemm_params cls[2][8];
        int nc[2] = {0, 0}, nk = 0;
        for (int i = 0; i < b.n; ++i) {
            if (b.p[i].flags & GI_GEMM_BF3) { const int x = (b.p[i].flags & GI_GEMM_X2) ? 1 : 0; cls[x][nc[x]++] = b.p[i]; }
            else b.p[nk++] = b.p[i];                                    // (nk <= i: in-place compaction is safe)
        }
        for (int x = 1; x >= 0; --x)
            if (nc[x] && r.ok()) r.chk(gi_gemm_batch(cls[x], nc[x], r.st));
        b.n = nk;
    }
    if (!wgrad) {                               // common tile for the whole launch
        long long b11 = 0;
        for (int i = 0; i < b.n; ++i) b11 += (long long)gi_cdiv(b.p[i].M, 64) * gi_cdiv(b.p[i].N, 64);
        const int tn = b11 > 8192 ? 2 : 1;
        for (int i = 0; i < b.n; ++i) { b.p[i].tm = 1; b.p[i].tn = tn; }
    }
    if (b.n && r.ok()) r.chk(gi_gemm_batch(b.p, b.n, r.st));
    for (int i = 0; i < b.npost && r.ok(); ++i) {
        const SlabEpilogue& e = b.post[i];
        r.chk(gi_slab_epilogue(e.slabs, e.nsplit, e.stride, e.rows, e.cols, e.ld, e.flags, e.bias, e.act,
                               e.ldact, e.out, e.ldo, r.st));
    }
    b.n = 0; b.npost = 0;
    r.skinny_used = 0;          // (the next batch's slabs are written behind these epilogues in stream order)
}

void add_fwd(Batch& b, Run& r, const float* W, const float* bias, int in, int out, const float* X,
             int ldx, int rows, float* Y, int ldy, bool selu, const float* Wnext = nullptr) {
    if (rows <= 0) return;
    gi_gemm_params& p = b.next();
    p.A = X; p.lda = ldx; p.B = W; p.ldb = in; p.bias = bias; p.C = Y; p.ldc = ldy;
    p.M = rows; p.N = out; p.K = in;
    p.flags = GI_EPI_BIAS | (selu ? GI_EPI_SELU : 0);
    if (const Run::Bf3* e = r.bf3_layer(W, rows)) {
        p.flags |= GI_GEMM_BF3 | GI_GEMM_BF3B_F32;                             // (B = W [out][in] as stored)
        if (e->amax && e->in_ok) {
            p.flags |= GI_GEMM_X2; p.a_amax = e->amax + GI_AMAX_WORDS; p.b_amax = e->wamax;
            p.x2_guard = r.guard; p.x2_guard_host = r.guard_host;           // activation rows outside the per-tensor range
        }
    }
    if (Wnext)                                    // Y is the next layer's input: it needs max |Y| if it runs fp16x2
        if (const Run::Bf3* nx = r.bf3_layer(Wnext, rows))
            if (nx->amax) p.c_amax = nx->amax + GI_AMAX_WORDS;
    if (r.dims && rows == r.R_bound) { p.m_dev = r.dims; return; }     // node-level rows: counted on the device
    if (p.flags & GI_GEMM_BF3) return;
    maybe_split_k(b, r, p);
}

void add_dgrad(Batch& b, Run& r, int widx, int n_out, int n_in, int ncols, const float* dZ,
               int lddz, int rows, float* dX, int lddx, const float* act, int ldact,
               bool accumulate, int widx_prev = -1) {
    if (rows <= 0) return;
    gi_gemm_params& p = b.next();
    p.A = dZ; p.lda = lddz; p.B = dgrad_operand(r, widx, n_out, n_in, p); p.C = dX; p.ldc = lddx;
    p.M = rows; p.N = ncols; p.K = n_out;
    dact(r, p, act, ldact, accumulate);
    if (widx_prev >= 0)                           // dX is the dZ of the previous layer's output: its fp16x2 launches
        if (const Run::Bf3* pv = r.bf3_layer(r.P[widx_prev], rows))          // (dgrad, weight gradient) need max |dX|
            if (pv->amax) p.c_amax = pv->amax + 2 * GI_AMAX_WORDS;
    if (ncols == n_in)
        if (const Run::Bf3* e = r.bf3_layer(r.P[widx], rows))
            if (e->img) {                                                     // W^T as fp32 [n_in][r4(n_out)]
                p.B = reinterpret_cast<const float*>(e->img); p.b_major = 0; p.ldb = gi_r4(n_out);
                p.flags |= GI_GEMM_BF3 | GI_GEMM_BF3B_F32;
                if (e->amax && e->dz_ok) {
                    p.flags |= GI_GEMM_X2; p.a_amax = e->amax + 2 * GI_AMAX_WORDS; p.b_amax = e->wamax;
                    p.x2_guard = r.guard ? r.guard + 2 : nullptr;            // dZ rows: counted, never a trip (sums over rows)
                }
                return;
            }
    maybe_split_k(b, r, p);
}

void flush_deferred(Run& r, Deferred& q);
void kick_deferred(Run& r, Deferred& q, SideStream* side, bool all);
// queued weight-gradient problems that make up one hand-over to the side stream (one launch)
inline int wgrad_kick_n() {
    static const int n = [] { const char* e = getenv("GI_KICK_N"); const int v = e ? atoi(e) : 8; return (v >= 1 && v <= 8) ? v : 8; }();   // (measurement aid)
    return n;
}

gi_reduce_desc reduce_desc(const SlabEntry& e, float* slabs, float* const* grads, int widx) {
    gi_reduce_desc q;
    q.slabs = slabs + e.off; q.dW = grads[widx]; q.db = grads[e.bidx];
    q.slab_stride = e.stride; q.n_slabs = e.nsplit * e.calls - (e.single_last ? e.nsplit - 1 : 0); q.N = e.n_out; q.K = e.n_in;
    q.ld = e.ld;
    return q;
}

void defer_wgrad(Run& r, Deferred& q, SlabPlan& sp, float* slabs, const int* widx, const Grp& g,
                 const float* dZ, int lddz, const float* X, int ldx, const int* b_idx, int rows,
                 const float* dz_amax = nullptr, const float* x_amax = nullptr) {
    if (q.n == 96) flush_deferred(r, q);          // list full (very deep configurations only)
    gi_gemm_params& p = q.next();
    SlabEntry& e0 = sp.e[widx[0]];
    p.A = dZ; p.lda = lddz; p.a_major = 1;
    p.B = X; p.ldb = ldx; p.b_major = 1; p.b_idx = b_idx;
    p.M = e0.n_out; p.N = e0.n_in + 1; p.K = rows; p.ldc = e0.ld;
    p.ones_col = e0.n_in;
    p.flags = GI_GEMM_SPLITK;
    p.nsplit = e0.nsplit; p.c_split_stride = e0.stride;
    p.tm = e0.tn == 12 ? 1 : e0.tn; p.tn = e0.tn == 12 ? 2 : e0.tn;     // 1x1 (64x64 tiles) or 2x2 (128x128), see wgrad_shape
    // The pass-0 rows (a few dozen per bond type, the stack's last call): ONE slab instead of the plan's nsplit — split
    // nine ways the launch was 2 200 workgroups that mostly store zeros, 50 us + their share of the final reduction
    // behind the last dZ chain with nothing left to overlap (profiles/r04/x2/critical_path_amax_cells.txt)
    const bool one_slab = g.n && g.dim_slot == 2;
    // Which arithmetic (plan_slabs chose the slab count; a launch is one arithmetic, launch_wgrad_batches sorts them):
    //   bf3   (the round-4 rule: node-level hidden layers, message / graph-level stacks of big batches): 16-bit pipe;
    //         fp16x2 when both operands have an amax cell — the producers' (node-level layers, the fp16x2 chains) or,
    //         round 6, one from the pool (Run::AmaxPool: gi_absmax in front of the launch) — else bf16x3;
    //   x2all (round 6: everything else that is big enough): fp16x2 under the same condition, else the fp32 MFMA with
    //         the plan's slab count.
    // A gathered B (the first layer of a message stack reads h[u_src]) needs published cells: the pool's gi_absmax
    // would measure rows the launch does not read.
    if ((e0.bf3 || e0.x2all) && !one_slab) {
        const float* ca = dz_amax;
        const float* cb = x_amax;
        if (!g.n && !(ca && cb))
            if (const Run::Bf3* e = r.bf3_layer(r.P[widx[0]], rows))
                if (e->amax) { if (e->dz_ok) ca = e->amax + 2 * GI_AMAX_WORDS; if (e->in_ok) cb = e->amax + GI_AMAX_WORDS; }
        const bool pool = r.wgrad_x2 && !b_idx && (e0.bf3 || wgrad_x2_all_enabled());
        if ((ca && cb) || pool) {
            if (!b_idx || e0.x2all) {
                p.flags |= GI_GEMM_BF3 | GI_GEMM_X2;
                p.a_amax = ca; p.b_amax = cb;
                q.want_amax[q.n - 1] = (ca ? 0 : 1) | (cb ? 0 : 2);
                // which kernel: the pipelined 128 x 256 one (one workgroup per CU, ~30 us per launch whatever its size)
                // for the big node-level problems, 128 x 128 tiles / 256 threads / 32 KB for the many small ones
                // (GI_WGRAD_T128: 0 never, 1 the round-6 set, 2 every fp16x2 weight gradient)
                static const int t128 = getenv("GI_WGRAD_T128") ? atoi(getenv("GI_WGRAD_T128")) : 0;
                if (t128 == 2 || (t128 == 1 && !e0.bf3)) p.flags |= GI_GEMM_T128;
            }
        } else if (e0.bf3 && !b_idx) {
            p.flags |= GI_GEMM_BF3;
        }
    }
    const int slot = q.n - 1;
    if (!(p.flags & GI_GEMM_BF3) && !one_slab && wgrad_sep_bias(e0.n_in)) {   // plain n_out x n_in problem; db by gi_bias_slabs
        p.N = e0.n_in; p.ones_col = -1;                                  // (the pass-0 rows: a few dozen, one slab, keep the column)
        q.sep_bias[slot] = 1;
        for (int t = 0; t < (g.n ? g.n : 1); ++t) sp.e[widx[t]].sep = 1;
    }
    if (g.n) {
        p.ngroups = g.n; p.grp_off = g.off;
        for (int t = 0; t < g.n; ++t) {
            SlabEntry& e = sp.e[widx[t]];
            p.Cg[t] = slabs + e.off + (long long)e.done * e.nsplit * e.stride;
            p.gsplit[t] = e.nsplit;
            if (one_slab && e.done == e.calls - 1) { p.gsplit[t] = 1; e.single_last = 1; }
            e.done++;
            q.widx[slot][t] = widx[t];
        }
        q.nw[slot] = g.n;
    } else {
        p.C = slabs + e0.off + (long long)e0.done * e0.nsplit * e0.stride;
        e0.done++;
        q.widx[slot][0] = widx[0];
        q.nw[slot] = 1;
    }
    if (r.side && q.n >= wgrad_kick_n() && !r.hold_kicks) kick_deferred(r, q, r.side, false);
}

// one launch per tile class among (up to) 8 consecutive queued problems
void launch_wgrad_batch(Run& r, const gi_gemm_params* p, int n, hipStream_t st) {
    gi_gemm_params a[8], b[8];
    int na = 0, nb = 0;
    for (int i = 0; i < n; ++i) {
        if (p[i].tm == 1) a[na++] = p[i];
        else b[nb++] = p[i];
    }
    if (na && r.ok()) r.chk(gi_gemm_batch(a, na, st));
    if (nb && r.ok()) r.chk(gi_gemm_batch(b, nb, st));
}

// The amax cells the queued fp16x2 problems still lack (Deferred::want_amax): one cell of the pool per distinct operand
// tensor of this backward call, filled by gi_absmax launches on `st` in front of the GEMMs that read them.  A problem
// the pool has no cell left for goes back to the fp32 MFMA (its slab count stays the plan's).
void resolve_pool_amax(Run& r, gi_gemm_params* p, const unsigned char* want, int n, hipStream_t st) {
    Run::AmaxPool& pool = r.pool;
    gi_absmax_desc ad[GI_ABSMAX_MAX];
    int na = 0;
    auto flush = [&]() { if (na && r.ok()) r.chk(gi_absmax(ad, na, st)); na = 0; };
    auto cell_of = [&](const float* x, int rows, int cols, int ld) -> float* {
        for (int i = 0; i < pool.used; ++i) {
            const Run::AmaxPool::Key& k = pool.have[i];
            if (k.x == x && k.rows == rows && k.cols == cols && k.ld == ld) return k.cell;
        }
        if (!pool.base || pool.used >= AMAX_POOL_CELLS) return nullptr;
        if (!pool.zeroed) {
            r.chk((int)hipMemsetAsync(pool.base, 0, sizeof(float) * AMAX_POOL_CELLS * GI_AMAX_WORDS, st));
            pool.zeroed = true; pool.st = st;
        }
        float* cell = pool.base + (long long)pool.used * GI_AMAX_WORDS;
        pool.have[pool.used++] = Run::AmaxPool::Key{x, rows, cols, ld, cell};
        if (na == GI_ABSMAX_MAX) flush();
        ad[na].x = x; ad[na].rows = rows; ad[na].cols = cols; ad[na].ld = ld; ad[na].out = cell;
        ++na;
        return cell;
    };
    bool any = false;
    for (int i = 0; i < n; ++i) any |= want && want[i];
    if (!any) return;
    if (pool.zeroed && pool.st != st) {            // (a second stream joins: order it behind the pool's kernels so far)
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) {
            r.chk((int)hipEventRecord(e, pool.st));
            r.chk((int)hipStreamWaitEvent(st, e, 0));
            (void)hipEventDestroy(e);              // (destruction is deferred until the event has completed)
        }
        pool.st = st;
    }
    for (int i = 0; i < n; ++i) {
        if (!want[i]) continue;
        gi_gemm_params& q = p[i];
        const int bcols = q.ones_col >= 0 ? q.ones_col : q.N;
        if ((want[i] & 1) && !q.a_amax) q.a_amax = cell_of(q.A, q.K, q.M, q.lda);       // dZ [rows, n_out]
        if ((want[i] & 2) && !q.b_amax) q.b_amax = cell_of(q.B, q.K, bcols, q.ldb);     // X  [rows, n_in]
        if (!q.a_amax || !q.b_amax) {              // pool exhausted
            q.flags &= ~(GI_GEMM_BF3 | GI_GEMM_X2); q.a_amax = q.b_amax = nullptr;
        }
    }
    flush();
}

// consecutive queued problems, up to 8 per launch; the bf16x3 ones (one workgroup per CU, equal tiles) are packed
// separately, biggest first, into launches of about one round of the device
void launch_wgrad_batches(Run& r, Deferred& dq, const gi_gemm_params* p_in, const unsigned char* sep, int n, hipStream_t st) {
    gi_gemm_params p[96];
    for (int i = 0; i < n; ++i) p[i] = p_in[i];
    resolve_pool_amax(r, p, dq.want_amax, n, st);
    // b3[0]: bf16x3; fp16x2: b3[1 + 2 (128 x 128-tile kernel) + 1 (gathered B)] (a launch is ONE kernel instantiation)
    gi_gemm_params rest[96], b3[5][96];
    int nr = 0, n3[5] = {0, 0, 0, 0, 0};
    GiBiasSlab* const bias = dq.bias;
    int& nbias = dq.nbias;
    for (int i = 0; i < n; ++i) {
        if (p[i].flags & GI_GEMM_BF3) {
            const int x = (p[i].flags & GI_GEMM_X2) ? 1 + ((p[i].flags & GI_GEMM_T128) ? 2 : 0) + (p[i].b_idx ? 1 : 0) : 0;
            b3[x][n3[x]++] = p[i];
        }
        else rest[nr++] = p[i];
        if (sep && sep[i]) {                                      // bias-gradient column of this problem's slabs
            const gi_gemm_params& q = p[i];
            const int ng = q.ngroups ? q.ngroups : 1;
            for (int g = 0; g < ng; ++g) {
                if (nbias == 160) { r.chk(gi_bias_slabs(bias, nbias, st)); nbias = 0; }      // (very deep configurations only)
                GiBiasSlab& b = bias[nbias++];
                b.dZ = q.A; b.lddz = q.lda; b.grp_off = q.ngroups ? q.grp_off : nullptr; b.g = g;
                b.rows = q.K; b.n_out = q.M;
                b.slab = q.ngroups ? q.Cg[g] : q.C; b.stride = q.c_split_stride;
                b.ld = q.ldc; b.col = q.N; b.nsplit = q.ngroups ? q.gsplit[g] : q.nsplit;
            }
        }
    }
    auto tiles = [](const gi_gemm_params& q) {
        int zs = q.nsplit;
        if (q.ngroups) { zs = 0; for (int g = 0; g < q.ngroups; ++g) zs += q.gsplit[g]; }
        return gi_cdiv(q.M, 128) * gi_cdiv(q.N, 256) * zs;
    };
    static const int cus = [] {
        int dev = 0, c = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        return c;
    }();
    for (int x = 4; x >= 0; --x) {
        gi_gemm_params* q = b3[x];
        for (int i = 1; i < n3[x]; ++i)                           // stable insertion sort, most tiles first
            for (int j = i; j > 0 && tiles(q[j]) > tiles(q[j - 1]); --j) std::swap(q[j], q[j - 1]);
        const int cap = x >= 3 ? 1 << 30 : cus;                   // (the 128 x 128 kernel: several workgroups per CU, any size)
        for (int base = 0; base < n3[x] && r.ok();) {
            int k = 0, t = 0;
            while (base + k < n3[x] && k < 8 && (k == 0 || t + tiles(q[base + k]) <= cap)) { t += tiles(q[base + k]); ++k; }
            r.chk(gi_gemm_batch(q + base, k, st));
            base += k;
        }
    }
    for (int base = 0; base < nr && r.ok(); base += 8) launch_wgrad_batch(r, rest + base, std::min(8, nr - base), st);
}

// the bias-gradient columns of every problem launched so far without a ones column: one launch, on a stream that is
// ordered behind their dZ (the stream of their weight-gradient GEMMs)
void flush_bias(Run& r, Deferred& q, hipStream_t st) {
    if (q.nbias && r.ok()) r.chk(gi_bias_slabs(q.bias, q.nbias, st));
    q.nbias = 0;
}

void flush_deferred(Run& r, Deferred& q) {
    launch_wgrad_batches(r, q, q.p, q.sep_bias, q.n, r.st);
    q.n = 0;
}

// ---- weight gradients on a second stream --------------------------------------------------------
// The dZ chain is a sequence of short dependent launches that leave MFMA slots idle (few blocks per
// CU, all in prologue / epilogue at the same time); the weight-gradient GEMMs only feed the final
// slab reduction.  With a caller-provided side stream they are launched there, 8 problems at a
// time, as soon as their operands exist (event from the main stream), and run in the gaps of the
// chain; the main stream joins before gi_reduce_slabs.
struct SideStream {
    hipStream_t st;
    int used;
    static constexpr int NEV = 48;
    // one pool per device (events belong to the device that is current when they are created);
    // callers hold the GIL / call from one thread per process, like every entry point of this ABI
    static hipEvent_t* pool() {
        constexpr int MAXDEV = 16;
        static hipEvent_t ev[MAXDEV][NEV];
        static bool made[MAXDEV] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev = (dev >= 0 && dev < MAXDEV) ? dev : 0;
        if (!made[dev]) {
            for (hipEvent_t& e : ev[dev]) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
            made[dev] = true;
        }
        return ev[dev];
    }
    hipEvent_t next() { return pool()[used++ % NEV]; }
};

// launch every complete batch of 8 queued problems (all of them when `all`) on the side stream
void kick_deferred(Run& r, Deferred& q, SideStream* side, bool all) {
    if (!side || !r.ok()) return;
    const int kn = wgrad_kick_n();
    const int n = all ? q.n : (q.n / kn) * kn;
    if (n == 0) return;
    hipEvent_t ready = side->next();
    r.chk((int)hipEventRecord(ready, r.st));
    r.chk((int)hipStreamWaitEvent(side->st, ready, 0));
    launch_wgrad_batches(r, q, q.p, q.sep_bias, n, side->st);
    // parameters whose last slab has just been queued: reduce them right behind, on the side stream
    // too, so that only the final pass's gradients are left for the end of the backward
    gi_reduce_desc descs[96 * GI_MAX_GROUPS > 160 ? 160 : 96 * GI_MAX_GROUPS];
    int nd = 0;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < q.nw[i]; ++k) {
            SlabEntry& e = r.sp->e[q.widx[i][k]];
            if (++e.launched == e.calls && !e.reduced && !e.sep && nd < 160) {   // (sep: bias column still to come)
                e.reduced = 1;
                descs[nd++] = reduce_desc(e, r.slabs, r.grads, q.widx[i][k]);
            }
        }
    if (nd && r.ok()) r.chk(gi_reduce_slabs(descs, nd, side->st));
    for (int i = n; i < q.n; ++i) {
        q.p[i - n] = q.p[i];
        q.nw[i - n] = q.nw[i];
        q.sep_bias[i - n] = q.sep_bias[i];
        q.want_amax[i - n] = q.want_amax[i];
        for (int k = 0; k < q.nw[i]; ++k) q.widx[i - n][k] = q.widx[i][k];
    }
    q.n -= n;
}

void join_side(Run& r, SideStream* side) {
    if (!side || !r.ok()) return;
    hipEvent_t done = side->next();
    r.chk((int)hipEventRecord(done, side->st));
    r.chk((int)hipStreamWaitEvent(r.st, done, 0));
}

void mlp_jobs_forward(Run& r, float* ws, const MlpJob* jobs, int n) {
    int maxL = 0;
    for (int j = 0; j < n; ++j) maxL = std::max(maxL, jobs[j].mlp->layers());
    for (int l = 0; l < maxL; ++l) {
        Batch b;
        for (int j = 0; j < n; ++j) {
            const MlpJob& q = jobs[j];
            const int L = q.mlp->layers();
            if (l >= L) continue;
            const float* src = (l == 0) ? q.X : ws + q.acts[l - 1];
            float* dst = (l == L - 1) ? q.out : ws + q.acts[l];
            add_fwd(b, r, r.P[q.mlp->w(l)], r.P[q.mlp->b(l)], q.mlp->fan_in(l), q.mlp->fan_out(l), src,
                    l == 0 ? q.ldx : q.ldh, q.rows, dst, l == L - 1 ? q.ldout : q.ldh, true,
                    l + 1 < L ? r.P[q.mlp->w(l + 1)] : nullptr);
        }
        flush_batch(r, b, false);
        for (int j = 0; j < n && r.drop; ++j) {
            const MlpJob& q = jobs[j];
            const int L = q.mlp->layers();
            if (l >= L) continue;
            const bool last = l == L - 1;
            drop_site(r, *q.mlp, l, 0, last ? q.out : ws + q.acts[l], last ? q.ldout : q.ldh, q.rows,
                      (last && q.out_fshift) ? q.out_fshift : r.fshift);
        }
    }
}

// Layers are aligned from the END (step s handles layer L_j-1-s of job j).  dZ of hidden layer l
// goes to the job's dz buffer l; first-layer input gradients go to the jobs' (distinct) dX;
// weight gradients are deferred.
void mlp_jobs_backward(Run& r, float* ws, SlabPlan& sp, float* slabs, Deferred& dq,
                       const MlpJob* jobs, int n) {
    int maxL = 0;
    const Grp none{0, nullptr, 0};
    for (int j = 0; j < n; ++j) maxL = std::max(maxL, jobs[j].mlp->layers());
    for (int s = 0; s < maxL; ++s) {
        Batch bd;
        for (int j = 0; j < n; ++j) {
            const MlpJob& q = jobs[j];
            const int L = q.mlp->layers(), l = L - 1 - s;
            if (l < 0) continue;
            const float* dZ = (l == L - 1) ? q.Zlast : ws + q.dzs[l];
            const int lddz = (l == L - 1) ? q.ldz : q.ldh;
            const float* Xl = (l == 0) ? q.X : ws + q.acts[l - 1];
            const int widx = q.mlp->w(l);
            defer_wgrad(r, dq, sp, slabs, &widx, none, dZ, lddz, Xl, l == 0 ? q.ldx : q.ldh, nullptr,
                        q.rows);
            if (l > 0) {
                add_dgrad(bd, r, widx, q.mlp->fan_out(l), q.mlp->fan_in(l), q.mlp->fan_in(l), dZ, lddz,
                          q.rows, ws + q.dzs[l - 1], q.ldh, ws + q.acts[l - 1], q.ldh, false, q.mlp->w(l - 1));
            } else if (q.dX) {
                add_dgrad(bd, r, widx, q.mlp->fan_out(0), q.mlp->fan_in(0), q.dx_cols, dZ, lddz, q.rows,
                          q.dX, q.lddx, nullptr, 0, q.accumulate);
            }
        }
        flush_batch(r, bd, false);
    }
}

int chain_bwd_params(gi_chain_params& c, const Run& r, float* ws, const Mlp* mlps, const Grp& g,
                     const float* Zlast, int ldz, int rows, const long long* acts,
                     const long long* dzs, int ldh, float* dX, int lddx, int dx_cols);
void defer_stack_wgrads(Run& r, float* ws, SlabPlan& sp, float* slabs, Deferred& dq, const Mlp* mlps,
                        const Grp& g, const float* X, int ldx, const int* a_idx, int rows,
                        const long long* acts, const long long* dzs, int ldh, const float* Zlast,
                        int ldz);

void join_side(Run& r, SideStream* side);

// The bond-type-grouped message MLP: dZ chain now, weight gradients deferred.
// Zlast of a message stack = selu'(m) * (segmented sum of `vals` rows over the message CSR), formed in
// place over the stack's forward output by its own launch (gi_seg_sum_dselu_f) in front of the dZ chain
// (folding it into the chain launch was built and measured a tie in round 2: tools/experiments/README.md)
struct SegIn { const float* vals; int ld; const int* idx; const int* off; };

void seg_launch(Run& r, const SegIn* seg, int rows, int cols, float* y, int ldy) {
    if (seg && seg->vals && r.ok())
        r.chk(gi_seg_sum_dselu_f(seg->vals, seg->ld, seg->idx, seg->off, rows, cols, y, ldy, r.fshift, r.st));
}

void msg_backward(Run& r, float* ws, SlabPlan& sp, float* slabs, Deferred& dq, const Mlp* mlps,
                  const Grp& g, const float* X, int ldx, const int* a_idx, int rows,
                  const long long* acts, const long long* dzs, int ldh, const float* Zlast, int ldz,
                  float* dX, int lddx, int dx_cols, const SegIn* seg = nullptr) {
    const int L = mlps[0].layers();
    if (g.n && r.ok() && rows > 0 && r.img_b[mlps == r.eatt0 ? 1 : 0] && dx_cols == mlps[0].in &&
        ldz >= gi_r4(mlps[0].out) && !p0_layerwise(r, g, rows, true)) {
        gi_chain_params c;                      // the whole dZ chain in one launch, then the wgrads
        seg_launch(r, seg, rows, mlps[0].out, const_cast<float*>(Zlast), ldz);
        if (chain_bwd_params(c, r, ws, mlps, g, Zlast, ldz, rows, acts, dzs, ldh, dX, lddx, dx_cols))
            r.chk(gi_mlp_chain(&c, 1, r.st));
        r.hold_kicks = r.p0_on_main;
        defer_stack_wgrads(r, ws, sp, slabs, dq, mlps, g, X, ldx, a_idx, rows, acts, dzs, ldh, Zlast,
                           ldz);
        return;
    }
    seg_launch(r, seg, rows, mlps[0].out, const_cast<float*>(Zlast), ldz);
    for (int l = L - 1; l >= 0; --l) {
        const float* dZ = (l == L - 1) ? Zlast : ws + dzs[l];
        const int lddz = (l == L - 1) ? ldz : ldh;
        const float* Xl = (l == 0) ? X : ws + acts[l - 1];
        int widx[GI_MAX_GROUPS];
        for (int t = 0; t < g.n; ++t) widx[t] = mlps[t].w(l);
        defer_wgrad(r, dq, sp, slabs, widx, g, dZ, lddz, Xl, l == 0 ? ldx : ldh,
                    l == 0 ? a_idx : nullptr, rows);
        const Mlp& q = mlps[0];
        if (l > 0)
            linear_dgrad(r, widx, g, q.fan_out(l), q.fan_in(l), q.fan_in(l), dZ, lddz, rows,
                         ws + dzs[l - 1], ldh, ws + acts[l - 1], ldh, false);
        else if (dX)
            linear_dgrad(r, widx, g, q.fan_out(0), q.fan_in(0), dx_cols, dZ, lddz, rows, dX,
                         lddx, nullptr, 0, false);
    }
}

// ---- resident-activation chains (gi_chain.hip) ---------------------------------------------------
// The per-bond-type message / energy stacks run as ONE launch per direction when every layer fits
// the chain kernel (<= GI_CHAIN_MAXL layers, widths 4..GI_CHAIN_MAXW); GI_CHAIN=0 or wider stacks
// take the layer-by-layer GEMM path above (same arithmetic, more launches).
bool chain_enabled() {
    static const bool v = !(getenv("GI_CHAIN") && atoi(getenv("GI_CHAIN")) == 0);
    return v;
}

bool chain_fits(const Mlp& q, int dx_cols) {
    if (!chain_enabled() || q.layers() > GI_CHAIN_MAXL) return false;
    for (int l = 0; l < q.layers(); ++l)
        if (q.fan_in(l) < 4 || q.fan_in(l) > GI_CHAIN_MAXW || q.fan_out(l) < 4 ||
            q.fan_out(l) > GI_CHAIN_MAXW)
            return false;
    return dx_cols >= 4 && dx_cols <= GI_CHAIN_MAXW;
}

// skeleton (dims only) of a stack's chain, forward or backward order (backward: every layer, the
// first Linear's input gradient last)
void chain_dims(gi_chain_params& c, const Mlp& q, int groups, bool backward) {
    memset(&c, 0, sizeof(c));
    const int L = q.layers();
    c.nlayers = L; c.ngroups = groups; c.backward = backward ? 1 : 0;
    for (int i = 0; i < L; ++i) {
        const int l = backward ? L - 1 - i : i;
        c.layer[i].K = backward ? q.fan_out(l) : q.fan_in(l);
        c.layer[i].N = backward ? q.fan_in(l) : q.fan_out(l);
    }
}

long long chain_image_floats(const Mlp& q, int groups, bool backward, long long* stride) {
    gi_chain_params c;
    chain_dims(c, q, groups, backward);
    const long long n = gi_mlp_chain_image_floats(&c);
    if (stride) *stride = n > 0 ? n / groups : 0;
    return n > 0 ? n : 0;
}

// write the packed weight image of the grouped stacks `mlps` (once per forward / backward call)
void chain_pack(Run& r, const Mlp* mlps, int groups, bool backward, float* image) {
    if (!r.ok()) return;
    gi_chain_params c;
    chain_dims(c, mlps[0], groups, backward);
    const int L = mlps[0].layers();
    for (int i = 0; i < L; ++i)
        for (int t = 0; t < groups; ++t) c.layer[i].W[t] = r.P[mlps[t].w(backward ? L - 1 - i : i)];
    c.image = image;
    c.x2_wamax = r.chain_amax[mlps == r.eatt0 ? 1 : 0];
    r.chk(gi_mlp_chain_pack(&c, 1, r.st));
}

void chain_groups(gi_chain_params& c, const Grp& g, int rows) {
    c.grp_off = g.off; c.ngroups = g.n; c.rows = rows;
    for (int t = 0; t < g.n; ++t) c.group_rows[t] = g.host_rows ? g.host_rows[t] : g.max_rows;
}

// Y = MLP(X[idx]) for the grouped stacks `mlps`; hidden activations -> acts, last layer -> final_dst
void chain_fwd_params(gi_chain_params& c, const Run& r, float* ws, const Mlp* mlps, const Grp& g,
                      const float* X, int ldx, const int* idx, int rows, const long long* acts,
                      int ldh, float* final_dst, int ld_final) {
    memset(&c, 0, sizeof(c));
    c.image = r.img_f[mlps == r.eatt0 ? 1 : 0];
    const Mlp& q = mlps[0];
    const int L = q.layers();
    c.nlayers = L; c.X = X; c.ldx = ldx; c.x_idx = idx; c.backward = 0;
    chain_groups(c, g, rows);
    if (r.dims) c.tile_rows_dev = r.dims + g.dim_slot;
    c.x2_wamax = r.chain_amax[mlps == r.eatt0 ? 1 : 0];
    if (r.img_fx[mlps == r.eatt0 ? 1 : 0]) {                      // one row's result depends on that row only: blocking and
        c.image = r.img_fx[mlps == r.eatt0 ? 1 : 0];              // bounded launches, cached and recomputed pass-0 rows agree bit for bit
        c.x2_wamax = r.chain_amax_f[mlps == r.eatt0 ? 1 : 0];
        c.x2_rows32 = 1;
    }
    c.skip_flag = r.skip;
    float* const cells = (c.x2_rows32 && g.dim_slot != 2) ? r.msg_cells[mlps == r.eatt0 ? 1 : 0] : nullptr;
    c.x_amax = msg_cell(cells, 0, 0);
    for (int l = 0; l < L; ++l) {
        gi_chain_layer& y = c.layer[l];
        y.K = q.fan_in(l); y.N = q.fan_out(l);
        y.out = (l == L - 1) ? final_dst : ws + acts[l];
        y.ldo = (l == L - 1) ? ld_final : ldh;
        if (l < L - 1) y.out_amax = msg_cell(cells, 0, l + 1);      // = the input of layer l + 1
        for (int t = 0; t < g.n; ++t) { y.W[t] = r.P[mlps[t].w(l)]; y.bias[t] = r.P[mlps[t].b(l)]; }
    }
}

// dZ chain of the same stacks: Zlast = dZ of the last layer; dZ of hidden layer l -> dzs[l]; the
// first layer's input gradient (dx_cols leading columns) -> dX when dX != null.  Returns the number
// of chain layers (0: nothing to launch).
int chain_bwd_params(gi_chain_params& c, const Run& r, float* ws, const Mlp* mlps, const Grp& g,
                     const float* Zlast, int ldz, int rows, const long long* acts,
                     const long long* dzs, int ldh, float* dX, int lddx, int dx_cols) {
    memset(&c, 0, sizeof(c));
    const Mlp& q = mlps[0];
    const int L = q.layers();
    c.image = r.img_b[mlps == r.eatt0 ? 1 : 0];
    c.image_stride = r.img_b_stride[mlps == r.eatt0 ? 1 : 0];
    c.X = Zlast; c.ldx = ldz; c.x_idx = nullptr; c.backward = 1;
    c.x2_wamax = r.chain_amax[mlps == r.eatt0 ? 1 : 0];
    {   // (measurement aid) GI_CHAIN_BWD_X2R=1: the dZ chains through the row-independent kernel too (measured: +0.6 % at
        // the headline batch, -0.7 % ZINC shape, +1.4 % ChEMBL shape; for the pass-0 rows only: ties — profiles/r05/ab)
        static const bool bwd_x2r = getenv("GI_CHAIN_BWD_X2R") && atoi(getenv("GI_CHAIN_BWD_X2R"));
        if (bwd_x2r && c.x2_wamax) c.x2_rows32 = 1;
    }
    chain_groups(c, g, rows);
    float* const cells = (c.x2_wamax && g.dim_slot != 2) ? r.msg_cells[mlps == r.eatt0 ? 1 : 0] : nullptr;
    c.x_amax = msg_cell(cells, 1, L - 1);                   // Zlast = dZ of the last layer
    int n = 0;
    for (int l = L - 1; l >= 0; --l) {
        if (l == 0 && !dX) break;
        gi_chain_layer& y = c.layer[n++];
        y.K = q.fan_out(l);
        y.N = (l == 0) ? dx_cols : q.fan_in(l);
        if (l > 0) y.out_amax = msg_cell(cells, 1, l - 1);   // this layer writes dZ of layer l - 1
        // W_l is [fan_out][fan_in] row-major: for l == 0 only its leading dx_cols columns are used,
        // which needs fan_in == dx_cols (rows are read with stride N)
        y.out = (l == 0) ? dX : ws + dzs[l - 1];
        y.ldo = (l == 0) ? lddx : ldh;
        y.act = (l == 0) ? nullptr : ws + acts[l - 1];
        y.ldact = ldh;
        for (int t = 0; t < g.n; ++t) y.W[t] = r.P[mlps[t].w(l)];
    }
    c.nlayers = n;
    return n;
}

// weight gradients of every layer of a grouped stack (deferred), after its dZ chain was enqueued
void defer_stack_wgrads(Run& r, float* ws, SlabPlan& sp, float* slabs, Deferred& dq, const Mlp* mlps,
                        const Grp& g, const float* X, int ldx, const int* a_idx, int rows,
                        const long long* acts, const long long* dzs, int ldh, const float* Zlast,
                        int ldz) {
    const int L = mlps[0].layers();
    for (int l = L - 1; l >= 0; --l) {
        const float* dZ = (l == L - 1) ? Zlast : ws + dzs[l];
        const int lddz = (l == L - 1) ? ldz : ldh;
        const float* Xl = (l == 0) ? X : ws + acts[l - 1];
        int widx[GI_MAX_GROUPS];
        for (int t = 0; t < g.n; ++t) widx[t] = mlps[t].w(l);
        float* const cells = g.dim_slot != 2 ? r.msg_cells[mlps == r.eatt0 ? 1 : 0] : nullptr;
        defer_wgrad(r, dq, sp, slabs, widx, g, dZ, lddz, Xl, l == 0 ? ldx : ldh,
                    l == 0 ? a_idx : nullptr, rows, msg_cell(cells, 1, l), msg_cell(cells, 0, l));
    }
}

// ---- AttGGNN: the message MLP and the energy MLP of a pass are siblings (same input rows, same
// bond-type grouping): layer l of both goes into one launch, like the readout's sibling stacks.
struct SegIn;
void seg_launch(Run& r, const SegIn* seg, int rows, int cols, float* y, int ldy);

struct EdgeChain {
    const Mlp* mlps;               // [Fe] per-bond-type stacks
    const long long* acts;         // hidden activations (ws offsets)
    const long long* dzs;          // hidden dZ buffers (ws offsets)
    int ldh;
    float* out; int ldout;         // forward: last layer's output; backward: its dZ (in place)
    float* dX;                     // backward: first-layer input gradient [E, ldH] or null
    const struct SegIn* seg = nullptr;   // backward: `out` is first formed in place from these (see SegIn)
};

void grouped_problem(gi_gemm_params& p, const Grp& g) {
    p.ngroups = g.n; p.grp_off = g.off; p.max_group_rows = g.max_rows;
}

void edge_chains_forward(Run& r, float* ws, const EdgeChain* ch, int n, const Grp& g,
                         const float* X, int ldx, const int* a_idx, int rows) {
    if (n == 2 && r.ok() && rows > 0 && r.img_f[0] && r.img_f[1] && ldx >= gi_r4(ch[0].mlps[0].in)) {
        gi_chain_params c[2];                   // both stacks' whole forward in ONE launch
        for (int j = 0; j < 2; ++j)
            chain_fwd_params(c[j], r, ws, ch[j].mlps, g, X, ldx, a_idx, rows, ch[j].acts, ch[j].ldh,
                             ch[j].out, ch[j].ldout);
        r.chk(gi_mlp_chain(c, 2, r.st));
        return;
    }
    int maxL = 0;
    for (int j = 0; j < n; ++j) maxL = std::max(maxL, ch[j].mlps[0].layers());
    for (int l = 0; l < maxL; ++l) {
        Batch b;
        for (int j = 0; j < n; ++j) {
            const EdgeChain& c = ch[j];
            const Mlp& q = c.mlps[0];
            const int L = q.layers();
            if (l >= L) continue;
            gi_gemm_params& p = b.next();
            p.A = (l == 0) ? X : ws + c.acts[l - 1];
            p.lda = (l == 0) ? ldx : c.ldh;
            p.a_idx = (l == 0) ? a_idx : nullptr;
            p.C = (l == L - 1) ? c.out : ws + c.acts[l];
            p.ldc = (l == L - 1) ? c.ldout : c.ldh;
            p.M = rows; p.N = q.fan_out(l); p.K = q.fan_in(l); p.ldb = q.fan_in(l);
            p.flags = GI_EPI_BIAS | GI_EPI_SELU;
            grouped_problem(p, g);
            for (int t = 0; t < g.n; ++t) {
                p.Bg[t] = r.P[c.mlps[t].w(l)]; p.biasg[t] = r.P[c.mlps[t].b(l)];
            }
        }
        flush_batch(r, b, false);
        for (int j = 0; j < n && r.drop; ++j) {
            const EdgeChain& c = ch[j];
            const int L = c.mlps[0].layers();
            if (l >= L) continue;
            drop_site(r, c.mlps[0], l, r.pass, l == L - 1 ? c.out : ws + c.acts[l],
                      l == L - 1 ? c.ldout : c.ldh, rows, r.fshift);
        }
    }
}

// layers aligned from the end; weight gradients deferred
void edge_chains_backward(Run& r, float* ws, SlabPlan& sp, float* slabs, Deferred& dq,
                          const EdgeChain* ch, int n, const Grp& g, const float* X, int ldx,
                          const int* a_idx, int rows, int lddx, int dx_cols) {
    if (n == 2 && r.ok() && rows > 0 && r.img_b[0] && r.img_b[1] && dx_cols == ch[0].mlps[0].in &&
        dx_cols == ch[1].mlps[0].in && ch[0].ldout >= gi_r4(ch[0].mlps[0].out) &&
        ch[1].ldout >= gi_r4(ch[1].mlps[0].out)) {
        gi_chain_params c[2];                   // both dZ chains in ONE launch, then the wgrads
        int nl[2];
        for (int j = 0; j < 2; ++j) {
            nl[j] = chain_bwd_params(c[j], r, ws, ch[j].mlps, g, ch[j].out, ch[j].ldout, rows,
                                     ch[j].acts, ch[j].dzs, ch[j].ldh, ch[j].dX, lddx, dx_cols);
            seg_launch(r, ch[j].seg, rows, ch[j].mlps[0].out, ch[j].out, ch[j].ldout);
        }
        if (nl[0] && nl[1]) r.chk(gi_mlp_chain(c, 2, r.st));
        else if (nl[0]) r.chk(gi_mlp_chain(&c[0], 1, r.st));
        else if (nl[1]) r.chk(gi_mlp_chain(&c[1], 1, r.st));
        r.hold_kicks = r.p0_on_main;
        for (int j = 0; j < 2; ++j)
            defer_stack_wgrads(r, ws, sp, slabs, dq, ch[j].mlps, g, X, ldx, a_idx, rows, ch[j].acts,
                               ch[j].dzs, ch[j].ldh, ch[j].out, ch[j].ldout);
        return;
    }
    for (int j = 0; j < n; ++j) seg_launch(r, ch[j].seg, rows, ch[j].mlps[0].out, ch[j].out, ch[j].ldout);
    int maxL = 0;
    for (int j = 0; j < n; ++j) maxL = std::max(maxL, ch[j].mlps[0].layers());
    for (int s = 0; s < maxL; ++s) {
        Batch bd;
        for (int j = 0; j < n; ++j) {
            const EdgeChain& c = ch[j];
            const Mlp& q = c.mlps[0];
            const int L = q.layers(), l = L - 1 - s;
            if (l < 0) continue;
            const float* dZ = (l == L - 1) ? c.out : ws + c.dzs[l];
            const int lddz = (l == L - 1) ? c.ldout : c.ldh;
            const float* Xl = (l == 0) ? X : ws + c.acts[l - 1];
            int widx[GI_MAX_GROUPS];
            for (int t = 0; t < g.n; ++t) widx[t] = c.mlps[t].w(l);
            defer_wgrad(r, dq, sp, slabs, widx, g, dZ, lddz, Xl, l == 0 ? ldx : c.ldh,
                        l == 0 ? a_idx : nullptr, rows);
            if (l == 0 && !c.dX) continue;
            gi_gemm_params& p = bd.next();
            p.A = dZ; p.lda = lddz; p.M = rows; p.K = q.fan_out(l);
            grouped_problem(p, g);
            for (int t = 0; t < g.n; ++t)
                p.Bg[t] = dgrad_operand(r, widx[t], q.fan_out(l), q.fan_in(l), p);
            if (l > 0) {
                p.C = ws + c.dzs[l - 1]; p.ldc = c.ldh; p.N = q.fan_in(l);
                dact(r, p, ws + c.acts[l - 1], c.ldh, false);
            } else {
                p.C = c.dX; p.ldc = lddx; p.N = dx_cols;
            }
        }
        flush_batch(r, bd, false);
    }
}

}  // namespace

// ================================ C ABI ==========================================================
// A stream of the lowest priority the device offers, for gi_ggnn_backward's side_stream: the
// weight-gradient GEMMs should only fill what the dZ chain (on the caller's stream) leaves idle.
extern "C" int gi_side_stream_create(void** out) {
    if (!out) return GI_EINVAL;
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return (int)e;
    hipStream_t st = nullptr;
    if (getenv("GI_SIDE_PRIO") && atoi(getenv("GI_SIDE_PRIO")) == 0) least = (least + greatest) / 2;   // (measurement aid: not the lowest)
    e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least);
    if (e != hipSuccess) return (int)e;
    *out = (void*)st;
    return 0;
}

extern "C" int gi_side_stream_destroy(void* stream) {
    if (!stream) return 0;
    return (int)hipStreamDestroy((hipStream_t)stream);
}

extern "C" int gi_ggnn_num_params(const gi_ggnn_dims* d) {
    Model m;
    const int rc = build_model(d, m);
    return rc ? rc : m.nparams;
}

// the layers of this call that run as bf16x3 / fp16x2 launches.  Builds the table the launches look their layer up in
// (r.bf3: W^T image for the backward — the forward stages the fp32 weights as stored: 4 bytes per element through L2
// instead of the image's 6, 84 -> 76 us per launch — and the amax cells) and, per `what`, enqueues on `st`:
//   BF3_DO_AMAX  zero the cells, max |W| of every layer (gi_absmax), the weights' dynamic-range check (the forward),
//   BF3_DO_PACK  the W^T images (the backward, or the forward on its side stream: GI_RUN_PREPACK_BWD).
// `backward` selects which table is left in r (a forward that prepacks calls this twice: images first, then its own).
enum { BF3_DO_AMAX = 1, BF3_DO_PACK = 2 };
void bf3_prepare(Run& r, const Model& m, float* ws, const Ws& w, bool backward, int rows, int what, hipStream_t st) {
    r.nbf3 = 0;
    if (!bf3_enabled() || r.drop || w.bf3_floats <= 0) return;
    // Too few rows for the 16-bit pipe: nothing to prepare — EXCEPT the max |W| cells of gi_graph.wcache when this forward
    // is the one that derives the cache: a later, larger batch of the same weights will find it marked valid (round 6:
    // a B = 1 forward followed by a B = 1000 one read cells nobody had written).
    const bool derive_only = rows < BF3_MIN_ROWS;
    if (derive_only && !(r.wc_bf3 && !r.wc_valid && (what & BF3_DO_AMAX) && !backward)) return;
    const Mlp* t1[4] = {&m.att, &m.emb, &m.add1, &m.conn1};
    gi_bf3_pack_desc d[GI_BF3_PACK_MAX];
    gi_absmax_desc ad[GI_BF3_PACK_MAX], wd[GI_BF3_PACK_MAX];       // (flattened for gi_absmax; [out][in] for the guard)
    unsigned short* img = reinterpret_cast<unsigned short*>(ws + w.bf3);
    float* am = ws + w.amax;
    const bool x2 = r.x2 && gi_b3p_enable(-1);
    long long used = 0;
    int n = 0;
    for (const Mlp* q : t1)
        for (int l = 0; l < q->layers() && n < GI_BF3_PACK_MAX; ++l) {
            if (!bf3_layer_ok(*q, l)) continue;
            const int fi = q->fan_in(l), fo = q->fan_out(l);
            d[n].W = r.P[q->w(l)]; d[n].ld = fi; d[n].transpose = backward ? 1 : 0;
            d[n].rows = backward ? fi : fo; d[n].cols = backward ? fo : fi;
            d[n].image = img + used; d[n].as_f32 = 1;     // (W^T as fp32: 4 bytes per element through L2, not 6)
            r.bf3[n].W = d[n].W; r.bf3[n].img = backward ? d[n].image : nullptr;
            r.bf3[n].in_ok = l > 0; r.bf3[n].dz_ok = l + 1 < q->layers();
            // fp16x2 (gi_x2.h): three amax cells per layer — max |W| (gi_absmax, in the forward), max |layer input| and
            // max |dZ of its output| (the c_amax of the launches that produce them; zeroed once per forward: the
            // backward runs on the same workspace)
            r.bf3[n].amax = x2 ? am + 4LL * GI_AMAX_WORDS * n : nullptr;
            r.bf3[n].wamax = !x2 ? nullptr : (r.wc_bf3 ? r.wc_bf3 + (long long)GI_AMAX_WORDS * n : r.bf3[n].amax);
            ad[n].x = d[n].W; ad[n].rows = 1; ad[n].cols = fi * fo; ad[n].ld = fi * fo; ad[n].out = r.bf3[n].wamax;
            wd[n].x = d[n].W; wd[n].rows = fo; wd[n].cols = fi; wd[n].ld = fi; wd[n].out = r.bf3[n].wamax;
            used += std::max(gi_bf3_image_elems(fo, fi), gi_bf3_image_elems(fi, fo));
            ++n;
        }
    if (n && backward && (what & BF3_DO_PACK) && !derive_only) r.chk(gi_bf3_pack(d, n, st));
    if (n && x2 && (what & BF3_DO_AMAX)) {
        if (!derive_only) r.chk((int)hipMemsetAsync(am, 0, sizeof(float) * 4 * GI_AMAX_WORDS * n, st));      // (the activations' / dZ cells: every forward)
        if (!(r.wc_bf3 && r.wc_valid)) {                  // max |W| + the weights' dynamic-range check: unless cached
            if (r.wc_bf3) r.chk((int)hipMemsetAsync(r.wc_bf3, 0, sizeof(float) * GI_AMAX_WORDS * n, st));
            r.chk(gi_absmax(ad, n, st));
            if (r.guard && r.ok()) r.chk(gi_x2_weight_guard(wd, n, r.guard + 1, r.guard_host, st));
        }
    }
    r.nbf3 = derive_only ? 0 : n;
}

// "The weight images a forward prepacked for its backward are written" (GI_RUN_PREPACK_BWD -> GI_BWD_PREPACKED), PER
// WORKSPACE: the forward stamps (ws, arithmetic flavour) into a small registry and records the entry's event on the side
// stream behind its packs; a backward handed GI_BWD_PREPACKED looks its ws up — found with the same flavour: it waits
// for that event instead of packing; NOT found (the flag was set without such a forward, another ws, a forward that
// failed): it packs itself.  Round-5 advisor: the flag used to be trusted blindly and the event was one per device,
// shared by every model and stream.  Several forwards before their backwards (GraphGeneratorRL.py:131-132) each have
// their entry; the oldest entry is recycled (its backward then simply packs again).
struct PrepackEntry { const float* ws; bool x2; hipEvent_t ev; int dev; bool valid; unsigned long long age; };
static PrepackEntry g_prepack[32];
static unsigned long long g_prepack_clock = 0;
static std::mutex g_prepack_mu;
hipEvent_t prepack_stamp(const float* ws, bool x2) {
    std::lock_guard<std::mutex> lock(g_prepack_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    PrepackEntry* slot = nullptr;
    for (PrepackEntry& e : g_prepack)
        if (e.valid && e.ws == ws && e.dev == dev) slot = &e;
    if (!slot) {
        slot = &g_prepack[0];
        for (PrepackEntry& e : g_prepack) {
            if (!e.valid) { slot = &e; break; }
            if (e.age < slot->age) slot = &e;
        }
    }
    if (slot->ev && slot->dev != dev) { (void)hipEventDestroy(slot->ev); slot->ev = nullptr; }
    if (!slot->ev && hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming) != hipSuccess) { slot->valid = false; return nullptr; }
    slot->ws = ws; slot->x2 = x2; slot->dev = dev; slot->valid = true; slot->age = ++g_prepack_clock;
    return slot->ev;
}
void prepack_forget(const float* ws) {          // a forward that packs nothing ahead into this workspace
    std::lock_guard<std::mutex> lock(g_prepack_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (PrepackEntry& e : g_prepack)
        if (e.valid && e.ws == ws && e.dev == dev) e.valid = false;
}
hipEvent_t prepack_lookup(const float* ws, bool x2) {     // (both calls of a two-phase backward find it)
    std::lock_guard<std::mutex> lock(g_prepack_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (PrepackEntry& e : g_prepack)
        if (e.valid && e.ws == ws && e.dev == dev && e.x2 == x2) return e.ev;
    return nullptr;
}

static bool sizes_ok(int S, int E, int U, int D0) {
    return S >= 0 && E >= 0 && U >= 0 && U <= E && D0 >= 0 && D0 <= U;
}

extern "C" long long gi_ggnn_workspace_floats(const gi_ggnn_dims* d, int S, int E, int U, int D0) {
    Model m;
    if (build_model(d, m) || !sizes_ok(S, E, U, D0)) return GI_EINVAL;
    Ws w;
    make_ws(m, S, E, U, D0, w);
    return w.total;
}

extern "C" long long gi_ggnn_hx0_offset(const gi_ggnn_dims* d, int S, int E, int U, int D0) {
    Model m;
    if (build_model(d, m) || !sizes_ok(S, E, U, D0)) return GI_EINVAL;
    Ws w;
    make_ws(m, S, E, U, D0, w);
    return w.hx[0];
}

extern "C" int gi_ggnn_ldhx(const gi_ggnn_dims* d) {
    Model m;
    if (build_model(d, m)) return GI_EINVAL;
    return gi_r4(d->H + d->Fn);
}

extern "C" long long gi_ggnn_slab_floats(const gi_ggnn_dims* d, int S, int U, const int* Ut) {
    Model m;
    const int E = U;
    const int* Et = Ut;
    if (build_model(d, m) || S < 0 || E < 0) return GI_EINVAL;
    static_assert(sizeof(SlabPlan) < (1 << 16), "plan size");
    SlabPlan sp;
    if (m.nparams > 160) return GI_ELIMIT;
    plan_slabs(m, S, E, Et, sp);
    return sp.total;
}

// Debug/test hook: offset (floats) and leading dimension of a named workspace buffer.
extern "C" int gi_ggnn_ws_query(const gi_ggnn_dims* d, int S, int E, int U, int D0, const char* name,
                                int i, int j, long long* off, int* ld) {
    Model m;
    if (build_model(d, m) || !sizes_ok(S, E, U, D0) || !name || !off || !ld) return GI_EINVAL;
    if (i < 0 || i > MAXP || j < 0 || j >= MAXL) return GI_EINVAL;
    Ws w;
    make_ws(m, S, E, U, D0, w);
    struct Item { const char* n; long long o; int l; };
    const Item items[] = {
        {"hx", w.hx[i], w.ldhx}, {"eact", w.eact[i < MAXP ? i : 0][j], w.ldEh},
        {"m", w.m[i < MAXP ? i : 0], w.ldM}, {"agg", w.agg[i < MAXP ? i : 0], w.ldM},
        {"gi", w.gi[i < MAXP ? i : 0], w.ld3H}, {"gh", w.gh[i < MAXP ? i : 0], w.ld3H},
        {"att_act", w.att_act[j], w.ldAtt}, {"en", w.en, w.ldG},
        {"emb_act", w.emb_act[j], w.ldEmb}, {"emb", w.embo, w.ldG},
        {"add1_act", w.add1_act[j], w.ldM1}, {"add1", w.add1o, w.ldA},
        {"conn1_act", w.conn1_act[j], w.ldM1}, {"conn1", w.conn1o, w.ldC},
        {"cat_add", w.cat_add, w.ldCA}, {"cat_conn", w.cat_conn, w.ldCC}, {"gemb", w.gemb, w.ldG},
        {"add2_act", w.add2_act[j], w.ldM2}, {"conn2_act", w.conn2_act[j], w.ldM2},
        {"term2_act", w.term2_act[j], w.ldM2}, {"dcat_add", w.dcat_add, w.ldCA},
        {"dcat_conn", w.dcat_conn, w.ldCC}, {"dgemb", w.dgemb, w.ldG}, {"dh", w.dh, w.ldH},
        {"dh2", w.dh2, w.ldH}, {"dxe", w.dxe, w.ldH},
        {"aact", w.aact[i < MAXP ? i : 0][j], w.ldEa}, {"een", w.een[i < MAXP ? i : 0], w.ldM},
    };
    for (const Item& it : items)
        if (!strcmp(it.n, name)) { *off = it.o; *ld = it.l; return 0; }
    return GI_EINVAL;
}

// AlphaDropout training mode needs the graph without row sharing (gi_compact_count_ex, nodedup): an
// independent mask per padded slot and per edge
static int dropout_graph_ok(const gi_ggnn_dims& d, int S, int E, int U, int D0) {
    if (!d.dropout) return 0;
    if (S != d.B * d.N || U != E || D0 != 0) return GI_EINVAL;
    return 0;
}

extern "C" int gi_ggnn_forward(const gi_ggnn_dims* dp, const float* const* params,
                               const gi_graph* gp, float* ws, float* out, int ldout, void* stream) {
    return gi_ggnn_forward_ex(dp, params, gp, ws, out, ldout, stream, nullptr, 0);
}

extern "C" int gi_ggnn_forward_ex(const gi_ggnn_dims* dp, const float* const* params,
                                  const gi_graph* gp, float* ws, float* out, int ldout, void* stream,
                                  void* side_stream, int run_flags) {
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (run_flags & ~(GI_RUN_PREPACK_BWD | GI_RUN_NO_X2)) return GI_EINVAL;
    Model m;
    int rc = build_model(dp, m);
    if (rc) return rc;
    if (!gp) return GI_EINVAL;
    const int S = gp->S, E = gp->E, U = gp->U;
    const int* gfix = gp->gfix;
    const int* u_src = gp->u_src;
    const int* in_perm = gp->in_perm;
    const int* Ut = gp->Ut;
    if (!params || !gfix || !ws || !out || S < 0 || E < 0 || U < 0 || U > E || !Ut) return GI_EINVAL;
    if (E > 0 && (!u_src || !in_perm || U == 0)) return GI_EINVAL;
    const gi_ggnn_dims& d = m.d;
    if (ldout < m.NA + m.NC + 1) return GI_EINVAL;
    if (d.kind == GI_KIND_ATTGGNN && E == 0) in_perm = in_perm ? in_perm : gfix;   // never read
    gi_compact_layout_t L;
    rc = gi_compact_layout(d.B, d.N, d.Fe, &L);
    if (rc) return rc;
    Ws w;
    if (gp->D0 < 0 || gp->D0 > U || (gp->D0 > 0 && (!gp->d_src || !gp->cmat || gp->ldc0 < gp->D0)))
        return GI_EINVAL;
    if (d.kind == GI_KIND_ATTGGNN && gp->D0 > 0 && (!gp->e2d || !gp->cls_off || !gp->cls_edges))
        return GI_EINVAL;
    if (int drc = dropout_graph_ok(d, S, E, U, gp->D0)) return drc;
    if (gp->bounded && (d.dropout || gp->D0 <= 0)) return GI_EINVAL;   // (bounded: S, E, U, D0 are bounds)
    make_ws(m, S, E, U, gp->D0, w);
    Run r{(hipStream_t)stream, params, 0};
    r.drop = d.dropout != 0; r.seed = d.drop_seed; r.fshift = w.fshift;
    r.skinny = ws + w.skinny; r.skinny_floats = w.skinny_floats;
    r.x2 = x2_enabled() && !(run_flags & GI_RUN_NO_X2);
    r.guard = gp->x2_guard; r.guard_host = gp->x2_guard_host;
    const int R = w.R;
    if (gp->bounded) { r.dims = gfix + L.dims; r.d0_dev = gfix + L.counts + 20; r.R_bound = R; }
    int maxUt = 0;
    for (int t = 0; t < d.Fe; ++t) maxUt = std::max(maxUt, Ut[t]);
    const Grp bytype{d.Fe, gfix + L.type_off, maxUt, Ut};
    Grp bytype0{d.Fe, gfix + L.type_off0, w.D0};   // pass-0 rows (upper bound per type: all)
    bytype0.dim_slot = 2;
    const int* seg_off = gfix + L.seg_off;
    const int* cidx = gfix + L.cidx;
    const int* mask = gfix + L.node_mask;
    const bool attn = d.kind == GI_KIND_ATTGGNN;

    r.eatt0 = m.eatt;
    // gi_graph.wcache: the weights-only data of this forward live in (and, when valid, come from) the caller's cache
    Wc wc;
    wcache_layout(m, wc);
    float* const wcache = (gp->wcache && !r.drop && r.x2 && bf3_enabled()) ? gp->wcache : nullptr;
    if (wcache && ((uintptr_t)wcache & 15)) return GI_EINVAL;
    r.wc_valid = wcache && gp->wcache_valid != 0;
    r.wc_bf3 = wcache ? wcache + wc.bf3_wamax : nullptr;
    if (d.passes > 0 && E > 0)          // packed weight images of the chain kernel, once per forward
        for (int k = 0; k < (attn ? 2 : 1); ++k)
            if (w.img_f_n[k] > 0) {
                r.img_f[k] = ws + w.img_f[k];
                if (!r.drop && chain_fwd_x2_enabled(r.x2)) {   // the row-independent fp16x2 chain: its image instead of the fp32 one
                    float* const img = (wcache && wc.img_fx[k] >= 0) ? wcache + wc.img_fx[k] : ws + w.img_fx[k];
                    float* const cells = (wcache && wc.img_fx[k] >= 0) ? wcache + wc.chain_amax_f[k] : ws + w.chain_amax_f[k];
                    if (!(r.wc_valid && wc.img_fx[k] >= 0)) {
                    Run rp = r;
                    rp.chain_amax[k] = cells;
                    chain_pack(rp, k ? m.eatt : m.msg, d.Fe, false, img);
                    r.chk(rp.rc);
                    }
                    r.img_fx[k] = img;
                    r.chain_amax_f[k] = cells;
                    r.img_f[k] = r.img_fx[k];                   // (nothing may read the fp32 image: it was not packed)
                } else {
                    chain_pack(r, k ? m.eatt : m.msg, d.Fe, false, r.img_f[k]);
                }
            }
    // Everything else that depends on the weights only goes to the side stream when there is one: the amax cells of
    // the fp16x2 layers (needed by the readout, hundreds of microseconds from here) and — GI_RUN_PREPACK_BWD — the
    // images the backward would otherwise pack at its start, in front of its first launches.
    SideStream fside{(hipStream_t)side_stream, 0};
    hipStream_t prep = r.st;
    hipEvent_t cells_ready = nullptr;
    if (side_stream) {
        hipEvent_t start = fside.next();
        r.chk((int)hipEventRecord(start, r.st));            // (the weights were last written on the caller's stream)
        r.chk((int)hipStreamWaitEvent(fside.st, start, 0));
        prep = fside.st;
    }
    // amax cells of the message / energy stacks' activations (this forward's chains publish into them, the backward's
    // chains add the dZ maxima, its fp16x2 weight gradients read both): zeroed first thing on the side stream
    const bool msgx = E > 0 && d.passes > 0 && msg_wgrad_x2_possible(m, r.x2);
    float* msg_cells_base[2] = {nullptr, nullptr};
    hipEvent_t msg_cells_ready = nullptr;
    if (msgx) {
        for (int k = 0; k < (attn ? 2 : 1); ++k)
            if (w.img_f_n[k] > 0 && r.img_fx[k]) {
                msg_cells_base[k] = ws + w.msg_amax[k];
                r.chk((int)hipMemsetAsync(msg_cells_base[k], 0, sizeof(float) * MSG_CELLS_PER_PASS * d.passes, prep));
            }
        if (side_stream) {
            msg_cells_ready = fside.next();
            r.chk((int)hipEventRecord(msg_cells_ready, fside.st));
        }
    }
    if ((run_flags & GI_RUN_PREPACK_BWD) && !r.drop) {
        bf3_prepare(r, m, ws, w, true, R, BF3_DO_PACK, prep);
        if (d.passes > 0 && E > 0)
            for (int k = 0; k < (attn ? 2 : 1); ++k)
                if (w.img_b_n[k] > 0) {
                    Run rp = r;                                   // (chain_pack reads the stream and the flavour from its Run)
                    rp.st = prep;
                    rp.chain_amax[k] = chain_x2_enabled(r.x2) ? ws + w.chain_amax[k] : nullptr;
                    chain_pack(rp, k ? m.eatt : m.msg, d.Fe, true, ws + w.img_b[k]);
                    r.chk(rp.rc);
                }
    }
    // the forward chains' weights through the fp16x2 guard (output channels / input columns below the per-tensor range)
    if (r.guard && r.ok() && !r.wc_valid)
        for (int k = 0; k < (attn ? 2 : 1); ++k)
            if (r.img_fx[k]) {
                const Mlp* mlps = k ? m.eatt : m.msg;
                gi_absmax_desc wd[GI_ABSMAX_MAX];
                int n = 0;
                const int L = mlps[0].layers();
                for (int l = 0; l < L; ++l)
                    for (int t = 0; t < d.Fe; ++t) {
                        wd[n].x = r.P[mlps[t].w(l)]; wd[n].rows = mlps[0].fan_out(l); wd[n].cols = mlps[0].fan_in(l);
                        wd[n].ld = wd[n].cols;
                        wd[n].out = r.chain_amax_f[k] + ((long long)l * d.Fe + t) * GI_AMAX_WORDS;
                        if (++n == GI_ABSMAX_MAX || (l == L - 1 && t == d.Fe - 1)) {
                            r.chk(gi_x2_weight_guard(wd, n, r.guard + 1, r.guard_host, prep));
                            n = 0;
                        }
                    }
            }
    bf3_prepare(r, m, ws, w, false, R, BF3_DO_AMAX, prep);
    hipEvent_t packed = nullptr;                  // this workspace's "images written" event (prepack_stamp)
    if ((run_flags & GI_RUN_PREPACK_BWD) && !r.drop) packed = prepack_stamp(ws, r.x2);
    else prepack_forget(ws);
    if (side_stream) {
        cells_ready = fside.next();
        r.chk((int)hipEventRecord(cells_ready, fside.st));
        if (packed) r.chk((int)hipEventRecord(packed, fside.st));
    } else if (packed) {
        r.chk((int)hipEventRecord(packed, r.st));
    }
    // pass-0 row cache (inference loops): only in front of the one-launch stack path, whose kernel can skip
    int* const p0c = static_cast<int*>(gp->p0_cache);
    const bool p0cache = p0c && w.D0 > 0 && E > 0 && !r.drop && r.img_f[0] && (!attn || r.img_f[1]) &&
                         w.ldhx >= gi_r4(m.msg[0].in);
    // ---- message passes (gnn/summation_mpnn.py:128-144) ----------------------------------------
    for (int p = 0; p < d.passes; ++p) {
        const float* hx = ws + w.hx[p];
        r.pass = p;
        for (int k = 0; k < 2; ++k)     // (pass 0 on the class rows: one-slab fp32 weight gradients, no cells)
            r.msg_cells[k] = (msg_cells_base[k] && !(p == 0 && w.D0 > 0)) ? msg_cells_base[k] + p * MSG_CELLS_PER_PASS : nullptr;
        if (msg_cells_ready && (r.msg_cells[0] || r.msg_cells[1])) {
            r.chk((int)hipStreamWaitEvent(r.st, msg_cells_ready, 0));    // (zeroed ~100 us ago: never stalls)
            msg_cells_ready = nullptr;
        }
        if (p == 0 && p0cache) {
            r.chk(gi_p0_cache_lookup(gfix, d.B, d.N, d.Fe, p0c, attn ? 2 : 1, ws + w.m[0],
                                     attn ? ws + w.een[0] : nullptr, w.ldM, r.st));
            r.skip = p0c;
        }
        if (attn) {
            // AttentionGGNN.aggregate_message (gnn/mpnn.py:370-389): message and energy MLPs of the
            // edge's bond type on h_src(e), softmax over each node's incoming edges, weighted sum
            // pass 0 (h = [x | 0]): both MLP families on the D0 (feature class, bond type) rows, the
            // softmax reads them through the edge -> pass-0 row index
            const bool p0 = p == 0 && w.D0 > 0;
            if (E > 0) {
                EdgeChain ch[2] = {
                    {m.msg, w.eact[p], w.edz[p], w.ldEh, ws + w.m[p], w.ldM, nullptr},
                    {m.eatt, w.aact[p], w.adz[p], w.ldEa, ws + w.een[p], w.ldM, nullptr}};
                if (p0) edge_chains_forward(r, ws, ch, 2, bytype0, hx, w.ldhx, gp->d_src, w.D0);
                else edge_chains_forward(r, ws, ch, 2, bytype, hx, w.ldhx, u_src, U);
                if (p0 && p0cache) {
                    r.skip = nullptr;
                    r.chk(gi_p0_cache_insert(gfix, d.B, d.N, d.Fe, p0c, 2, ws + w.m[0], ws + w.een[0], w.ldM, r.st));
                }
            }
            r.chk(gi_seg_softmax_fwd_n(ws + w.een[p], ws + w.m[p], w.ldM, p0 ? gp->e2d : in_perm, seg_off,
                                       R, d.M, ws + w.agg[p], w.ldM, r.dims, r.st));
        } else if (p == 0 && w.D0 > 0) {
            // pass 0: h = [x | 0], one message row per (feature class, bond type); a_v = cmat . m0
            mlp_forward(r, ws, m.msg, bytype0, hx, w.ldhx, gp->d_src, w.D0, w.eact[0], w.ldEh,
                        ws + w.m[0], w.ldM);
            if (p0cache) {
                r.skip = nullptr;
                r.chk(gi_p0_cache_insert(gfix, d.B, d.N, d.Fe, p0c, 1, ws + w.m[0], nullptr, w.ldM, r.st));
            }
            if (r.ok()) {
                gi_gemm_params q;
                gemm_defaults(q);
                q.A = gp->cmat; q.lda = gp->ldc0; q.B = ws + w.m[0]; q.ldb = w.ldM; q.b_major = 1;
                q.C = ws + w.agg[0]; q.ldc = w.ldM; q.M = R; q.N = d.M; q.K = w.D0;
                q.m_dev = r.dims; q.k_dev = r.d0_dev;      // (bounded forward: R and D0 live on the device)
                q.tm = 1; q.tn = 1;
                r.chk(gi_gemm(&q, r.st));
            }
        } else {
            if (E > 0)   // m_u = MLP_type(u)(h_src(u)), gnn/mpnn.py:284-294, once per message row
                mlp_forward(r, ws, m.msg, bytype, hx, w.ldhx, u_src, U, w.eact[p], w.ldEh,
                            ws + w.m[p], w.ldM);
            // a_v = sum of incoming messages (:141)
            r.chk(gi_seg_sum_n(ws + w.m[p], w.ldM, in_perm, seg_off, R, d.M, ws + w.agg[p], w.ldM, 0, r.dims,
                               r.st));
        }
        // GRU update (gnn/mpnn.py:296-297): projections + gates in ONE launch (gi_gru.hip, round 6) ...
        if (gi_gru_fused_ok(d.H, d.M, w.ldM, w.ldhx, w.ld3H)) {
            r.chk(gi_gru_fused_fwd(ws + w.agg[p], w.ldM, hx, w.ldhx, params[m.gru_wih], params[m.gru_whh],
                                   params[m.gru_bih], params[m.gru_bhh], ws + w.gi[p], ws + w.gh[p], w.ld3H,
                                   ws + w.hx[p + 1], seg_off, R, r.dims, d.H, d.M, r.st));
            continue;
        }
        // ... or (GI_GRU_FUSED=0, widths that are not multiples of 4): both input projections in one launch, then the gate kernel
        {
            Batch b;
            add_fwd(b, r, params[m.gru_wih], params[m.gru_bih], d.M, 3 * d.H, ws + w.agg[p], w.ldM, R,
                    ws + w.gi[p], w.ld3H, false);
            add_fwd(b, r, params[m.gru_whh], params[m.gru_bhh], d.H, 3 * d.H, hx, w.ldhx, R,
                    ws + w.gh[p], w.ld3H, false);
            flush_batch(r, b, false);
        }
        r.chk(gi_gru_gates_fwd_n(ws + w.gi[p], ws + w.gh[p], w.ld3H, hx, ws + w.hx[p + 1], w.ldhx,
                                 seg_off, R, d.H, d.Fn, r.dims, r.st));
    }
    // ---- readout (gnn/mpnn.py:299-303) -----------------------------------------------------------
    if (cells_ready) r.chk((int)hipStreamWaitEvent(r.st, cells_ready, 0));     // amax cells zeroed, max |W| in them
    const float* hx = ws + w.hx[d.passes];
    {   // the four node-level stacks, layer by layer in shared launches
        MlpJob jobs[4] = {};
        // widest stacks first: their workgroups run twice as long (K = 500 against 250), so the
        // launch's last, partially filled round of workgroups consists of the short ones
        jobs[0] = {&m.add1, hx, w.ldhx, R, w.add1_act, w.ldM1, ws + w.add1o, w.ldA};
        jobs[1] = {&m.conn1, hx, w.ldhx, R, w.conn1_act, w.ldM1, ws + w.conn1o, w.ldC};
        jobs[2] = {&m.att, hx, w.ldhx, R, w.att_act, w.ldAtt, ws + w.en, w.ldG};
        jobs[3] = {&m.emb, hx, w.ldhx, R, w.emb_act, w.ldEmb, ws + w.embo, w.ldG};
        mlp_jobs_forward(r, ws, jobs, 4);
    }
    r.chk(gi_gather_readout_fwd(ws + w.en, ws + w.embo, w.ldG, cidx, mask, d.B, d.N, d.G,
                                d.big_positive, ws + w.cat_add + m.NA, w.ldCA,
                                ws + w.cat_conn + m.NC, w.ldCC, ws + w.gemb, w.ldG, r.st));
    if (gi_fuse_flags() & GI_FUSE_SLOTS) {
        r.chk(gi_expand_slots2(ws + w.add1o, w.ldA, d.A, ws + w.cat_add, w.ldCA, ws + w.conn1o, w.ldC, d.C,
                               ws + w.cat_conn, w.ldCC, cidx, d.B, d.N, r.st));
    } else {
        r.chk(gi_expand_slots(ws + w.add1o, w.ldA, cidx, d.B, d.N, d.A, ws + w.cat_add, w.ldCA, r.st));
        r.chk(gi_expand_slots(ws + w.conn1o, w.ldC, cidx, d.B, d.N, d.C, ws + w.cat_conn, w.ldCC, r.st));
    }
    {   // the three graph-level stacks write straight into the logits
        MlpJob jobs[3] = {};
        jobs[0] = {&m.add2, ws + w.cat_add, w.ldCA, d.B, w.add2_act, w.ldM2, out, ldout};
        jobs[1] = {&m.conn2, ws + w.cat_conn, w.ldCC, d.B, w.conn2_act, w.ldM2, out + m.NA, ldout};
        jobs[2] = {&m.term2, ws + w.gemb, w.ldG, d.B, w.term2_act, w.ldM2, out + m.NA + m.NC, ldout};
        // dropout mode: `out` is [2 B, ldout]; rows [B, 2 B) take the factors of the logits
        for (MlpJob& j : jobs) j.out_fshift = r.drop ? (long long)d.B * ldout : 0;
        mlp_jobs_forward(r, ws, jobs, 3);
    }
    // Whatever the caller enqueues on `stream` after this call — the backward, or a kernel that reuses the workspace's
    // memory because the tape was dropped without one — is ordered behind the side stream's packs into `ws` (they
    // finished hundreds of microseconds ago: a wait that never stalls).  The caller therefore needs no cross-stream
    // bookkeeping of its own for `ws` (torch: no Tensor.record_stream, which would keep the caching allocator from
    // reusing the block until the low-priority side stream has drained).
    if (side_stream && packed) r.chk((int)hipStreamWaitEvent(r.st, packed, 0));
    if (!r.ok()) prepack_forget(ws);              // (a failed forward's images are not to be trusted)
    return r.rc;
}

// fp16x2 instead of bf16x3 on the GI_GEMM_BF3 launches of gi_ggnn_forward / backward (environment GI_X2, default 1):
// on = 1 / 0 sets, on < 0 queries; returns the previous setting.
extern "C" int gi_x2_enable(int on) {
    const int prev = x2_enabled() ? 1 : 0;
    if (on >= 0) g_x2 = on ? 1 : 0;
    return prev;
}

extern "C" int gi_bf3_enable(int on) {
    const int was = bf3_enabled() ? 1 : 0;
    if (on >= 0) g_bf3 = on ? 1 : 0;
    return was;
}

extern "C" long long gi_p0_cache_words(const gi_ggnn_dims* d) {
    Model m;
    if (int rc = build_model(d, m)) return rc;
    Ws w;
    make_ws(m, 0, 0, 0, 0, w);
    return gi_p0_cache_words_for((m.d.kind == GI_KIND_ATTGGNN ? 2 : 1) * w.ldM);
}

extern "C" long long gi_ggnn_wcache_floats(const gi_ggnn_dims* d) {
    Model m;
    if (int rc = build_model(d, m)) return rc;
    Wc c;
    wcache_layout(m, c);
    return c.total;
}

extern "C" int gi_ggnn_first_readout_param(const gi_ggnn_dims* d) {
    Model m;
    const int rc = build_model(d, m);
    return rc ? rc : m.att.base;
}

extern "C" int gi_ggnn_backward(const gi_ggnn_dims* dp, const float* const* params,
                                const gi_graph* gp, float* ws, float* slabs, const float* y_out,
                                int ldout, const float* d_out, int lddout, float* const* grads,
                                void* stream, void* side_stream) {
    return gi_ggnn_backward_phase(dp, params, gp, ws, slabs, y_out, ldout, d_out, lddout, grads,
                                  stream, side_stream, GI_BWD_ALL);
}

extern "C" int gi_ggnn_backward_phase(const gi_ggnn_dims* dp, const float* const* params,
                                      const gi_graph* gp, float* ws, float* slabs,
                                      const float* y_out, int ldout, const float* d_out, int lddout,
                                      float* const* grads, void* stream, void* side_stream, int phase) {
    bool prepacked = (phase & GI_BWD_PREPACKED) != 0;
    const bool no_x2 = (phase & GI_BWD_NO_X2) != 0;
    phase &= ~(GI_BWD_PREPACKED | GI_BWD_NO_X2);
    if (phase != GI_BWD_ALL && phase != GI_BWD_READOUT && phase != GI_BWD_PASSES) return GI_EINVAL;
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    Model m;
    int rc = build_model(dp, m);
    if (rc) return rc;
    if (!gp) return GI_EINVAL;
    const int S = gp->S, E = gp->E, U = gp->U;
    const int* gfix = gp->gfix;
    const int* u_src = gp->u_src;
    const int* in_perm = gp->in_perm;
    const int* mu_off = gp->mu_off;
    const int* mu_dst = gp->mu_dst;
    const int* mu_slot = gp->mu_slot;
    const int* out_perm = gp->out_perm;
    const int* Ut = gp->Ut;
    if (!params || !gfix || !ws || !slabs || !y_out || !d_out || !grads || S < 0 || E < 0 || U < 0 ||
        U > E || !Ut)
        return GI_EINVAL;
    if (E > 0 && (!u_src || !in_perm || !mu_off || !mu_dst || !mu_slot || !out_perm || U == 0))
        return GI_EINVAL;
    if (m.nparams > 160) return GI_ELIMIT;
    if (gp->wcache) return GI_EINVAL;     // (a forward that used the weights cache left no max |W| cells in ws)
    const gi_ggnn_dims& d = m.d;
    gi_compact_layout_t L;
    rc = gi_compact_layout(d.B, d.N, d.Fe, &L);
    if (rc) return rc;
    Ws w;
    if (gp->D0 < 0 || gp->D0 > U || (gp->D0 > 0 && (!gp->d_src || !gp->cmat || gp->ldc0 < gp->D0)))
        return GI_EINVAL;
    if (d.kind == GI_KIND_ATTGGNN && gp->D0 > 0 && (!gp->e2d || !gp->cls_off || !gp->cls_edges))
        return GI_EINVAL;
    if (int drc = dropout_graph_ok(d, S, E, U, gp->D0)) return drc;
    make_ws(m, S, E, U, gp->D0, w);
    SlabPlan sp;
    plan_slabs(m, S, U, Ut, sp);
    Grp bytype0{d.Fe, gfix + L.type_off0, w.D0};
    bytype0.dim_slot = 2;                                  // (marks the pass-0 rows: p0_layerwise)
    Run r{(hipStream_t)stream, params, 0};
    r.drop = d.dropout != 0; r.seed = d.drop_seed; r.fshift = w.fshift;
    r.skinny = ws + w.skinny; r.skinny_floats = w.skinny_floats;
    r.x2 = x2_enabled() && !no_x2;
    r.guard = gp->x2_guard; r.guard_host = nullptr;       // (the backward only counts: dZ rows never trip)
    if (prepacked) {                                      // images written by THIS workspace's forward (side stream)?
        if (hipEvent_t ev = prepack_lookup(ws, r.x2)) r.chk((int)hipStreamWaitEvent(r.st, ev, 0));
        else prepacked = false;                           // no such forward on record: pack here
    }
    const long long out_fshift = r.drop ? (long long)d.B * ldout : 0;   // logits -> their factors
    const int R = w.R;
    int maxUt = 0;
    for (int t = 0; t < d.Fe; ++t) maxUt = std::max(maxUt, Ut[t]);
    const Grp bytype{d.Fe, gfix + L.type_off, maxUt, Ut};
    const int* seg_off = gfix + L.seg_off;
    const int* src_off = gfix + L.src_off;
    const int* cidx = gfix + L.cidx;
    const int* mask = gfix + L.node_mask;
    const int NA = m.NA, NC = m.NC;
    const bool attn = d.kind == GI_KIND_ATTGGNN;

    Deferred dq;
    SideStream side_obj{(hipStream_t)side_stream, 0};
    SideStream* const passes_side = side_stream ? &side_obj : nullptr;
    r.side = passes_side;
    r.sp = &sp; r.slabs = slabs; r.grads = grads;
    r.wgrad_x2 = wgrad_x2_pool_possible(m, r.x2);
    r.pool.base = r.wgrad_x2 ? ws + w.amax_pool : nullptr;
    const Grp none{0, nullptr, 0};
    const float* hxP = ws + w.hx[d.passes];
    float* dh = ws + w.dh;
    float* dh2 = ws + w.dh2;
    float* dhb = ws + w.dhb;
    float* dhc = ws + w.dhc;
    float* dhd = ws + w.dhd;
    auto readout_params = [&](auto&& fn) {
        const Mlp* stacks[] = {&m.att, &m.emb, &m.add1, &m.conn1, &m.add2, &m.conn2, &m.term2};
        for (const Mlp* q : stacks)
            for (int l = 0; l < q->layers(); ++l) fn(q->w(l));
    };
    if (phase == GI_BWD_PASSES)          // the readout half ran (and was reduced) in an earlier call
        readout_params([&](int widx) { sp.e[widx].reduced = 1; });
    if (phase != GI_BWD_PASSES) {
    bf3_prepare(r, m, ws, w, true, S + 1, prepacked ? 0 : BF3_DO_PACK, r.st);
    // ---- tier 2 (gnn/modules.py:265-279) ---------------------------------------------------------
    if (gi_fuse_flags() & GI_FUSE_TIER2_DSELU) {
        r.chk(gi_selu_bwd_cols3_f(d_out, lddout, y_out, ldout, out_fshift, d.B, NA, ws + w.dzA, w.ldNA, NC,
                                  ws + w.dzC, w.ldNC, 1, ws + w.dzT, 4, r.st));
    } else {
        r.chk(gi_selu_bwd_rows_f(d_out, lddout, nullptr, y_out, ldout, ws + w.dzA, w.ldNA, d.B, NA,
                                 out_fshift, r.st));
        r.chk(gi_selu_bwd_rows_f(d_out + NA, lddout, nullptr, y_out + NA, ldout, ws + w.dzC, w.ldNC, d.B,
                                 NC, out_fshift, r.st));
        r.chk(gi_selu_bwd_rows_f(d_out + NA + NC, lddout, nullptr, y_out + NA + NC, ldout, ws + w.dzT, 4,
                                 d.B, 1, out_fshift, r.st));
    }
    {
        MlpJob jobs[3] = {};
        jobs[0] = {&m.add2, ws + w.cat_add, w.ldCA, d.B, w.add2_act, w.ldM2, nullptr, 0, w.add2_dz,
                   ws + w.dzA, w.ldNA, ws + w.dcat_add, w.ldCA, NA + d.G, false};
        jobs[1] = {&m.conn2, ws + w.cat_conn, w.ldCC, d.B, w.conn2_act, w.ldM2, nullptr, 0,
                   w.conn2_dz, ws + w.dzC, w.ldNC, ws + w.dcat_conn, w.ldCC, NC + d.G, false};
        jobs[2] = {&m.term2, ws + w.gemb, w.ldG, d.B, w.term2_act, w.ldM2, nullptr, 0, w.term2_dz,
                   ws + w.dzT, 4, ws + w.dgemb, w.ldG, d.G, false};
        mlp_jobs_backward(r, ws, sp, slabs, dq, jobs, 3);
    }
    // ---- gather + tier-1 glue: dZ of the last att/emb/add1/conn1 layers, in place ----------------
    r.chk(gi_gather_readout_bwd_f(ws + w.en, ws + w.embo, w.ldG, cidx, mask, d.B, d.N, d.G, S,
                                  d.big_positive, ws + w.dgemb, w.ldG, ws + w.dcat_add + NA, w.ldCA,
                                  ws + w.dcat_conn + NC, w.ldCC, ws + w.zpart_g, r.fshift, r.st));
    if (gi_fuse_flags() & GI_FUSE_SLOTS) {
        r.chk(gi_compress_slots2_f(ws + w.add1o, w.ldA, d.A, ws + w.dcat_add, w.ldCA, ws + w.zpart_a, w.ldA,
                                   ws + w.conn1o, w.ldC, d.C, ws + w.dcat_conn, w.ldCC, ws + w.zpart_c,
                                   w.ldC, cidx, d.B, d.N, S, r.fshift, r.st));
    } else {
        r.chk(gi_compress_slots_f(ws + w.add1o, w.ldA, cidx, d.B, d.N, d.A, S, ws + w.dcat_add, w.ldCA,
                                  ws + w.zpart_a, w.ldA, r.fshift, r.st));
        r.chk(gi_compress_slots_f(ws + w.conn1o, w.ldC, cidx, d.B, d.N, d.C, S, ws + w.dcat_conn, w.ldCC,
                                  ws + w.zpart_c, w.ldC, r.fshift, r.st));
    }
    {   // zero-row gradients of the four stacks: per-graph partial sums -> row S, one launch
        float* en_z = ws + w.en + (long long)S * w.ldG;
        float* emb_z = ws + w.embo + (long long)S * w.ldG;
        float* add_z = ws + w.add1o + (long long)S * w.ldA;
        float* conn_z = ws + w.conn1o + (long long)S * w.ldC;
        const gi_colsum_desc cs[4] = {
            {ws + w.zpart_g, w.ldZG, d.B, d.G, en_z, en_z},
            {ws + w.zpart_g + d.G, w.ldZG, d.B, d.G, emb_z, emb_z},
            {ws + w.zpart_a, w.ldA, d.B, d.A, add_z, add_z},
            {ws + w.zpart_c, w.ldC, d.B, d.C, conn_z, conn_z}};
        r.chk(gi_colsum_multi(cs, 4, r.st));
    }
    // ---- node-level readout MLPs -> dh -------------------------------------------------------------
    {   // every sibling writes its own d h; the GRU-gate backward of the last pass sums the four
        MlpJob jobs[4] = {};
        jobs[0] = {&m.add1, hxP, w.ldhx, R, w.add1_act, w.ldM1, nullptr, 0, w.add1_dz, ws + w.add1o,
                   w.ldA, dh, w.ldH, d.H, false};
        jobs[1] = {&m.conn1, hxP, w.ldhx, R, w.conn1_act, w.ldM1, nullptr, 0, w.conn1_dz,
                   ws + w.conn1o, w.ldC, dhb, w.ldH, d.H, false};
        jobs[2] = {&m.emb, hxP, w.ldhx, R, w.emb_act, w.ldEmb, nullptr, 0, w.emb_dz, ws + w.embo,
                   w.ldG, dhc, w.ldH, d.H, false};
        jobs[3] = {&m.att, hxP, w.ldhx, R, w.att_act, w.ldAtt, nullptr, 0, w.att_dz, ws + w.en, w.ldG,
                   dhd, w.ldH, d.H, false};
        // No weight-gradient batches beside the node-level dgrad launches: those fill the device by
        // themselves (2 760 workgroups each at the headline batch), two streams only contend there;
        // everything queued goes out behind them, under the message passes' short launches (round 2: step
        // 2.362 -> 2.344 ms, GEMM-family per-launch figure 0.349 -> 0.374 of peak; re-measured in round 3 with the
        // dgrad launches on the bf16 pipe: 2.20 ms held against 2.25-2.28 released).
        // Round 5, with the backward bound by the weight-gradient queue: holding still wins at the headline batch
        // (7.3 k node rows: 1.935 against 1.944 ms) and loses from ~10 k rows on (B = 2000: 3.35 -> 3.24 ms, ZINC shape
        // 4.05 -> 4.00, ChEMBL shape 3.24 -> 3.21, B = 4000 a tie; profiles/r05/ab/ab_hold_kicks.txt).
        // GI_HOLD_KICKS = 0 / 1 forces either.
        static const int hold_env = getenv("GI_HOLD_KICKS") ? atoi(getenv("GI_HOLD_KICKS")) : -1;
        r.hold_kicks = hold_env >= 0 ? hold_env != 0 : R <= 9000;
        mlp_jobs_backward(r, ws, sp, slabs, dq, jobs, 4);
        r.hold_kicks = false;
        if (r.side) kick_deferred(r, dq, r.side, false);
    }
    }   // phase != GI_BWD_PASSES
    if (phase == GI_BWD_READOUT) {
        // finish the readout parameters now: their weight-gradient GEMMs and slab reductions are
        // queued (side stream if there is one) so that the caller can start exchanging the gradients
        // of these parameters while the message passes are still being differentiated
        if (r.side) {
            kick_deferred(r, dq, r.side, true);          // also reduces every finished parameter ...
            flush_bias(r, dq, r.side->st);               // ... but those whose bias column comes from its own launch: now
            gi_reduce_desc descs[160];
            int nd = 0;
            readout_params([&](int widx) {
                if (sp.e[widx].reduced) return;
                sp.e[widx].reduced = 1;
                descs[nd++] = reduce_desc(sp.e[widx], slabs, grads, widx);
            });
            if (nd && r.ok()) r.chk(gi_reduce_slabs(descs, nd, r.side->st));
        } else {
            flush_deferred(r, dq);
            flush_bias(r, dq, r.st);
            gi_reduce_desc descs[160];
            int nd = 0;
            readout_params([&](int widx) {
                sp.e[widx].reduced = 1;
                descs[nd++] = reduce_desc(sp.e[widx], slabs, grads, widx);
            });
            if (r.ok()) r.chk(gi_reduce_slabs(descs, nd, r.st));
        }
        return r.rc;
    }
    r.eatt0 = m.eatt;
    if (d.passes > 0 && E > 0)          // packed weight images of the dZ chains, once per backward
        for (int k = 0; k < (attn ? 2 : 1); ++k)
            if (w.img_b_n[k] > 0) {
                if (chain_x2_enabled(r.x2)) r.chain_amax[k] = ws + w.chain_amax[k];
                r.img_b[k] = ws + w.img_b[k];
                r.img_b_stride[k] = w.img_b_stride[k];
                if (!prepacked) chain_pack(r, k ? m.eatt : m.msg, d.Fe, true, r.img_b[k]);
            }
    // ---- message passes, reversed -------------------------------------------------------------------
    // d h scatter (segmented sum of the message stacks' input gradients over the source CSR): its own
    // launch(es) behind the pass's dZ chain, or — GI_FUSE_DH_SCATTER, vector gate kernel (H % 4 == 0) —
    // folded into the gate backward of the next iteration, which is the only reader of that sum
    const bool fuse_scatter = (gi_fuse_flags() & GI_FUSE_DH_SCATTER) && (d.H & 3) == 0 && d.H >= 4;
    const float* scat0 = nullptr;
    const float* scat1 = nullptr;
    const bool msgx = E > 0 && msg_wgrad_x2_possible(m, r.x2) && chain_x2_enabled(r.x2);
    for (int p = d.passes - 1; p >= 0; --p) {
        for (int k = 0; k < (attn ? 2 : 1); ++k)   // the cells this pass's forward chains published into (gi_ggnn_forward_ex)
            r.msg_cells[k] = (msgx && w.img_b_n[k] > 0 && !(p == 0 && w.D0 > 0)) ? ws + w.msg_amax[k] + p * MSG_CELLS_PER_PASS : nullptr;
        const float* hx = ws + w.hx[p];
        float* gi = ws + w.gi[p];
        float* gh = ws + w.gh[p];
        float* agg = ws + w.agg[p];
        const bool last = (p == d.passes - 1);
        if (scat0) {    // the later pass's d h scatter rides in this launch (GI_FUSE_DH_SCATTER)
            r.chk(gi_gru_gates_bwd_ex(gi, gh, w.ld3H, hx, w.ldhx, dh, last ? dhb : nullptr,
                                      last ? dhc : nullptr, last ? dhd : nullptr, dh2, w.ldH, seg_off, R,
                                      d.H, scat0, scat1, w.ldH, out_perm, src_off, r.st));
            scat0 = scat1 = nullptr;
        } else {
            r.chk(gi_gru_gates_bwd(gi, gh, w.ld3H, hx, w.ldhx, dh, last ? dhb : nullptr,
                                   last ? dhc : nullptr, last ? dhd : nullptr, dh2, w.ldH, seg_off, R,
                                   d.H, r.st));
        }
        float* dagg = ws + w.dagg[p];
        // Pass 0 of the class-row path: the main queue has ~100 us of short launches left, the weight-gradient queue
        // everything deferred so far.  GI_P0_GRU_MAIN (default 1): hand that over NOW and keep pass 0's own GRU weight
        // gradients for the main queue's last launch (with the pass-0 message stack's), so that both queues end together.
        if (p == 0 && w.D0 > 0 && r.side && p0_gru_on_main() && !r.p0_on_main) {
            kick_deferred(r, dq, r.side, true); flush_bias(r, dq, r.side->st); r.hold_kicks = r.p0_on_main = true;
        }
        {
            const int wih = m.gru_wih, whh = m.gru_whh;
            defer_wgrad(r, dq, sp, slabs, &wih, none, gi, w.ld3H, agg, w.ldM, nullptr, R);
            defer_wgrad(r, dq, sp, slabs, &whh, none, gh, w.ld3H, hx, w.ldhx, nullptr, R);
            // d agg = d gi W_ih;  d h_prev += d gh W_hh   (one launch)
            Batch bd;
            add_dgrad(bd, r, m.gru_wih, 3 * d.H, d.M, d.M, gi, w.ld3H, R, dagg, w.ldM, nullptr, 0,
                      false);
            if (p > 0)
                add_dgrad(bd, r, m.gru_whh, 3 * d.H, d.H, d.H, gh, w.ld3H, R, dh2, w.ldH, nullptr,
                          0, true);
            flush_batch(r, bd, false);
        }
        if (E > 0 && attn) {
            // backward of the softmax-weighted aggregation: per-edge contributions, then per message
            // row their sum times the SELU derivative of both stacks' last layer (in place)
            const bool p0 = p == 0 && w.D0 > 0;      // pass 0 ran on the class rows
            r.chk(gi_seg_softmax_bwd(ws + w.een[p], ws + w.m[p], w.ldM, p0 ? gp->e2d : in_perm, seg_off,
                                     R, d.M, dagg, w.ldM, ws + w.tmp_en, ws + w.tmp_emb, w.ldM, r.st));
            EdgeChain ch[2] = {
                {m.msg, w.eact[p], w.edz[p], w.ldEh, ws + w.m[p], w.ldM, p > 0 ? ws + w.dxe : nullptr},
                {m.eatt, w.aact[p], w.adz[p], w.ldEa, ws + w.een[p], w.ldM,
                 p > 0 ? ws + w.dxa : nullptr}};
            if (p0) {   // per class row: sum over its (hundreds of) edge slots, both stacks in one launch
                if (r.side && !r.p0_on_main) { kick_deferred(r, dq, r.side, true); flush_bias(r, dq, r.side->st); r.hold_kicks = r.p0_on_main = true; }
                r.chk(gi_class_sum_dselu(ws + w.tmp_emb, ws + w.tmp_en, w.ldM, gp->cls_edges,
                                         gp->cls_off, w.D0, d.M, ws + w.m[p], ws + w.een[p], w.ldM,
                                         r.st));
                edge_chains_backward(r, ws, sp, slabs, dq, ch, 2, bytype0, hx, w.ldhx, gp->d_src, w.D0,
                                     w.ldH, d.H);
            } else {
                // per message row: the sum of its edges' contributions times the SELU derivative of the
                // stack's last layer (own launches in front of the dZ chains, edge_chains_backward)
                const SegIn seg_m{ws + w.tmp_emb, w.ldM, mu_slot, mu_off};
                const SegIn seg_e{ws + w.tmp_en, w.ldM, mu_slot, mu_off};
                ch[0].seg = &seg_m; ch[1].seg = &seg_e;
                edge_chains_backward(r, ws, sp, slabs, dq, ch, 2, bytype, hx, w.ldhx, u_src, U, w.ldH,
                                     d.H);
            }
            if (p > 0 && fuse_scatter) {
                scat0 = ws + w.dxe; scat1 = ws + w.dxa;
            } else if (p > 0) {
                r.chk(gi_seg_sum(ws + w.dxe, w.ldH, out_perm, src_off, R, d.H, dh2, w.ldH, 1, r.st));
                r.chk(gi_seg_sum(ws + w.dxa, w.ldH, out_perm, src_off, R, d.H, dh2, w.ldH, 1, r.st));
            }
        } else if (p == 0 && w.D0 > 0) {
            // pass 0: d m0 = selu'(m0) * (cmat^T . d agg): split-K over the R rows, slabs summed with
            // the SELU backward folded in; then the MLP backward on the D0 class rows (no d h needed)
            if (r.side && !r.p0_on_main) { kick_deferred(r, dq, r.side, true); flush_bias(r, dq, r.side->st); r.hold_kicks = r.p0_on_main = true; }   // nothing queued waits for the pass-0 chain
            gi_gemm_params q;
            gemm_defaults(q);
            q.A = gp->cmat; q.lda = gp->ldc0; q.a_major = 1;
            q.B = dagg; q.ldb = w.ldM; q.b_major = 1;
            q.C = ws + w.p0slab; q.ldc = w.ldM; q.M = w.D0; q.N = d.M; q.K = R;
            q.flags = GI_GEMM_SPLITK; q.nsplit = w.p0split;
            q.c_split_stride = gi_r4l((long long)std::max(w.D0, 1) * w.ldM);
            q.tm = 1; q.tn = 1;
            r.chk(gi_gemm(&q, r.st));
            r.chk(gi_slab_sum_dselu(ws + w.p0slab, w.p0split, q.c_split_stride, w.D0, d.M, w.ldM,
                                    ws + w.m[0], w.ldM, r.st));
            msg_backward(r, ws, sp, slabs, dq, m.msg, bytype0, hx, w.ldhx, gp->d_src, w.D0, w.eact[0],
                         w.edz[0], w.ldEh, ws + w.m[0], w.ldM, nullptr, w.ldH, d.H);
        } else if (E > 0) {
            // d m_u = selu'(m_u) * sum over the edges reading row u of d agg[dst(e)]
            // (backward of the segmented sum + last SELU, over the message CSR)
            const SegIn seg{dagg, w.ldM, mu_dst, mu_off};
            msg_backward(r, ws, sp, slabs, dq, m.msg, bytype, hx, w.ldhx, u_src, U, w.eact[p],
                         w.edz[p], w.ldEh, ws + w.m[p], w.ldM, p > 0 ? ws + w.dxe : nullptr, w.ldH,
                         d.H, &seg);
            if (p > 0 && fuse_scatter)
                scat0 = ws + w.dxe;
            else if (p > 0)   // scatter d h_src back to nodes: segmented sum over the source CSR
                r.chk(gi_seg_sum(ws + w.dxe, w.ldH, out_perm, src_off, R, d.H, dh2, w.ldH, 1, r.st));
        } else {
            // no edges: the message MLP weights still need zeroed slabs
            for (int k = 0; k < (attn ? 2 : 1); ++k)
                for (int t = 0; t < d.Fe; ++t) {
                    const Mlp& q = k ? m.eatt[t] : m.msg[t];
                    for (int l = 0; l < q.layers(); ++l) {
                        SlabEntry& e = sp.e[q.w(l)];
                        r.chk((int)hipMemsetAsync(
                            slabs + e.off + (long long)e.done * e.nsplit * e.stride, 0,
                            sizeof(float) * e.nsplit * e.stride, r.st));
                        e.done++;
                    }
                }
        }
        r.hold_kicks = false;
        std::swap(dh, dh2);
    }
    // ---- all weight-gradient GEMMs, 8 problems per launch, then slabs -> parameter gradients -------
    if (r.side && r.p0_on_main) {
        flush_deferred(r, dq);            // a dozen workgroups, right behind their chain: no hand-over to wait for
        flush_bias(r, dq, r.st);          // (nothing: the pass-0 problems keep their ones column)
        join_side(r, r.side);
    } else if (r.side) {
        kick_deferred(r, dq, r.side, true);
        flush_bias(r, dq, r.side->st);
        join_side(r, r.side);
    } else {
        flush_deferred(r, dq);
        flush_bias(r, dq, r.st);
    }
    gi_reduce_desc descs[160];            // whatever has not been reduced on the side stream yet
    int nd = 0;
    auto add_desc = [&](int widx, int) {
        SlabEntry& e = sp.e[widx];
        if (e.reduced) return;
        e.reduced = 1;
        descs[nd++] = reduce_desc(e, slabs, grads, widx);
    };
    auto add_mlp_desc = [&](const Mlp& q) {
        for (int l = 0; l < q.layers(); ++l) add_desc(q.w(l), q.b(l));
    };
    for (int t = 0; t < d.Fe; ++t) add_mlp_desc(m.msg[t]);
    if (attn)
        for (int t = 0; t < d.Fe; ++t) add_mlp_desc(m.eatt[t]);
    add_desc(m.gru_wih, m.gru_bih);
    add_desc(m.gru_whh, m.gru_bhh);
    add_mlp_desc(m.att); add_mlp_desc(m.emb); add_mlp_desc(m.add1); add_mlp_desc(m.conn1);
    add_mlp_desc(m.add2); add_mlp_desc(m.conn2); add_mlp_desc(m.term2);
    if (r.ok()) r.chk(gi_reduce_slabs(descs, nd, r.st));
    return r.rc;
}
