"""
Test infrastructure: pin the oracle's SELU branches to the ones an implementation under test took.

SELU'(x) jumps from scale*alpha (1.758) to scale (1.051) at x = 0.  Of the ~1e7 activations of a
B = 1000 training step a handful have |x| ~ 1e-7 and round to opposite sides of 0 in two correct fp32
implementations; each such flip moves single gradient tensors by up to ~5e-3 of their max — in the
reference's own fp32 arithmetic too (its fp32 and fp64 runs differ by that much).  A per-tensor 1e-4
gradient comparison therefore runs the ORACLE'S OWN fp32 autograd (`oracle.ggnn_oracle.
forward_backward`) with every SELU branch forced to the sign pattern read back from the
implementation under test (`O.SELU_BRANCH_HOOK`), and counts how many activations that changes: the
count must stay below 1e-6 of all activations, i.e. the pin only resolves measure-zero ties.

The sign patterns arrive in the HIP data layout (compact node rows + zero row, message rows per
(source node, bond type), pass-0 class rows; tests/ref_dataflow.py `compact`) and are mapped here into
the oracle's layout (every bond-type MLP on every edge in nonzero order; all padded node slots;
AttGGNN: neighbour-padded [V, maxdeg]).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch


class Signs:
    """`y > 0` masks of every SELU output of one forward, in the HIP layout.

    msg[p] / eatt[p]: list over layers of bool [rows_p, width] (type-major message rows; pass 0 of
    GGNN: class rows); att / emb / add1 / conn1: lists of bool [R, width]; add2 / conn2 / term2:
    lists of bool [B, width].  `p0[p]` says whether pass p ran on class rows."""

    def __init__(self):
        self.msg: List[List[torch.Tensor]] = []
        self.eatt: List[List[torch.Tensor]] = []
        self.p0: List[bool] = []
        self.node: Dict[str, List[torch.Tensor]] = {}
        self.graph: Dict[str, List[torch.Tensor]] = {}


def signs_from_dataflow(tape, model: str = "GGNN") -> Signs:
    """From a tests/ref_dataflow.py forward tape (CPU stand-in for the HIP path)."""
    s = Signs()
    for ps in tape["passes"]:
        L = len(ps["acts_t"][0])
        s.msg.append([torch.cat([a[l] for a in ps["acts_t"]], 0) > 0 for l in range(L)])
        s.p0.append(bool(ps["p0"]))
        if model == "AttGGNN":
            La = len(ps["aacts_t"][0])
            s.eatt.append([torch.cat([a[l] for a in ps["aacts_t"]], 0) > 0 for l in range(La)])
    for key, name in (("att_acts", "gather.att_nn"), ("emb_acts", "gather.emb_nn"),
                      ("add1", "APDReadout.fAddNet1"), ("conn1", "APDReadout.fConnNet1")):
        s.node[name] = [a > 0 for a in tape[key]]
    for key, name in (("add2", "APDReadout.fAddNet2"), ("conn2", "APDReadout.fConnNet2"),
                      ("term2", "APDReadout.fTermNet2")):
        s.graph[name] = [a > 0 for a in tape[key]]
    return s


def signs_from_hip(dims, graph, ws, out, attn: bool) -> Signs:
    """From the workspace of a `ggnn_forward_raw` call (read through gi_ggnn_ws_query)."""
    from graphinvent_amd import ops
    s = Signs()
    R, U, B = graph.S + 1, graph.U, out.shape[0]

    def view(name, rows, width, i=0, j=0):
        return (ops.ws_view(ws, dims, graph, name, rows, i, j)[:, :width] > 0).cpu()

    for p in range(dims.passes):
        p0 = (p == 0 and graph.D0 > 0)                  # pass 0 runs on the class rows (both models)
        rows = graph.D0 if p0 else U
        s.p0.append(p0)
        s.msg.append([view("eact", rows, dims.enn_hidden, p, l) for l in range(dims.enn_depth)] +
                     [view("m", rows, dims.M, p)])
        if attn:
            s.eatt.append([view("aact", rows, dims.eatt_hidden, p, l) for l in range(dims.eatt_depth)] +
                          [view("een", rows, dims.M, p)])
    for name, act, last, depth, hid, width in (
            ("gather.att_nn", "att_act", "en", dims.att_depth, dims.att_hidden, dims.G),
            ("gather.emb_nn", "emb_act", "emb", dims.emb_depth, dims.emb_hidden, dims.G),
            ("APDReadout.fAddNet1", "add1_act", "add1", dims.mlp1_depth, dims.mlp1_hidden, dims.A),
            ("APDReadout.fConnNet1", "conn1_act", "conn1", dims.mlp1_depth, dims.mlp1_hidden, dims.C)):
        s.node[name] = [view(act, R, hid, 0, l) for l in range(depth)] + [view(last, R, width)]
    NA, NC = dims.N * dims.A, dims.N * dims.C
    o = (out > 0).cpu()
    for name, act, cols in (("APDReadout.fAddNet2", "add2_act", slice(0, NA)),
                            ("APDReadout.fConnNet2", "conn2_act", slice(NA, NA + NC)),
                            ("APDReadout.fTermNet2", "term2_act", slice(NA + NC, NA + NC + 1))):
        s.graph[name] = [view(act, B, dims.mlp2_hidden, 0, l) for l in range(dims.mlp2_depth)] + \
                        [o[:, cols]]
    return s


TIE_TOL = 1e-5          # |pre-activation| below which two fp32 evaluations may legitimately disagree on x > 0


class OraclePins:
    """The callable to install as `oracle.ggnn_oracle.SELU_BRANCH_HOOK` around ONE oracle forward.

    g: the compact-graph index arrays as numpy (keys of tests/ref_dataflow.compact: cidx, in_perm,
    type_off, type_off0, u_src, d_src, slot_of, S, E, U, D0), nodes / edges: the batch (numpy)."""

    def __init__(self, signs: Signs, g: dict, nodes: np.ndarray, edges: np.ndarray, model: str = "GGNN"):
        self.s, self.model = signs, model
        B, N, Fn = nodes.shape
        Fe = edges.shape[3]
        self.B, self.N, self.Fe = B, N, Fe
        adj = edges.sum(3) != 0
        eb, ei, ej = np.nonzero(adj)                       # the oracle's edge order (row-major nonzero)
        self.E = eb.size
        self.etype = edges[eb, ei, ej, :].argmax(1)
        in_perm = np.asarray(g["in_perm"]).astype(np.int64)
        assert in_perm.size == self.E
        self.edge_row = torch.from_numpy(in_perm)          # oracle edge k -> message row
        self.cidx = torch.from_numpy(np.asarray(g["cidx"]).astype(np.int64))
        # pass-0 class rows: message row u -> class row (same bond type, same source feature row)
        self.edge_row0: Optional[torch.Tensor] = None
        if any(signs.p0):
            feat = nodes.reshape(B * N, Fn)
            slot_of = np.asarray(g["slot_of"]).astype(np.int64)
            S = int(g["S"])

            def feat_of(crow):                              # feature row of a compact row
                return feat[slot_of[crow]].tobytes() if crow < S else b""

            toff0 = np.asarray(g["type_off0"]).astype(np.int64)
            d_src = np.asarray(g["d_src"]).astype(np.int64)
            table = {}
            for t in range(Fe):
                for d in range(toff0[t], toff0[t + 1]):
                    table[(t, feat_of(d_src[d]))] = d
            toff = np.asarray(g["type_off"]).astype(np.int64)
            u_src = np.asarray(g["u_src"]).astype(np.int64)
            u2d = np.empty(int(g["U"]), dtype=np.int64)
            for t in range(Fe):
                for u in range(toff[t], toff[t + 1]):
                    u2d[u] = table[(t, feat_of(u_src[u]))]
            self.edge_row0 = torch.from_numpy(u2d[in_perm])
        if model == "AttGGNN":                             # neighbour-padded layout of the oracle
            deg = adj.sum(2)[adj.sum(2) > 0].astype(np.int64)          # per node with edges, in order
            self.V, self.maxdeg = deg.size, int(deg.max()) if deg.size else 0
            self.node_of_edge = torch.from_numpy(np.repeat(np.arange(self.V), deg))
            self.slot_in_node = torch.from_numpy(
                np.concatenate([np.arange(k) for k in deg]) if deg.size else np.zeros(0, np.int64))
        self.calls: Dict[tuple, int] = {}
        self.total = 0
        self.flipped = 0
        self.max_flipped_abs = 0.0         # largest |pre-activation| (the ORACLE's own) the pin decided differently

    def _account(self, mask, x):
        self.total += mask.numel()
        moved = mask != (x > 0)
        n = int(moved.sum())
        self.flipped += n
        if n:
            self.max_flipped_abs = max(self.max_flipped_abs, float(x.detach()[moved].abs().max()))
        return mask

    def __call__(self, prefix: str, layer: int, x: torch.Tensor):
        k = (prefix, layer)
        call = self.calls.get(k, 0)
        self.calls[k] = call + 1
        natural = x > 0
        if prefix.startswith("msg_nns.") or prefix.startswith("att_nns."):
            t = int(prefix.split(".")[1])
            p = call
            src = self.s.msg if prefix.startswith("msg_nns.") else self.s.eatt
            rowmap = self.edge_row0 if self.s.p0[p] else self.edge_row
            hip = src[p][layer][rowmap]                    # [E, width] in the oracle's edge order
            is_t = torch.from_numpy(self.etype == t)
            if self.model == "AttGGNN":                    # x: [V, maxdeg, width]
                mask = natural.clone()
                sel = is_t.nonzero(as_tuple=True)[0]
                mask[self.node_of_edge[sel], self.slot_in_node[sel]] = hip[sel]
            else:                                          # x: [E, width]
                mask = torch.where(is_t[:, None], hip, natural)
            return self._account(mask, x)
        if prefix in self.s.node:                          # x: [B, N, width] over all padded slots
            hip = self.s.node[prefix][layer][self.cidx]
            return self._account(hip.view(self.B, self.N, -1), x)
        if prefix in self.s.graph:                         # x: [B, width]
            return self._account(self.s.graph[prefix][layer], x)
        raise KeyError(prefix)


def graph_arrays(graph) -> dict:
    """Index arrays of a graphinvent_amd.ops.CompactGraph as numpy (the keys OraclePins reads)."""
    names = ("cidx", "in_perm", "type_off", "type_off0", "u_src", "d_src", "slot_of")
    g = {n: getattr(graph, n).cpu().numpy() for n in names}
    g.update(S=graph.S, E=graph.E, U=graph.U, D0=graph.D0)
    return g


class MaskQuantumPin:
    """The callable to install as `oracle.ggnn_oracle.MASK_QUANTUM_HOOK` around ONE oracle forward.

    A graph whose every slot is masked (empty / single atom) takes its attention softmax over
    fl32(e - 1e6) (gnn/modules.py:47-49): energies on a 1/16 grid.  The implementation under test
    evaluates the same literal expression on ITS e; where its e and the oracle's straddle a grid
    midpoint the two land one quantum apart and the graph's logits move by up to ~3e-3 — in the
    reference's own fp32-vs-fp64 comparison too.  The pin hands the oracle, for those graphs only, the
    quanta computed from the implementation's energies `en` ([R, G] pre-mask outputs of gather.att_nn per
    compact row, read back from its workspace) and keeps the books that make the substitution a tie-break
    and nothing more:
      quanta / moved : slots x features of fully-masked graphs, and how many of them differ from the
                       oracle's own quantum;
      max_steps      : the largest |difference| in quanta (1 = a rounding tie);
      max_de         : max |e_impl - e_oracle| over those slots (the un-quantised agreement)."""

    def __init__(self, en: torch.Tensor, cidx, B: int, N: int, big: float):
        self.en = en.detach().float().cpu()
        self.cidx = torch.as_tensor(np.asarray(cidx).astype(np.int64))
        self.B, self.N, self.big = B, N, float(big)
        self.quanta = self.moved = 0
        self.max_steps = 0.0
        self.max_de = 0.0
        self.graphs = 0

    def __call__(self, raw: torch.Tensor, masked: torch.Tensor, node_mask: torch.Tensor):
        if raw.dtype != torch.float32:
            return None                                    # fp64 oracle runs: no quantisation to pin
        full = (node_mask.view(self.B, self.N) == 0).all(1)            # fully-masked graphs
        self.graphs = int(full.sum())
        if self.graphs == 0:
            return None
        G = raw.shape[-1]
        e_impl = self.en[self.cidx][:, :G].view(self.B, self.N, G)
        q_impl = e_impl - torch.tensor(self.big, dtype=torch.float32)  # the literal fl32(e - 1e6)
        sel = full[:, None, None].expand_as(masked)
        d = (q_impl - masked)[sel]
        self.quanta = int(d.numel())
        self.moved = int((d != 0).sum())
        ulp = float(torch.tensor(self.big, dtype=torch.float32).abs().frexp()[1] - 24)   # log2 ulp at big
        self.max_steps = float(d.abs().max()) / 2.0 ** ulp if d.numel() else 0.0
        self.max_de = float((e_impl - raw)[sel].abs().max()) if d.numel() else 0.0
        return torch.where(sel, q_impl, masked)


def oracle_pinned(O, P, cfg, nodes, edges, target, signs: Signs, g: dict, model: str = "GGNN",
                  mask_pin: Optional[MaskQuantumPin] = None, upstream=None):
    """`O.forward_backward` in fp32 with the SELU branches of `signs` (and, with `mask_pin`, the energy
    quanta of fully-masked graphs); returns (logits, loss, grads, flipped activations, all activations).
    upstream: see O.forward_backward (grads = J^T . upstream)."""
    pins = OraclePins(signs, g, nodes.numpy(), edges.numpy(), model)
    O.SELU_BRANCH_HOOK = pins
    O.MASK_QUANTUM_HOOK = mask_pin
    try:
        out, loss, grads = O.forward_backward(P, cfg, nodes, edges, target, model, upstream=upstream)
    finally:
        O.SELU_BRANCH_HOOK = None
        O.MASK_QUANTUM_HOOK = None
    # a pin may only resolve TIES at the kink: every activation whose branch it changed must be a rounding-noise
    # distance from 0 in the oracle's own evaluation (pre-activations are O(1), two correct fp32 evaluations differ
    # by ~1e-6); a wrong index map, or an implementation error that moves a real activation across 0, fails here
    assert pins.max_flipped_abs < TIE_TOL, (pins.max_flipped_abs, pins.flipped, pins.total)
    return out, loss, grads, pins.flipped, pins.total


def oracle_quantum_pinned_logits(O, P, cfg, nodes, edges, mask_pin: MaskQuantumPin, model: str = "GGNN"):
    """Plain fp32 oracle forward with only the energy quanta of fully-masked graphs pinned."""
    O.MASK_QUANTUM_HOOK = mask_pin
    try:
        with torch.no_grad():
            return O.FORWARDS[model](P, cfg, nodes, edges)
    finally:
        O.MASK_QUANTUM_HOOK = None


def mask_pin_from_hip(dims, graph, ws, B: int, big: float) -> MaskQuantumPin:
    """From the workspace of a `ggnn_forward_raw` call: the pre-mask attention energies per compact row."""
    from graphinvent_amd import ops
    en = ops.ws_view(ws, dims, graph, "en", graph.S + 1)[:, :dims.G].cpu()
    return MaskQuantumPin(en, graph.cidx.cpu().numpy(), B, dims.N, big)


def assert_masked_rows_are_quantum_ties(O, model, cfg, P, n8, e8, out, model_name: str = "GGNN", tol: float = 1e-4):
    """For a forward of the HIP `model` on (n8, e8) with logits `out`: the rows of fully-masked graphs, which the
    goldens / the plain oracle only bound at 5e-3 (fl32(e - 1e6) energy quantisation, gnn/modules.py:47-48), agree
    with the oracle at `tol` once the oracle takes the quanta the HIP forward computed (MaskQuantumPin): every
    difference beyond 1e-4 on those rows is a 1/16 quantum tie and nothing else.  Runs one more (no_grad) HIP
    forward through ggnn_forward_raw to read the energies back.  Returns the pin (books)."""
    from graphinvent_amd import lib as L
    from graphinvent_amd.gnn import mpnn
    kind = L.KIND_ATTGGNN if model_name == "AttGGNN" else L.KIND_GGNN
    dev = next(model.parameters()).device
    nodes = torch.from_numpy(np.ascontiguousarray(n8)).float().to(dev)
    edges = torch.from_numpy(np.ascontiguousarray(e8)).float().to(dev)
    with torch.no_grad():
        out2, (dims, graph, ws) = mpnn.ggnn_forward_raw(model.constants, nodes, edges, list(model.parameters()), kind)
    out = torch.as_tensor(out).detach().float().cpu()
    assert torch.equal(out2.cpu(), out), "the HIP forward is deterministic: same logits as the caller's"
    pin = mask_pin_from_hip(dims, graph, ws, n8.shape[0], cfg["big_positive"])
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).float()
    ref = oracle_quantum_pinned_logits(O, P, cfg, t(n8), t(e8), pin, model_name)
    masked = np.nonzero(~e8.reshape(e8.shape[0], -1).any(1))[0]
    assert pin.graphs == len(masked)
    if len(masked):
        scale = max(float(ref.abs().max()), 1e-30)
        err = float((out[masked] - ref[masked]).abs().max()) / scale
        assert err < tol, err
        assert pin.max_steps <= 1 and pin.max_de < 2e-5, (pin.max_steps, pin.max_de)
    return pin
