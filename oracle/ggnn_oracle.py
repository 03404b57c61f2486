"""
ORACLE — test infrastructure only.  NOT part of the product path.

CPU restatement (plain torch-CPU ops, functional style, fp32 or fp64) of the algorithm the
reference implements for the GGNN message-passing + APD-readout hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this file; it is
the checker and the reported CPU baseline, never the thing shipped.

Where the arithmetic lives: the reference (``/root/reference/graphinvent/gnn/*``) is pure Python
on third-party PyTorch (pinned ``pytorch=1.8.0`` in ``environments/graphinvent.yml:95``; ATen CPU
``addmm``/``elu``/``sigmoid``/``tanh``/``_softmax``/``index``/``nonzero``).  This file restates the
reference's *call sequence* on the same ATen ops, deliberately keeping the reference's wasteful
dataflow (dense [V,E] summation matrix, every edge-type MLP evaluated on every edge) so that timing
it is a fair stand-in for timing the reference on CPU (``cpu_baseline.kind == "port"``).

Parity pinning: the reference ships NO tests and NO golden vectors for this path (SURVEY.md §4,
§8c).  The oracle is therefore pinned against *outputs of the reference itself*, generated in the
build container by ``tests/golden/make_golden.py`` (which imports the unmodified reference
``gnn.mpnn.GGNN`` from ``/root/reference/graphinvent``) and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` replays them.

Every function cites the reference file:line (relative to ``/root/reference/graphinvent/``) it
follows.
"""
from __future__ import annotations

import math
from collections import OrderedDict, namedtuple
from typing import Dict, List, Tuple

import numpy as np
import torch

# ----------------------------------------------------------------------------------------------
# configuration (what the reference passes around as the ``constants`` namedtuple)
# ----------------------------------------------------------------------------------------------

#: default GGNN hyper-parameters — parameters/defaults.py:280-300; derived dims for the shipped
#: GDB-13 preprocessing (5 atom types, 3 formal charges, ignore_H, no chirality, 3 bond types):
#: parameters/constants.py:159-211.
GDB13_DEFAULTS = dict(
    device="cpu",
    big_positive=1e6,
    big_negative=-1e6,
    n_node_features=8,
    n_edge_features=3,
    max_n_nodes=13,
    len_f_add_per_node=45,
    len_f_conn_per_node=3,
    hidden_node_features=100,
    message_size=100,
    message_passes=3,
    enn_depth=4,
    enn_hidden_dim=250,
    enn_dropout_p=0.0,
    gather_width=100,
    gather_att_depth=4,
    gather_att_hidden_dim=250,
    gather_att_dropout_p=0.0,
    gather_emb_depth=4,
    gather_emb_hidden_dim=250,
    gather_emb_dropout_p=0.0,
    mlp1_depth=4,
    mlp1_hidden_dim=500,
    mlp1_dropout_p=0.0,
    mlp2_depth=4,
    mlp2_hidden_dim=500,
    mlp2_dropout_p=0.0,
    # AttGGNN only (parameters/defaults.py:340-363): per-bond-type message and attention MLPs
    msg_depth=4,
    msg_hidden_dim=250,
    msg_dropout_p=0.0,
    att_depth=4,
    att_hidden_dim=250,
    att_dropout_p=0.0,
)


def make_config(**overrides) -> dict:
    """Hyper-parameter dict with the reference's GGNN/GDB-13 defaults, overridable per key."""
    cfg = dict(GDB13_DEFAULTS)
    unknown = set(overrides) - set(cfg)
    if unknown:
        raise KeyError(f"unknown config keys: {sorted(unknown)}")
    cfg.update(overrides)
    return cfg


def shaped_config(n_atom_types: int, n_formal_charge: int, max_n_nodes: int,
                  n_edge_features: int = 3, **overrides) -> dict:
    """Config for a dataset shape.  len_f_add_per_node = atom types x charges x bond types when
    H / chirality features are off (parameters/constants.py:56-89,184)."""
    return make_config(
        n_node_features=n_atom_types + n_formal_charge,
        n_edge_features=n_edge_features,
        max_n_nodes=max_n_nodes,
        len_f_add_per_node=n_atom_types * n_formal_charge * n_edge_features,
        len_f_conn_per_node=n_edge_features,
        **overrides,
    )


def as_constants(cfg: dict):
    """The namedtuple form the reference model constructors take (constants.py:259-260)."""
    return namedtuple("CONSTANTS", sorted(cfg))(**cfg)


def apd_width(cfg: dict) -> int:
    n = cfg["max_n_nodes"]
    return n * cfg["len_f_add_per_node"] + n * cfg["len_f_conn_per_node"] + 1


# ----------------------------------------------------------------------------------------------
# parameters: names, shapes, deterministic initialisation
# ----------------------------------------------------------------------------------------------

def _mlp_shapes(prefix: str, fan_in: int, hidden: int, depth: int, fan_out: int):
    """gnn/modules.py:126-142 — `depth` hidden layers + one output layer; inside the Sequential
    the Linear modules sit at indices 0,3,6,... (Linear, SELU, AlphaDropout triples)."""
    sizes = [fan_in] + [hidden] * depth + [fan_out]
    out = []
    for layer, (i, o) in enumerate(zip(sizes, sizes[1:])):
        out.append((f"{prefix}.seq.{3 * layer}.weight", (o, i)))
        out.append((f"{prefix}.seq.{3 * layer}.bias", (o,)))
    return out


def param_shapes(cfg: dict, model: str = "GGNN") -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys and shapes of the reference ``GGNN`` (gnn/mpnn.py:238-282) or
    ``AttentionGGNN`` (gnn/mpnn.py:313-368) in registration order (gnn/modules.py:24-37, 193-235).
    AttentionGGNN registers ``msg_nns`` before ``att_nns`` (both ModuleLists are created first,
    :316-317) although the MLPs are constructed interleaved (:319-335)."""
    H, M, G = cfg["hidden_node_features"], cfg["message_size"], cfg["gather_width"]
    Fn, Fe, N = cfg["n_node_features"], cfg["n_edge_features"], cfg["max_n_nodes"]
    A, C = cfg["len_f_add_per_node"], cfg["len_f_conn_per_node"]
    items: List[Tuple[str, Tuple[int, ...]]] = []
    if model == "AttGGNN":
        for t in range(Fe):
            items += _mlp_shapes(f"msg_nns.{t}", H, cfg["msg_hidden_dim"], cfg["msg_depth"], M)
        for t in range(Fe):
            items += _mlp_shapes(f"att_nns.{t}", H, cfg["att_hidden_dim"], cfg["att_depth"], M)
    else:
        for t in range(Fe):
            items += _mlp_shapes(f"msg_nns.{t}", H, cfg["enn_hidden_dim"], cfg["enn_depth"], M)
    items += [("gru.weight_ih", (3 * H, M)), ("gru.weight_hh", (3 * H, H)),
              ("gru.bias_ih", (3 * H,)), ("gru.bias_hh", (3 * H,))]
    items += _mlp_shapes("gather.att_nn", Fn + H, cfg["gather_att_hidden_dim"],
                         cfg["gather_att_depth"], G)
    items += _mlp_shapes("gather.emb_nn", H, cfg["gather_emb_hidden_dim"],
                         cfg["gather_emb_depth"], G)
    items += _mlp_shapes("APDReadout.fAddNet1", H, cfg["mlp1_hidden_dim"], cfg["mlp1_depth"], A)
    items += _mlp_shapes("APDReadout.fConnNet1", H, cfg["mlp1_hidden_dim"], cfg["mlp1_depth"], C)
    items += _mlp_shapes("APDReadout.fAddNet2", N * A + G, cfg["mlp2_hidden_dim"],
                         cfg["mlp2_depth"], N * A)
    items += _mlp_shapes("APDReadout.fConnNet2", N * C + G, cfg["mlp2_hidden_dim"],
                         cfg["mlp2_depth"], N * C)
    items += _mlp_shapes("APDReadout.fTermNet2", G, cfg["mlp2_hidden_dim"], cfg["mlp2_depth"], 1)
    return OrderedDict(items)


def init_params(cfg: dict, seed: int = 0, dtype=torch.float32,
                model: str = "GGNN") -> "OrderedDict[str, torch.Tensor]":
    """Deterministic parameters with the reference's init *distributions* (Xavier-uniform Linear
    weights gnn/modules.py:163, torch-default Linear bias and GRUCell U(-1/sqrt(fan),+)), drawn
    from numpy's PCG64 so the same seed gives the same weights on every machine and torch
    version.  (Not seed-for-seed equal to ``torch.manual_seed`` + the reference ctor.)"""
    rng = np.random.default_rng(seed)
    H = cfg["hidden_node_features"]
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    shapes = param_shapes(cfg, model)
    for key, shape in shapes.items():
        if key.startswith("gru."):
            bound = 1.0 / math.sqrt(H)
        elif key.endswith(".weight"):
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
        else:
            fan_in = shapes[key[:-4] + "weight"][1]
            bound = 1.0 / math.sqrt(fan_in)
        arr = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        out[key] = torch.from_numpy(arr).to(dtype)
    return out


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------

SELU_ALPHA = 1.6732632423543772848170429916717
SELU_SCALE = 1.0507009873554804934193349852946

#: Test-only hook (tests/pins.py), None in every timed / baseline use.  SELU' jumps from
#: scale*alpha (1.758) to scale (1.051) at 0, so two correct fp32 implementations that round an
#: activation of |x| ~ 1e-7 to opposite sides of 0 differ by O(1) in that element's derivative.  A
#: gradient comparison at the 1e-4 bar therefore pins the branch: the hook is called as
#: ``hook(prefix, layer, pre_activation)`` and returns a bool mask "take the x > 0 branch" (or None
#: for the natural ``x > 0``); the oracle's own autograd then differentiates that branch.
SELU_BRANCH_HOOK = None


def _selu(x: torch.Tensor, prefix: str, layer: int) -> torch.Tensor:
    mask = SELU_BRANCH_HOOK(prefix, layer, x) if SELU_BRANCH_HOOK is not None else None
    if mask is None:
        return torch.selu(x)
    # (the exponential of the branch NOT taken must stay finite: exp(x) = inf for x > 88.7 — activations a trained model
    # does reach — turns autograd's 0 * inf into NaN; a pin only ever moves ties |x| < 1e-5 onto this branch)
    return SELU_SCALE * torch.where(mask, x, SELU_ALPHA * (torch.exp(torch.clamp(x, max=1.0)) - 1.0))


#: Test-only: ``torch.nn.AlphaDropout(dropout_p)`` in TRAINING mode (gnn/modules.py:142; identity in
#: eval mode and at the shipped default p = 0).  None, or a callable ``(prefix, layer, x) -> None |
#: (p, keep)``: ``keep`` a bool tensor shaped like x (the masks an implementation under test drew,
#: tests/dropout_masks.py) or None for torch's own generator.
DROPOUT_HOOK = None
ALPHA_DROPOUT_ALPHA = 1.7580993408473766     # -selu_scale * selu_alpha, ATen Dropout.cpp


def _alpha_dropout(x: torch.Tensor, prefix: str, layer: int) -> torch.Tensor:
    """ATen's ``_dropout_impl<feature=false, alpha=true>`` with a given keep mask, op for op (so the
    roundings are torch's):  a = ((alpha^2 p + 1)(1 - p))^-1/2,  b = (keep - 1) * alpha a + alpha a p,
    y = x * (keep * a) + b."""
    spec = DROPOUT_HOOK(prefix, layer, x) if DROPOUT_HOOK is not None else None
    if spec is None:
        return x
    p, keep = spec
    if p == 0:
        return x
    if keep is None:
        return torch.nn.functional.alpha_dropout(x, p, training=True)
    alpha = ALPHA_DROPOUT_ALPHA
    a = 1.0 / math.sqrt((alpha * alpha * p + 1) * (1 - p))
    noise = keep.to(x.dtype)
    b = noise.add(-1).mul_(alpha * a).add_(alpha * a * p)
    noise = noise.mul(a)
    return x * noise + b


def mlp(P: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor) -> torch.Tensor:
    """gnn/modules.py:111-170 — Linear -> SELU -> AlphaDropout(p) (identity in eval mode and at the
    default p = 0) for every layer *including the last*."""
    layer = 0
    while f"{prefix}.seq.{3 * layer}.weight" in P:
        x = _selu(torch.nn.functional.linear(
            x, P[f"{prefix}.seq.{3 * layer}.weight"], P[f"{prefix}.seq.{3 * layer}.bias"]),
            prefix, layer)
        x = _alpha_dropout(x, prefix, layer)
        layer += 1
    return x


def gru_cell(P: Dict[str, torch.Tensor], x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """torch.nn.GRUCell as used at gnn/mpnn.py:249-253,296-297.  Gate rows are ordered r,z,n;
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1-z)*n + z*h."""
    H = h.shape[1]
    gi = torch.nn.functional.linear(x, P["gru.weight_ih"], P["gru.bias_ih"])
    gh = torch.nn.functional.linear(h, P["gru.weight_hh"], P["gru.bias_hh"])
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1.0 - z) * n + z * h


#: Test-only hook (tests/pins.py), None in every timed / baseline use.  For a graph whose EVERY slot is
#: masked (empty / single atom: no node has an edge, gnn/summation_mpnn.py:146) the softmax of
#: gnn/modules.py:47-49 runs on fl32(e - 1e6), i.e. on energies quantised to 1/16 (ulp of fp32 at 1e6);
#: two correct fp32 implementations whose e differ by 1e-7 put an e that sits within rounding of a grid
#: midpoint on different quanta, and that graph's outputs then differ by up to ~3e-3 (the reference's own
#: fp32 and fp64 runs do).  The hook is called as ``hook(e, fl32(e - big * masked), node_mask)`` with
#: detached tensors and returns the masked energies to use instead (or None): the pin of tests/pins.py
#: substitutes, on fully-masked graphs only, the quanta an implementation under test computed.
MASK_QUANTUM_HOOK = None


def graph_gather(P, cfg, hidden, nodes, node_mask) -> torch.Tensor:
    """gnn/modules.py:39-52 — attention energies from MLP_att(cat(h, x)), minus big_positive on
    masked slots (added in working precision *before* the softmax), softmax over the node axis
    per feature, weighted sum of MLP_emb(h)."""
    dtype = hidden.dtype
    cat = torch.cat((hidden, nodes), dim=2)
    energy_mask = (node_mask == 0).to(dtype) * cfg["big_positive"]
    raw = mlp(P, "gather.att_nn", cat)
    energies = raw - energy_mask.unsqueeze(-1)
    if MASK_QUANTUM_HOOK is not None:
        pinned = MASK_QUANTUM_HOOK(raw.detach(), energies.detach(), node_mask)
        if pinned is not None:          # exact: both sit on the 1/16 grid around -1e6; gradient path unchanged
            energies = energies + (pinned - energies.detach())
    attention = torch.softmax(energies, dim=1)
    embedding = mlp(P, "gather.emb_nn", hidden)
    return torch.sum(attention * embedding, dim=1)


def global_readout(P, hidden, graph_emb) -> torch.Tensor:
    """gnn/modules.py:237-281 — tier 1 on every (padded) node slot, flatten per graph, tier 2 on
    [flattened tier-1 output, graph embedding]; concatenate f_add, f_conn, f_term."""
    B = hidden.shape[0]
    f_add_1 = mlp(P, "APDReadout.fAddNet1", hidden).reshape(B, -1)
    f_conn_1 = mlp(P, "APDReadout.fConnNet1", hidden).reshape(B, -1)
    f_add_2 = mlp(P, "APDReadout.fAddNet2", torch.cat((f_add_1, graph_emb), dim=1))
    f_conn_2 = mlp(P, "APDReadout.fConnNet2", torch.cat((f_conn_1, graph_emb), dim=1))
    f_term_2 = mlp(P, "APDReadout.fTermNet2", graph_emb)
    return torch.cat((f_add_2, f_conn_2, f_term_2), dim=1)


def message_terms(P, cfg, nghb_hidden, edge_onehot) -> torch.Tensor:
    """gnn/mpnn.py:284-294 — every edge-type MLP runs on every edge row; its input and output are
    both multiplied by that edge's indicator for the type; the Fe results are summed."""
    total = None
    for t in range(cfg["n_edge_features"]):
        gate = edge_onehot[:, t:t + 1]
        term = gate * mlp(P, f"msg_nns.{t}", gate * nghb_hidden)
        total = term if total is None else total + term
    return total


# ----------------------------------------------------------------------------------------------
# the hot path
# ----------------------------------------------------------------------------------------------

def ggnn_forward(P: Dict[str, torch.Tensor], cfg: dict, nodes: torch.Tensor,
                 edges: torch.Tensor) -> torch.Tensor:
    """``GGNN.forward`` = ``SummationMPNN.forward`` (gnn/summation_mpnn.py:80-149) with the GGNN
    hooks (gnn/mpnn.py:284-303).  nodes [B,N,Fn], edges [B,N,N,Fe] -> APD logits
    [B, N*A + N*Fe + 1] (SELU'd, no softmax).  Computes in ``nodes.dtype``."""
    dtype = nodes.dtype
    H = cfg["hidden_node_features"]
    adjacency = edges.sum(dim=3)                                         # :100
    eb, ei, ej = adjacency.nonzero(as_tuple=True)                        # :103-105 (row-major order)
    nb, ni = adjacency.sum(-1).nonzero(as_tuple=True)                    # :107
    # dense 0/1 [V,E] matrix: entry (v,e) set when edge e arrives at node v    :109-114
    summation = ((nb.view(-1, 1) == eb) & (ni.view(-1, 1) == ei)).to(dtype)
    edge_onehot = edges[eb, ei, ej, :]                                   # :116
    hidden = torch.zeros(nodes.shape[0], nodes.shape[1], H, dtype=dtype)  # :119-123
    hidden[:, :, :nodes.shape[2]] = nodes
    node_rows = hidden[nb, ni, :]                                        # :124
    for _ in range(cfg["message_passes"]):                               # :126
        nghb = hidden[eb, ej, :]                                         # :129
        terms = message_terms(P, cfg, nghb, edge_onehot)                 # :131-133
        messages = summation @ terms                                     # :139
        node_rows = gru_cell(P, messages, node_rows)                     # :141 -> mpnn.py:296
        hidden = hidden.clone()
        hidden[nb, ni, :] = node_rows                                    # :142
    node_mask = adjacency.sum(-1) != 0                                   # :144
    graph_emb = graph_gather(P, cfg, hidden, nodes, node_mask)           # mpnn.py:301
    return global_readout(P, hidden, graph_emb)                          # mpnn.py:302


def attggnn_forward(P: Dict[str, torch.Tensor], cfg: dict, nodes: torch.Tensor,
                    edges: torch.Tensor) -> torch.Tensor:
    """``AttentionGGNN.forward`` = ``AggregationMPNN.forward`` (gnn/aggregation_mpnn.py:83-168)
    with ``aggregate_message`` (gnn/mpnn.py:370-389): neighbour-padded [V, maxdeg, .] layout, every
    bond-type MLP (message and attention) on every padded neighbour slot, gated by the slot's
    bond-type indicator, energies of padding slots pushed down by big_positive, softmax over the
    neighbour axis per feature, weighted sum, GRU update of the nodes that have neighbours."""
    dtype = nodes.dtype
    H = cfg["hidden_node_features"]
    adjacency = edges.sum(dim=3)                                                  # :113
    eb, ei, ej = adjacency.nonzero(as_tuple=True)                                 # :116-117
    nb, ni = adjacency.sum(-1).nonzero(as_tuple=True)                             # :119
    degrees = adjacency[nb, ni, :].sum(-1).long()                                 # :120-122
    V = nb.shape[0]
    maxdeg = int(degrees.max()) if V else 0                                       # :123
    slot_in_node = torch.cat([torch.arange(int(k)) for k in degrees]) if V else eb   # :134-136
    node_of_edge = torch.repeat_interleave(torch.arange(V), degrees)              # :138-140
    nghb_mask = torch.zeros(V, maxdeg, dtype=dtype)                               # :142-146
    nghb_mask[node_of_edge, slot_in_node] = 1
    nghb_edges = torch.zeros(V, maxdeg, cfg["n_edge_features"], dtype=dtype)      # :128-131,148-149
    nghb_edges[node_of_edge, slot_in_node, :] = edges[eb, ei, ej, :]
    hidden = torch.zeros(nodes.shape[0], nodes.shape[1], H, dtype=dtype)          # :152-156
    hidden[:, :, :nodes.shape[2]] = nodes
    for _ in range(cfg["message_passes"]):                                        # :158
        node_rows = hidden[nb, ni, :]                                             # :160
        nghb_hidden = torch.zeros(V, maxdeg, H, dtype=dtype)                      # :124-127,161-162
        nghb_hidden[node_of_edge, slot_in_node, :] = hidden[eb, ej, :]
        # aggregate_message, gnn/mpnn.py:370-389
        energy_mask = (nghb_mask == 0).to(dtype) * cfg["big_positive"]
        emb = None
        en = None
        for t in range(cfg["n_edge_features"]):
            gate = nghb_edges[:, :, t].unsqueeze(-1)
            e_t = gate * mlp(P, f"msg_nns.{t}", nghb_hidden)
            a_t = gate * mlp(P, f"att_nns.{t}", nghb_hidden)
            emb = e_t if emb is None else emb + e_t
            en = a_t if en is None else en + a_t
        attention = torch.softmax(en - energy_mask.unsqueeze(-1), dim=1)
        messages = torch.sum(attention * emb, dim=1)
        new_rows = gru_cell(P, messages, node_rows)                               # :169-170
        hidden = hidden.clone()
        hidden[nb, ni, :] = new_rows
    node_mask = adjacency.sum(-1) != 0                                            # :164
    graph_emb = graph_gather(P, cfg, hidden, nodes, node_mask)
    return global_readout(P, hidden, graph_emb)


FORWARDS = {"GGNN": ggnn_forward, "AttGGNN": None}   # filled below


def kl_loss(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Workflow.py:850-858 — KLDivLoss(batchmean)(log_softmax(out), target / sum(target))."""
    logp = torch.log_softmax(output, dim=1)
    tgt = target / torch.sum(target, dim=1, keepdim=True)
    return torch.nn.functional.kl_div(logp, tgt, reduction="batchmean")


FORWARDS["AttGGNN"] = attggnn_forward


def forward_backward(P, cfg, nodes, edges, target, model: str = "GGNN", upstream=None):
    """One forward + loss + backward; returns (logits, loss, grads-by-key).  Train-step order of
    Workflow.py:785-796 up to (not including) the optimizer.  upstream != None: the gradients are J^T . upstream — the
    backward operator applied to a GIVEN d loss / d logits instead of this evaluation's own (tests: the backward's
    arithmetic apart from the conditioning of softmax - target on a fitted model)."""
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in P.items())
    out = FORWARDS[model](leaves, cfg, nodes, edges)
    loss = kl_loss(out, target)
    if upstream is not None:
        grads = torch.autograd.grad(out, list(leaves.values()), grad_outputs=upstream.to(out.dtype))
    else:
        grads = torch.autograd.grad(loss, list(leaves.values()))
    return out.detach(), loss.detach(), OrderedDict(zip(leaves.keys(), grads))


class OracleGGNN(torch.nn.Module):
    """nn.Module wrapper over the functional oracle (same state_dict keys as the reference) so
    tests and the CPU-baseline timer can drive it with a torch optimizer."""

    def __init__(self, cfg: dict, seed: int = 0, model: str = "GGNN"):
        super().__init__()
        self.cfg = dict(cfg)
        self.model = model
        self._keys = list(param_shapes(cfg, model).keys())
        self._flat = torch.nn.ParameterList(
            [torch.nn.Parameter(v) for v in init_params(cfg, seed, model=model).values()])

    def named_oracle_params(self):
        return OrderedDict(zip(self._keys, self._flat))

    def forward(self, nodes, edges):
        return FORWARDS[self.model](self.named_oracle_params(), self.cfg, nodes, edges)
