"""CPU proof that the MI355X dataflow (compact rows + shared zero row + routed message MLPs +
segmented sums + hand-written backward, tests/ref_dataflow.py) equals the reference algorithm
(oracle + autograd).  fp64 so formula errors cannot hide behind rounding."""
import os

import numpy as np
import pytest
import torch

from oracle import ggnn_oracle as O
from tests import ref_dataflow as D
from tests.golden.spec import TINY, TINY_ATT, tiny_inputs
from graphinvent_amd import synthetic


def _check(cfg, n8, e8, a8, seed, tol=1e-9, model="GGNN"):
    P = O.init_params(cfg, seed=seed, dtype=torch.float64, model=model)
    nodes, edges, tgt = (torch.from_numpy(x).double() for x in (n8, e8, a8))
    out_ref, loss_ref, g_ref = O.forward_backward(P, cfg, nodes, edges, tgt, model=model)
    out, tape = D.forward(P, cfg, nodes, edges, keep=True, model=model)
    assert (out - out_ref).abs().max() < tol * max(1.0, out_ref.abs().max())
    o = out.detach().clone().requires_grad_(True)
    O.kl_loss(o, tgt).backward()
    grads = D.backward(P, cfg, tape, o.grad)
    assert set(grads) == set(g_ref)
    for k in g_ref:
        scale = max(float(g_ref[k].abs().max()), 1e-12)
        assert float((grads[k] - g_ref[k]).abs().max()) / scale < 1e-7, k


def test_tiny_edge_cases_fp64():
    _check(O.make_config(**TINY), *tiny_inputs(), seed=11)


def test_asymmetric_adjacency_and_inactive_neighbour_fp64():
    """Inputs outside the data contract the reference still accepts: a directed edge whose
    neighbour slot has an all-zero feature row and no edges of its own."""
    cfg = O.make_config(**TINY)
    n8, e8, a8 = tiny_inputs()
    n8[4] = 0; e8[4] = 0
    n8[4, 0, 0] = 1; n8[4, 0, 3] = 1
    e8[4, 0, 5, 1] = 1                      # node 0 receives from empty slot 5, nothing back
    _check(cfg, n8, e8, a8, seed=5)


def test_gdb13_shape_small_hidden_fp64():
    cfg = O.make_config(hidden_node_features=24, message_size=20, enn_hidden_dim=16,
                        gather_att_hidden_dim=16, gather_emb_hidden_dim=16, gather_width=12,
                        mlp1_hidden_dim=20, mlp2_hidden_dim=20, enn_depth=1, gather_att_depth=1,
                        gather_emb_depth=1, mlp1_depth=1, mlp2_depth=1)
    n8, e8, a8 = synthetic.make_batch(40, **synthetic.SHAPES["gdb13"], seed=2)
    _check(cfg, n8, e8, a8, seed=3)


def test_attggnn_tiny_edge_cases_fp64():
    """AttentionGGNN: edge-list segment softmax == the reference's neighbour-padded softmax."""
    _check(O.make_config(**TINY_ATT), *tiny_inputs(), seed=11, model="AttGGNN")


def test_attggnn_gdb13_shape_small_hidden_fp64():
    cfg = O.make_config(hidden_node_features=24, message_size=20, msg_hidden_dim=16, msg_depth=2,
                        att_hidden_dim=12, att_depth=1,
                        gather_att_hidden_dim=16, gather_emb_hidden_dim=16, gather_width=12,
                        mlp1_hidden_dim=20, mlp2_hidden_dim=20, gather_att_depth=1,
                        gather_emb_depth=1, mlp1_depth=1, mlp2_depth=1)
    n8, e8, a8 = synthetic.make_batch(40, **synthetic.SHAPES["gdb13"], seed=2)
    _check(cfg, n8, e8, a8, seed=3, model="AttGGNN")


def test_compact_invariants(golden_dir):
    d = np.load(os.path.join(golden_dir, "gdb13_1K-debug_valid.npz"))
    nodes, edges = d["nodes"], d["edges"]
    g = D.compact(nodes, edges)
    S, E, U = g["S"], g["E"], g["U"]
    B, N = nodes.shape[:2]
    assert g["err"] == 0 and E == int((edges.sum(3) != 0).sum())
    assert S == int(((nodes != 0).any(2)).sum())            # well-formed data: active == occupied
    # message rows = distinct (source slot, bond type) pairs, bond-type-major
    pairs = {(b, j, int(edges[b, i, j].argmax())) for b, i, j in zip(*np.nonzero(edges.sum(3)))}
    assert U == len(pairs) <= E and g["type_off"][-1] == U
    assert sorted(g["out_perm"].tolist()) == list(range(U))
    assert sorted(g["mu_slot"].tolist()) == list(range(E))
    assert g["seg_off"][S] == E == g["seg_off"][S + 1] and g["src_off"][S + 1] == U == g["src_off"][S]
    assert g["mu_off"][0] == 0 and g["mu_off"][U] == E and np.all(np.diff(g["mu_off"]) >= 1)
    assert np.all(np.diff(g["seg_off"]) >= 0)
    # dst-CSR: slot k of destination c carries the message row of (its k-th neighbour, bond type)
    eb, ei, ej = np.nonzero(edges.sum(3))
    et = edges[eb, ei, ej].argmax(1)
    for k in range(E):
        u = g["in_perm"][k]
        t = int(np.searchsorted(g["type_off"], u, side="right") - 1)
        assert t == et[k] and g["u_src"][u] == g["cidx"][eb[k] * N + ej[k]]
        assert g["seg_off"][g["cidx"][eb[k] * N + ei[k]]] <= k < g["seg_off"][g["cidx"][eb[k] * N + ei[k]] + 1]
    # message CSR: row u lists exactly the dst-CSR slots that read it, with their destinations
    for u in range(U):
        slots = g["mu_slot"][g["mu_off"][u]:g["mu_off"][u + 1]]
        assert np.all(g["in_perm"][slots] == u) and np.all(np.diff(slots) > 0)
        assert np.all(g["mu_dst"][g["mu_off"][u]:g["mu_off"][u + 1]] == g["cidx"][eb[slots] * N + ei[slots]])
    # source CSR over message rows
    for c in range(S):
        rows = g["out_perm"][g["src_off"][c]:g["src_off"][c + 1]]
        assert np.all(g["u_src"][rows] == c) and np.all(np.diff(rows) > 0)
    multi = edges[:4].copy()
    multi[0, 0, 1, :] = [1, 1, 0]                       # two bond types on one pair: flagged (bit 3), parallel edges
    assert D.compact(nodes[:4], multi)["err"] == 8
    bad = edges[:4].copy()
    bad[0, 0, 1, 0] = 2                                 # an entry that is not 0 / 1: outside the contract (bit 0)
    assert D.compact(nodes[:4], bad)["err"] & 1


def test_pairs_with_several_bond_types_are_parallel_edges_like_the_reference():
    """The generation loop's dummy graph 0 accumulates bonds of several types on one atom pair
    (GraphGenerator.py:133, 424-427: it samples and applies actions for ever and is never reset); the reference then sums
    the per-type messages of the pair (gnn/mpnn.py:286-294).  The compact dataflow treats every set bond-type entry as
    an edge of its own: forward and backward equal the oracle's in fp64, self-loops and all-ones feature rows included."""
    cfg = O.make_config(**TINY)
    n8, e8, a8 = tiny_inputs()
    N = n8.shape[1]
    n8[0] = 1                                                # the dummy graph: all-ones feature rows,
    e8[0] = 0
    e8[0, 0, 0, 0] = 1                                       # a self-loop,
    e8[0, 0, 3, :] = [1, 0, 1]; e8[0, 3, 0, :] = [1, 0, 1]   # two bond types on one pair,
    e8[0, 2, 0, :] = 1                                       # all three, one direction only
    b = int(np.argmax(e8.reshape(e8.shape[0], -1).sum(1) * (np.arange(e8.shape[0]) > 0)))
    i, j = [int(x[0]) for x in np.nonzero(e8[b].sum(2))]     # (and in another graph, on top of an existing bond)
    e8[b, i, j, :] = 1; e8[b, j, i, :] = 1
    g = D.compact(n8, e8)
    assert g["err"] == 8 and g["E"] == int((e8 == 1).sum())
    _check(cfg, n8, e8, a8, seed=7)
