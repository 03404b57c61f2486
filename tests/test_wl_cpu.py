"""CPU proof for the next row reduction (tests/ref_wl.py): the colour-level (Weisfeiler-Lehman) dataflow of the
message passes gives the oracle's logits (fp64, so a wrong index map cannot hide behind rounding), and its index
arrays have the invariants a device implementation must reproduce.  Also records what it would save: on the
reference's shipped molecules the rows of passes 1+ collapse, on the uniform-random synthetic benchmark graphs
they hardly do."""
import os

import numpy as np
import torch

from graphinvent_amd import synthetic
from oracle import ggnn_oracle as O
from tests import ref_dataflow as D
from tests import ref_wl as W


def _small_cfg(sh=None, **over):
    base = dict(hidden_node_features=24, message_size=20, enn_hidden_dim=16, enn_depth=2, gather_att_hidden_dim=16,
                gather_emb_hidden_dim=16, gather_width=12, mlp1_hidden_dim=20, mlp2_hidden_dim=20, gather_att_depth=1,
                gather_emb_depth=1, mlp1_depth=1, mlp2_depth=1)
    base.update(over)
    if sh is None:
        return O.make_config(**base)
    return O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"], **base)


def _check(cfg, n8, e8, seed):
    P = O.init_params(cfg, seed=seed, dtype=torch.float64)
    nodes, edges = torch.from_numpy(n8).double(), torch.from_numpy(e8).double()
    ref = O.ggnn_forward(P, cfg, nodes, edges)
    out, stats = W.forward(P, cfg, nodes, edges)
    assert float((out - ref).abs().max()) < 1e-9 * max(float(ref.abs().max()), 1.0)
    return stats


def test_colour_level_dataflow_equals_the_oracle_on_the_shipped_molecules(golden_dir):
    d = np.load(os.path.join(golden_dir, "gdb13_1K-debug_valid.npz"))
    n8, e8 = d["nodes"], d["edges"]
    cfg = _small_cfg(n_node_features=n8.shape[2], n_edge_features=e8.shape[3], max_n_nodes=n8.shape[1],
                     len_f_add_per_node=(d["APDs"].shape[1] - 1 - n8.shape[1] * e8.shape[3]) // n8.shape[1],
                     len_f_conn_per_node=e8.shape[3])
    stats = _check(cfg, n8, e8, seed=5)
    rows, mrows = stats[0]["rows"], stats[0]["message_rows"]
    # real molecules repeat their local environments: passes 1 and 2 need a fraction of the rows
    assert stats[0]["colours_in"] < 0.05 * rows                       # pass 0: the feature classes (what ships today)
    assert stats[1]["colours_in"] < 0.2 * rows and stats[1]["message_pairs"] < 0.25 * mrows
    assert stats[2]["colours_in"] < 0.6 * rows
    print("shipped fixture: rows", rows, "message rows", mrows, [(s["colours_in"], s["message_pairs"]) for s in stats])


def test_colour_level_dataflow_on_synthetic_batches_and_edge_cases():
    sh = synthetic.SHAPES["gdb13"]
    cfg = _small_cfg(sh)
    n8, e8, _ = synthetic.make_batch(60, **sh, seed=3)               # incl. empty and single-atom graphs
    stats = _check(cfg, n8, e8, seed=2)
    rows = stats[0]["rows"]
    assert stats[-1]["colours_out"] <= rows and stats[0]["colours_in"] <= 64
    # an edgeless batch and a batch of identical graphs
    n0, e0, _ = synthetic.make_batch(6, **sh, seed=4)
    e0[:] = 0
    _check(cfg, n0, e0, seed=1)
    n1 = np.repeat(n8[7:8], 9, axis=0); e1 = np.repeat(e8[7:8], 9, axis=0)
    st = _check(cfg, n1, e1, seed=1)
    assert st[-1]["colours_out"] <= n8.shape[1] + 1                   # nine copies cost what one graph costs


def test_plan_invariants():
    sh = synthetic.SHAPES["gdb13"]
    n8, e8, _ = synthetic.make_batch(40, **sh, seed=9)
    g = D.compact(n8, e8)
    S, R = g["S"], g["S"] + 1
    x = np.zeros((R, n8.shape[2]), dtype=np.int64)
    x[:S] = n8.reshape(-1, n8.shape[2])[g["slot_of"]]
    plans = W.wl_plan(g, x, 3)
    for p, pl in enumerate(plans):
        assert pl["cls"].shape == (R,) and pl["cls"].max() + 1 == pl["ncls"]
        assert (pl["cls"][pl["rep"]] == np.arange(pl["ncls"])).all()                  # representatives carry their colour
        assert (np.diff(pl["rep"]) > 0).all()                                          # ids in order of first appearance
        assert pl["cnt"].shape == (pl["ncls_next"], len(pl["mpairs"]))
        # refinement: rows of one next colour share their previous colour; the partition only gets finer
        for k in range(pl["ncls_next"]):
            assert len(set(pl["cls"][pl["nxt"] == k].tolist())) == 1
        assert pl["ncls_next"] >= pl["ncls"]
        # the count matrix of a colour is the in-degree of its members
        deg = g["seg_off"][1:R + 1] - g["seg_off"][:R]
        assert (pl["cnt"].sum(1) == deg[pl["rep_next"]]).all()
        if p + 1 < len(plans):
            assert (plans[p + 1]["cls"] == pl["nxt"]).all()
