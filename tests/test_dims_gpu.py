"""-m gpu: model-level parity at dimensions OUTSIDE the tuned shapes (round-5 verdict, missing #2).

Every other model test uses H = M in {24, 64, 100, 128}, stack widths <= 250 / 500 and A <= 108.  The reference
accepts any of these (parameters/defaults.py:280-300 are job-file defaults; the action space follows the job's
atom / charge / implicit-H / chirality lists, parameters/constants.py:23-35,56-89,184), and each of the cases below
takes a code path of csrc/gi_model.hip that the tuned shapes never reach:

  wide_h      H = M = 256: the chain kernels at their width limit (GI_CHAIN_MAXW), the stacks' FIRST layers wide
              enough for the 16-bit pipe (no amax producer for h -> bf16x3 forward inside an fp16x2 model)
  implicit_h  A = 12 atom types x 3 charges x 4 implicit-H counts x 3 bond types = 432, 19 node features: the
              node-level stacks' LAST layer (500 -> 432) is wide (no amax producer for its dZ -> bf16x3 dgrad and
              weight gradient), the graph-level stacks reduce over K = 13 * 432 + 100 = 5 716 (split-K slabs)
  wide_enn    enn_hidden_dim = 300 > GI_CHAIN_MAXW: the message stacks are chain-INELIGIBLE and run layer by layer
              (grouped GEMM launches, forward and dgrad)
  ggnn_n88    GGNN (not AttentionGGNN) on ChEMBL-shaped graphs, N = 88

Protocol = tests/test_model_gpu.py::test_bench_batch_gradients_1e4_vs_fp32_oracle_autograd: logits, loss and every
gradient tensor at 1e-4 (max |d| / max |ref| per tensor) against the fp32 oracle's own forward + autograd with the
SELU branches and the masked graphs' energy quanta pinned to the HIP forward's (ties only: tests/pins.py asserts it),
live rows also against the PLAIN oracle — in the three arithmetic modes (fp16x2, bf16x3 only, fp32 MFMA only), and
`gi_prof_pipes` must show which matrix pipe each mode's GEMM launches actually took.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from graphinvent_amd import lib as L, synthetic
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
from tests import pins
from tests.test_model_gpu import assert_parity_with_both_pins, fully_masked_rows, make_model, to_dev

pytestmark = pytest.mark.gpu

MODES = ("fp16x2", "bf16x3", "fp32")


def _implicit_h_batch(B, seed):
    """GDB-13-sized graphs whose node rows carry a third one-hot block (implicit-H count, constants.py:23-35) and
    whose APD rows have the matching width (12 x 3 x 4 x 3 actions per node)."""
    n8, e8, _ = synthetic.make_batch(B, max_n_nodes=13, n_atom_types=12, n_formal_charge=3, seed=seed)
    rng = np.random.default_rng(seed + 1)
    nh = np.zeros(n8.shape[:2] + (4,), np.int8)
    occupied = n8.any(-1)
    h = rng.integers(0, 4, size=occupied.shape)
    for k in range(4):
        nh[..., k] = (occupied & (h == k)).astype(np.int8)
    n8 = np.concatenate([n8, nh], -1)
    width = 13 * 432 + 13 * 3 + 1
    a8 = np.zeros((B, width), np.int8)
    for b in range(B):
        np.add.at(a8[b], rng.integers(0, width, size=int(rng.integers(1, 5))), 1)
    return n8, e8, a8


def _case(name):
    """(config, batch) of a case; batches are big enough (>= 2 560 node rows, BF3_MIN_ROWS) for the 16-bit-pipe
    launch classes to be taken where the case has wide layers."""
    if name == "wide_h":
        sh = synthetic.SHAPES["gdb13"]
        cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"],
                              hidden_node_features=256, message_size=256)
        return cfg, synthetic.make_batch(420, **sh, seed=21)
    if name == "implicit_h":
        cfg = O.make_config(n_node_features=19, n_edge_features=3, max_n_nodes=13, len_f_add_per_node=432,
                            len_f_conn_per_node=3)
        return cfg, _implicit_h_batch(420, seed=22)
    if name == "wide_enn":
        sh = synthetic.SHAPES["gdb13"]
        cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"], enn_hidden_dim=300)
        return cfg, synthetic.make_batch(420, **sh, seed=23)
    if name == "ggnn_n88":
        sh = synthetic.SHAPES["chembl"]
        cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"])
        return cfg, synthetic.make_batch(64, **sh, seed=24)
    raise KeyError(name)


def _set_mode(lib, mode):
    lib.gi_bf3_enable(0 if mode == "fp32" else 1)
    lib.gi_x2_enable(1 if mode == "fp16x2" else 0)


def _pipes(lib):
    ms = (C.c_double * 3)(); work = (C.c_double * 3)(); n = (C.c_int * 3)()
    L.check(lib.gi_prof_pipes(ms, work, n), "gi_prof_pipes")
    return {"fp32": n[0], "bf16x3": n[1], "fp16x2": n[2]}


#: which pipes a case's GEMM-family launches must / must not have taken, per mode (launch counts of gi_prof_pipes)
def _check_pipes(name, mode, n):
    assert n["fp32"] > 0, (name, mode, n)                       # narrow layers, GRU projections: always fp32 MFMA
    if mode == "fp32":
        assert n["bf16x3"] == 0 and n["fp16x2"] == 0, (name, mode, n)
    elif mode == "bf16x3":
        assert n["fp16x2"] == 0 and n["bf16x3"] > 0, (name, mode, n)
    else:
        assert n["fp16x2"] > 0, (name, mode, n)


@pytest.mark.parametrize("name", ["wide_h", "implicit_h", "wide_enn", "ggnn_n88"])
def test_model_parity_outside_the_tuned_dimensions(name):
    cfg, (n8, e8, a8) = _case(name)
    assert len(fully_masked_rows(e8)) >= 1
    P = O.init_params(cfg, seed=31)
    lib = L.load()
    was = lib.gi_bf3_enable(-1), lib.gi_x2_enable(-1)
    old_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    report = {}
    try:
        for mode in MODES:
            _set_mode(lib, mode)
            model = make_model(cfg, P)
            params = list(model.parameters())
            nodes, edges, tgt = to_dev(n8, e8, a8)
            torch.cuda.synchronize()
            lib.gi_prof_enable(1)
            out, tape = mpnn.ggnn_forward_raw(model.constants, nodes, edges, params)
            dims, graph, ws = tape
            # (read back BEFORE the backward, which forms the last layers' dZ in place over their outputs)
            signs = pins.signs_from_hip(dims, graph, ws, out, attn=False)
            mask_pin = pins.mask_pin_from_hip(dims, graph, ws, n8.shape[0], cfg["big_positive"])
            g = pins.graph_arrays(graph)
            o_leaf = out.detach().clone().requires_grad_(True)
            loss = O.kl_loss(o_leaf, tgt)
            loss.backward()
            grads, _ = mpnn.ggnn_backward_raw(tape, out, o_leaf.grad, params)
            torch.cuda.synchronize()
            ms = (C.c_double * 2)(); busy = (C.c_double * 2)(); work = (C.c_double * 2)(); nl = (C.c_int * 2)()
            L.check(lib.gi_prof_collect(ms, busy, work, nl), "gi_prof_collect")
            lib.gi_prof_enable(0)
            report[mode] = _pipes(lib)
            _check_pipes(name, mode, report[mode])
            names = [k for k, _ in model.named_parameters()]
            assert_parity_with_both_pins(O, P, cfg, "GGNN", n8, e8, a8, out, loss, names, grads, signs, g, mask_pin)
    finally:
        lib.gi_prof_enable(0)
        lib.gi_bf3_enable(was[0]); lib.gi_x2_enable(was[1])
        torch.set_num_threads(old_threads)
    print(f"\n[{name}] GEMM-family launches per matrix pipe and mode: {report}")
