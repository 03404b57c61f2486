"""CPU check of tests/pins.py: the mapping of SELU sign patterns from the HIP data layout into the
oracle's layout.  The fp32 dataflow model (tests/ref_dataflow.py) stands in for the HIP path; a wrong
row mapping would show up as a large number of "flipped" activations and as gradient mismatches."""
import numpy as np
import pytest
import torch

from graphinvent_amd import synthetic
from oracle import ggnn_oracle as O
from tests import pins, ref_dataflow as D
from tests.golden.spec import TINY, TINY_ATT, tiny_inputs


def _run(cfg, n8, e8, a8, model, seed=3):
    P = O.init_params(cfg, seed=seed, model=model)
    nodes, edges, tgt = (torch.from_numpy(x).float() for x in (n8, e8, a8))
    out, tape = D.forward(P, cfg, nodes, edges, keep=True, model=model)
    o = out.detach().clone().requires_grad_(True)
    O.kl_loss(o, tgt).backward()
    grads = D.backward(P, cfg, tape, o.grad)
    signs = pins.signs_from_dataflow(tape, model)
    o_ref, l_ref, g_ref, flipped, total = pins.oracle_pinned(O, P, cfg, nodes, edges, tgt, signs,
                                                             tape["g"], model)
    assert O.SELU_BRANCH_HOOK is None
    assert total > 0 and flipped <= max(2, int(1e-5 * total)), (flipped, total)
    assert float((out - o_ref).abs().max()) < 1e-4 * float(o_ref.abs().max())
    for k in g_ref:
        scale = max(float(g_ref[k].abs().max()), 1e-12)
        assert float((grads[k] - g_ref[k]).abs().max()) / scale < 1e-4, k
    return flipped, total


def _live(n8, e8, a8):
    keep = e8.reshape(e8.shape[0], -1).any(1)
    return n8[keep], e8[keep], a8[keep]


def test_pins_ggnn_tiny_with_pass0_rows():
    flipped, total = _run(O.make_config(**TINY), *_live(*tiny_inputs()), "GGNN")
    assert flipped == 0


def test_pins_attggnn_tiny():
    _run(O.make_config(**TINY_ATT), *_live(*tiny_inputs()), "AttGGNN")


@pytest.mark.parametrize("model", ["GGNN", "AttGGNN"])
def test_pins_gdb13_shape(model):
    cfg = O.make_config(hidden_node_features=24, message_size=20, enn_hidden_dim=16, enn_depth=2,
                        msg_hidden_dim=16, msg_depth=2, att_hidden_dim=12, att_depth=1,
                        gather_att_hidden_dim=16, gather_emb_hidden_dim=16, gather_width=12,
                        mlp1_hidden_dim=20, mlp2_hidden_dim=20, gather_att_depth=1,
                        gather_emb_depth=1, mlp1_depth=1, mlp2_depth=1)
    n8, e8, a8 = _live(*synthetic.make_batch(60, **synthetic.SHAPES["gdb13"], seed=2))
    _run(cfg, n8, e8, a8, model)


def test_a_wrong_mapping_is_detected():
    """Sanity of the detector itself: shuffled message rows must flip many activations."""
    cfg = O.make_config(**TINY)
    n8, e8, a8 = _live(*tiny_inputs())
    P = O.init_params(cfg, seed=3)
    nodes, edges, tgt = (torch.from_numpy(x).float() for x in (n8, e8, a8))
    _, tape = D.forward(P, cfg, nodes, edges, keep=True)
    signs = pins.signs_from_dataflow(tape)
    g = dict(tape["g"])
    g["in_perm"] = np.roll(g["in_perm"], 1)
    _, _, _, flipped, total = pins.oracle_pinned(O, P, cfg, nodes, edges, tgt, signs, g)
    assert flipped > 1e-3 * total
