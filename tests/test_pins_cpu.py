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
    """Sanity of the detectors themselves: shuffled message rows flip many activations (> 1e-3 of all), and — since
    round 4 — oracle_pinned itself refuses a pin that moves anything but a tie at the kink (|pre-activation| < 1e-5)."""
    cfg = O.make_config(**TINY)
    n8, e8, a8 = _live(*tiny_inputs())
    P = O.init_params(cfg, seed=3)
    nodes, edges, tgt = (torch.from_numpy(x).float() for x in (n8, e8, a8))
    _, tape = D.forward(P, cfg, nodes, edges, keep=True)
    signs = pins.signs_from_dataflow(tape)
    g = dict(tape["g"])
    g["in_perm"] = np.roll(g["in_perm"], 1)
    with pytest.raises(AssertionError):                       # the pin would move real activations across 0
        pins.oracle_pinned(O, P, cfg, nodes, edges, tgt, signs, g)
    old, pins.TIE_TOL = pins.TIE_TOL, float("inf")
    try:
        _, _, _, flipped, total = pins.oracle_pinned(O, P, cfg, nodes, edges, tgt, signs, g)
    finally:
        pins.TIE_TOL = old
    assert flipped > 1e-3 * total


# ---- fl32(e - 1e6) quanta of fully-masked graphs (tests/pins.MaskQuantumPin) ---------------------------
def _masked_graph_rows(tape, e8):
    """Compact rows whose energies only fully-masked graphs use unmasked: their own slots (+ the zero row)."""
    g = tape["g"]
    B = e8.shape[0]
    N = e8.shape[1]
    full = ~e8.reshape(B, -1).any(1)
    cidx = np.asarray(g["cidx"]).reshape(B, N)
    return full, np.unique(cidx[full])


@pytest.mark.parametrize("model", ["GGNN", "AttGGNN"])
def test_mask_quantum_pin_accounts_for_every_masked_row_difference(model, monkeypatch):
    """An implementation whose attention energies differ from the oracle's on the rows of fully-masked graphs
    (here: by 0.02, far more than rounding, to move many quanta) disagrees with the plain fp32 oracle on those
    graphs only — and agrees with the oracle at 1e-5 once the oracle is handed its fl32(e - 1e6) quanta."""
    cfg = O.make_config(**(TINY if model == "GGNN" else TINY_ATT))
    n8, e8, a8 = tiny_inputs()
    P = O.init_params(cfg, seed=5, model=model)
    nodes, edges = torch.from_numpy(n8).float(), torch.from_numpy(e8).float()
    _, tape0 = D.forward(P, cfg, nodes, edges, keep=True, model=model)
    full, rows = _masked_graph_rows(tape0, e8)
    assert full.sum() >= 2                                         # the empty graph and the single atom
    en0 = tape0["att_acts"][-1]
    delta = torch.zeros_like(en0)
    delta[torch.from_numpy(rows)] = 0.02 * torch.randn(len(rows), en0.shape[1],
                                                       generator=torch.Generator().manual_seed(1))
    orig = D.gather_readout
    monkeypatch.setattr(D, "gather_readout", lambda en, *a: orig(en + delta, *a))
    out_impl, tape = D.forward(P, cfg, nodes, edges, keep=True, model=model)
    monkeypatch.setattr(D, "gather_readout", orig)
    B, N = n8.shape[0], n8.shape[1]
    pin = pins.MaskQuantumPin(en0 + delta, tape["g"]["cidx"], B, N, cfg["big_positive"])
    o_plain = O.FORWARDS[model](P, cfg, nodes, edges)
    o_pin = pins.oracle_quantum_pinned_logits(O, P, cfg, nodes, edges, pin, model)
    assert O.MASK_QUANTUM_HOOK is None
    scale = float(o_plain.abs().max())
    live = torch.from_numpy(~full)
    assert float((out_impl - o_plain)[live].abs().max()) < 1e-5 * scale        # untouched graphs
    assert float((out_impl - o_plain)[~live].abs().max()) > 1e-4 * scale       # the test has teeth
    assert float((out_impl - o_pin).abs().max()) < 1e-5 * scale                # every difference = quanta
    assert pin.graphs == int(full.sum()) and pin.quanta == pin.graphs * N * en0.shape[1]
    assert 0 < pin.moved <= pin.quanta and pin.max_steps >= 1 and abs(pin.max_de - float(delta.abs().max())) < 1e-5


def test_mask_quantum_pin_is_a_no_op_on_consistent_energies_and_on_fp64():
    cfg = O.make_config(**TINY)
    n8, e8, a8 = tiny_inputs()
    P = O.init_params(cfg, seed=5)
    nodes, edges = torch.from_numpy(n8).float(), torch.from_numpy(e8).float()
    out, tape = D.forward(P, cfg, nodes, edges, keep=True)
    B, N = n8.shape[0], n8.shape[1]
    pin = pins.MaskQuantumPin(tape["att_acts"][-1], tape["g"]["cidx"], B, N, cfg["big_positive"])
    o_plain = O.ggnn_forward(P, cfg, nodes, edges)
    o_pin = pins.oracle_quantum_pinned_logits(O, P, cfg, nodes, edges, pin)
    assert pin.max_de < 1e-5 and pin.max_steps <= 1                # same arithmetic up to BLAS rounding
    assert pin.moved <= 1e-2 * pin.quanta
    if pin.moved == 0:
        assert torch.equal(o_pin, o_plain)
    P64 = {k: v.double() for k, v in P.items()}
    pin64 = pins.MaskQuantumPin(tape["att_acts"][-1], tape["g"]["cidx"], B, N, cfg["big_positive"])
    a = pins.oracle_quantum_pinned_logits(O, P64, cfg, nodes.double(), edges.double(), pin64)
    assert pin64.quanta == 0 and torch.equal(a, O.ggnn_forward(P64, cfg, nodes.double(), edges.double()))


def test_oracle_backward_operator_with_a_given_upstream_gradient():
    """`forward_backward(..., upstream=v)` returns J^T v: with v = d loss / d logits of its own evaluation it is the
    ordinary gradient, it is linear in v, and it is what tests/test_x2_trial_gpu.py compares the HIP backward with
    (the backward operator apart from the conditioning of softmax - target on a fitted model)."""
    cfg = O.make_config(**TINY)
    n8, e8, a8 = _live(*tiny_inputs())
    P = O.init_params(cfg, seed=5)
    nodes, edges, tgt = (torch.from_numpy(x).double() for x in (n8, e8, a8))
    P = {k: v.double() for k, v in P.items()}
    out, loss, g = O.forward_backward(P, cfg, nodes, edges, tgt)
    o = out.clone().requires_grad_(True)
    O.kl_loss(o, tgt).backward()
    _, _, gv = O.forward_backward(P, cfg, nodes, edges, tgt, upstream=o.grad)
    _, _, g2 = O.forward_backward(P, cfg, nodes, edges, tgt, upstream=-2.0 * o.grad)
    for k in g:
        scale = max(float(g[k].abs().max()), 1e-300)
        assert float((gv[k] - g[k]).abs().max()) < 1e-12 * scale, k
        assert float((g2[k] + 2.0 * g[k]).abs().max()) < 1e-12 * scale, k
