"""Pins the oracle (oracle/ggnn_oracle.py) against outputs of the unmodified reference that
tests/golden/make_golden.py recorded in the build container.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import ggnn_oracle as O
from tests.golden.spec import TINY, digest


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def test_state_dict_contract():
    """Keys, order and shapes of the checkpoint wire format."""
    shapes = O.param_shapes(O.make_config())
    keys = list(shapes)
    assert keys[0] == "msg_nns.0.seq.0.weight" and shapes[keys[0]] == (250, 100)
    assert shapes["msg_nns.2.seq.12.weight"] == (100, 250)
    assert shapes["gru.weight_ih"] == (300, 100) and shapes["gru.bias_hh"] == (300,)
    assert shapes["gather.att_nn.seq.0.weight"] == (250, 108)
    assert shapes["APDReadout.fAddNet2.seq.0.weight"] == (500, 685)
    assert shapes["APDReadout.fAddNet2.seq.12.weight"] == (585, 500)
    assert shapes["APDReadout.fTermNet2.seq.12.weight"] == (1, 500)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 5914773
    assert O.apd_width(O.make_config()) == 625


def test_tiny_forward_backward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "golden_tiny.npz"))
    cfg = O.make_config(**TINY)
    P = O.init_params(cfg, seed=11)
    for k, v in P.items():                         # weights are machine independent
        assert np.array_equal(v.numpy(), g["param." + k]), k
    nodes, edges, tgt = (torch.from_numpy(g[k]).float() for k in ("nodes", "edges", "apds"))
    out, loss, grads = O.forward_backward(P, cfg, nodes, edges, tgt)
    assert rel(out.numpy(), g["logits"]) < 2e-6
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    for k, v in grads.items():
        assert rel(v.numpy(), g["grad." + k]) < 2e-5, k


def test_tiny_fp64_agrees_with_reference_fp32(golden_dir):
    """fp64 run of the oracle = tie-breaker; fully-masked graphs (rows 1,2: empty / single atom)
    carry the fl32(e - 1e6) quantisation and are checked with a looser bound."""
    g = np.load(os.path.join(golden_dir, "golden_tiny.npz"))
    cfg = O.make_config(**TINY)
    P = O.init_params(cfg, seed=11, dtype=torch.float64)
    nodes, edges = (torch.from_numpy(g[k]).double() for k in ("nodes", "edges"))
    out = O.ggnn_forward(P, cfg, nodes, edges).numpy()
    masked = np.array([1, 2])
    rest = np.setdiff1d(np.arange(out.shape[0]), masked)
    assert rel(out[rest], g["logits"][rest]) < 1e-5
    assert rel(out[masked], g["logits"][masked]) < 2e-2


def test_gdb13_default_dims_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "golden_gdb13.npz"))
    cfg = O.make_config()
    P = O.init_params(cfg, seed=int(g["seed"]))
    nodes, edges, tgt = (torch.from_numpy(g[k]).float() for k in ("nodes", "edges", "apds"))
    out, loss, grads = O.forward_backward(P, cfg, nodes, edges, tgt)
    assert rel(out.numpy(), g["logits"]) < 5e-6
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    for k, v in grads.items():
        d, ref = digest(v), g["gdigest." + k]
        scale = max(np.max(np.abs(ref[2:])), 1e-12)
        assert np.max(np.abs(d[2:] - ref[2:])) / scale < 1e-4, k
        assert abs(d[1] - ref[1]) <= 1e-4 * ref[1] + 1e-12, k


def test_fixture_quirks(golden_dir):
    """train rows 129-149 are all-zero padding -> NaN reference loss; valid/test
    have none; empty and single-atom graphs are present."""
    tr = np.load(os.path.join(golden_dir, "gdb13_1K-debug_train.npz"))
    assert tr["nodes"].shape == (150, 13, 8) and tr["edges"].shape == (150, 13, 13, 3)
    assert tr["APDs"].shape == (150, 625)
    assert not tr["APDs"][129:].any() and tr["APDs"][:129].sum(1).min() > 0
    n_nodes = tr["nodes"][:129].any(2).sum(1)
    assert (n_nodes == 0).sum() == 3
    no_edges = ~tr["edges"][:129].reshape(129, -1).any(1)
    assert ((n_nodes == 1) & no_edges).sum() == 4
    for split in ("valid", "test"):
        d = np.load(os.path.join(golden_dir, f"gdb13_1K-debug_{split}.npz"))
        assert d["APDs"].sum(1).min() > 0
    cfg = O.make_config()
    P = O.init_params(cfg, seed=1)
    nodes, edges, tgt = (torch.from_numpy(tr[k][120:150]).float() for k in ("nodes", "edges", "APDs"))
    assert torch.isnan(O.kl_loss(O.ggnn_forward(P, cfg, nodes, edges), tgt))


def test_attggnn_oracle_matches_reference(golden_dir):
    """AttentionGGNN (BASELINE config 5's model): oracle restatement vs the reference's outputs."""
    from tests.golden.spec import TINY_ATT
    g = np.load(os.path.join(golden_dir, "golden_att_tiny.npz"))
    cfg = O.make_config(**TINY_ATT)
    P = O.init_params(cfg, seed=13, model="AttGGNN")
    assert [k[6:] for k in g.files if k.startswith("param.")] == list(P)      # registration order
    for k, v in P.items():
        assert np.array_equal(v.numpy(), g["param." + k]), k
    nodes, edges, tgt = (torch.from_numpy(g[k]).float() for k in ("nodes", "edges", "apds"))
    out, loss, grads = O.forward_backward(P, cfg, nodes, edges, tgt, model="AttGGNN")
    assert rel(out.numpy(), g["logits"]) < 2e-6
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    for k, v in grads.items():
        assert rel(v.numpy(), g["grad." + k]) < 2e-5, k


SHAPE_GOLDENS = ("golden_zinc", "golden_att_gdb13", "golden_att_chembl", "golden_aromatic")


def load_shape_golden(golden_dir, name):
    """(cfg, params, arrays, model kind) of a fixture written by tests/golden/make_golden_shapes.py."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = O.make_config(**{k[4:]: int(g[k]) for k in g.files if k.startswith("cfg.")})
    kind = str(g["model"])
    return cfg, O.init_params(cfg, seed=int(g["seed"]), model=kind), g, kind


@pytest.mark.parametrize("name", SHAPE_GOLDENS)
def test_other_shapes_and_variants_match_reference(golden_dir, name):
    """The oracle against outputs of the unmodified reference on the other BASELINE shapes (ZINC, ChEMBL),
    AttentionGGNN at the default dimensions and the four-bond-type (aromatic) preprocessing variant.
    Logits and loss in fp32.  Gradient digests: two correct fp32 evaluations can differ by single SELU
    outputs landing on opposite sides of 0 (golden_zinc: 3.7e-3 on the fConnNet1 tensors between the
    reference's fp32 run and the oracle's, while the oracle's fp64 run agrees with the reference to 7e-6),
    and an fp64 evaluation differs from any fp32 one on fully-masked graphs (the `- 1e6` mask quantises
    their energies in fp32; golden_att_gdb13 contains such rows: 1.2e-2 against fp64, 6e-6 against fp32) —
    so every tensor must agree with the oracle at 1e-4 in at least one of the two precisions."""
    cfg, P, g, kind = load_shape_golden(golden_dir, name)
    nodes, edges, tgt = (torch.from_numpy(g[k]).float() for k in ("nodes", "edges", "apds"))
    out, loss, grads = O.forward_backward(P, cfg, nodes, edges, tgt, model=kind)
    assert out.shape == g["logits"].shape
    assert rel(out.numpy(), g["logits"]) < 5e-6
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    assert set("gdigest." + k for k in grads) == set(k for k in g.files if k.startswith("gdigest."))
    P64 = {k: v.double() for k, v in P.items()}
    _, _, grads64 = O.forward_backward(P64, cfg, nodes.double(), edges.double(), tgt.double(), model=kind)

    def err(v, ref):
        d = digest(v)
        scale = max(np.max(np.abs(ref[2:])), 1e-12)
        return max(np.max(np.abs(d[2:] - ref[2:])) / scale, abs(d[1] - ref[1]) / max(ref[1], 1e-12))
    for k in grads:
        ref = g["gdigest." + k]
        assert min(err(grads[k], ref), err(grads64[k], ref)) < 1e-4, (k, err(grads[k], ref), err(grads64[k], ref))
