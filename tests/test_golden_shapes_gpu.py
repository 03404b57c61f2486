"""-m gpu: the HIP models against outputs of the UNMODIFIED reference on the other BASELINE shapes and
variants (tests/golden/make_golden_shapes.py: GGNN on ZINC-shaped graphs, AttentionGGNN at the default
dimensions and on ChEMBL-shaped graphs, GGNN with four bond types) — the same anchor as
tests/test_model_gpu.py::test_golden_gdb13_default_dims, wider.  Logits at north_star's 1e-4 (rows of
fully-masked graphs: the reference's own fl32(e - 1e6) quantisation, 5e-3); gradients through the stored
digests (sum, |sum|, 32 samples per tensor): as a whole tightly, single tensors within the magnitude a
SELU-kink flip between two correct fp32 evaluations has (tests/test_oracle_golden.py shows
the reference's fp32 run and the oracle's differing by 3.7e-3 on one of these very fixtures)."""
import os

import numpy as np
import pytest
import torch

from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
from tests.golden.spec import digest
from tests.test_oracle_golden import SHAPE_GOLDENS, load_shape_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


@pytest.mark.parametrize("name", SHAPE_GOLDENS)
def test_hip_models_match_reference_outputs(golden_dir, name):
    cfg, P, g, kind = load_shape_golden(golden_dir, name)
    cls = mpnn.AttentionGGNN if kind == "AttGGNN" else mpnn.GGNN
    model = cls(O.as_constants(dict(cfg, device="cuda")))
    model.load_state_dict(P)
    model = model.to("cuda").train()
    nodes, edges, tgt = (torch.from_numpy(np.ascontiguousarray(g[k])).float().cuda()
                         for k in ("nodes", "edges", "apds"))
    out = model(nodes, edges)
    model.zero_grad()
    loss = O.kl_loss(out, tgt)
    loss.backward()
    out = out.detach().cpu()
    e8 = g["edges"]
    masked = np.nonzero(~e8.reshape(e8.shape[0], -1).any(1))[0]
    live = np.setdiff1d(np.arange(out.shape[0]), masked)
    assert rel(out[live], g["logits"][live]) < TOL
    if masked.size:
        assert rel(out[masked], g["logits"][masked]) < 5e-3
        from tests import pins
        pins.assert_masked_rows_are_quantum_ties(O, model, cfg, P, g["nodes"], g["edges"], out, kind)
    assert abs(float(loss) - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
    # ... and on the graphs that are not fully masked the reference's own logits give the same loss at 1e-4
    tl = lambda x: torch.as_tensor(np.asarray(x)).float()[torch.from_numpy(live)]
    ll = float(O.kl_loss(tl(g["logits"]), tl(g["apds"])))
    assert abs(float(O.kl_loss(tl(out), tl(g["apds"]))) - ll) < TOL * abs(ll)
    num = den = 0.0
    worst = (0.0, "")
    for k, p in model.named_parameters():
        d, ref = digest(p.grad.detach().cpu()), g["gdigest." + k]
        num += float(np.sum((d[2:] - ref[2:]) ** 2))
        den += float(np.sum(ref[2:] ** 2))
        scale = max(np.max(np.abs(ref[2:])), 1e-12)
        worst = max(worst, (float(np.max(np.abs(d[2:] - ref[2:])) / scale), k))
    print(f"\n[unpinned, {name}] gradient digests vs the reference's: global L2 {(num / den) ** 0.5:.2e}, worst tensor "
          f"{worst[0]:.2e} ({worst[1]})")
    assert (num / den) ** 0.5 < 5e-3, (num / den) ** 0.5
    assert worst[0] < 2e-2, worst
