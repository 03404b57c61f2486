"""-m gpu: AlphaDropout in training mode (gnn/modules.py:130-142 with dropout_p > 0).

The reference draws an independent mask per MLP layer output element, over every edge row and every
padded node slot.  The HIP path (no row sharing in this mode, masks from a counter-based hash) is
checked three ways:
  * the elementwise kernel against ATen's AlphaDropout arithmetic on the same mask: bit-exact;
  * the whole model against the ORACLE's fp32 forward/backward fed the exported masks
    (tests/dropout_masks.py): logits / loss / every gradient at the usual tolerances;
  * statistics (keep rate, independence across seeds / sites) and eval() == the p = 0 model.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from graphinvent_amd import lib as L
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
from tests import ref_dataflow as D
from tests.dropout_masks import OracleDropout, export_mask
from tests.golden.spec import TINY, TINY_ATT, tiny_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def test_alpha_dropout_kernel_is_atens_arithmetic_on_the_exported_mask():
    lib = L.load()
    torch.manual_seed(0)
    rows, cols, ld, p, seed, site = 777, 250, 252, 0.15, 0x1234567890ABCDEF, 37
    s = torch.selu(torch.randn(rows, ld) * 2)
    q = L.DropoutParams()
    L.check(lib.gi_dropout_setup(p, seed, site, C.byref(q)), "setup")
    buf = torch.zeros(2, rows, ld)
    buf[0] = s
    dev = buf.to(DEV)
    L.check(lib.gi_alpha_dropout_fwd(dev.data_ptr(), ld, rows, cols, rows * ld, C.byref(q),
                                     torch.cuda.current_stream().cuda_stream), "dropout")
    got = dev.cpu()
    keep = export_mask(seed, site, p, rows, cols)
    O.DROPOUT_HOOK = lambda prefix, layer, x: (p, keep)
    try:
        want = O._alpha_dropout(s[:, :cols], "x", 0)
    finally:
        O.DROPOUT_HOOK = None
    assert torch.equal(got[0, :, :cols], want)                      # bit-exact
    assert torch.equal(got[0, :, cols:], s[:, cols:])               # padding columns untouched
    # the stored backward factor: keep * a * selu'(z), through the SELU output
    y = s[:, :cols]
    dselu = torch.where(y > 0, torch.full_like(y, O.SELU_SCALE), y + O.SELU_SCALE * O.SELU_ALPHA)
    a = 1.0 / np.sqrt((O.ALPHA_DROPOUT_ALPHA ** 2 * p + 1) * (1 - p))
    assert rel(got[1, :, :cols], keep.float() * a * dselu) < 1e-6
    rate = keep.float().mean().item()
    assert abs(rate - (1 - p)) < 4 * np.sqrt(p * (1 - p) / keep.numel())
    other = export_mask(seed, site + 1, p, rows, cols)
    assert abs((keep == other).float().mean().item() - ((1 - p) ** 2 + p ** 2)) < 0.01   # independent
    assert not torch.equal(keep, export_mask(seed + 1, site, p, rows, cols))


def _model(cls, cfg, P):
    model = cls(O.as_constants(dict(cfg, device="cuda")))
    model.load_state_dict(P)
    return model.to(DEV)


def _live_inputs():
    n8, e8, a8 = tiny_inputs()
    live = np.nonzero(e8.reshape(e8.shape[0], -1).any(1))[0]       # drop the fully-masked graphs
    return n8[live], e8[live], a8[live]


DROPS = dict(enn_dropout_p=0.1, gather_att_dropout_p=0.15, gather_emb_dropout_p=0.05,
             mlp1_dropout_p=0.2, mlp2_dropout_p=0.1)


@pytest.mark.parametrize("name", ["GGNN", "AttGGNN"])
def test_training_mode_matches_the_oracle_fed_the_same_masks(name):
    if name == "GGNN":
        cfg, cls = O.make_config(**TINY, **DROPS), mpnn.GGNN
    else:
        drops = dict(DROPS, msg_dropout_p=0.1, att_dropout_p=0.2)
        drops.pop("enn_dropout_p")
        cfg, cls = O.make_config(**TINY_ATT, **drops), mpnn.AttentionGGNN
    P = O.init_params(cfg, seed=5, model=name)
    n8, e8, a8 = _live_inputs()
    model = _model(cls, cfg, P).train()
    model.dropout_seed = 20250925
    nodes, edges, tgt = (torch.from_numpy(x).float().to(DEV) for x in (n8, e8, a8))
    out = model(nodes, edges)
    assert out.shape == (n8.shape[0], O.apd_width(cfg))
    loss = O.kl_loss(out, tgt)
    model.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    # the oracle's own fp32 forward / backward with the masks the HIP path drew
    g = D.compact(n8, e8, nodedup=True)
    hook = OracleDropout(cfg, list(P.keys()), model.last_dropout_seed, g, n8, e8, name)
    O.DROPOUT_HOOK = hook
    try:
        t = lambda x: torch.from_numpy(x).float()
        o32, l32, g32 = O.forward_backward(P, cfg, t(n8), t(e8), t(a8), name)
    finally:
        O.DROPOUT_HOOK = None
    assert hook.sites > 10 and 0.7 < hook.kept / hook.drawn < 0.95     # masks really applied
    assert rel(out, o32) < TOL
    assert abs(float(loss.detach()) - float(l32)) < TOL * abs(float(l32))
    worst = max((rel(grads[k], g32[k]), k) for k in grads)
    assert worst[0] < 2e-3, worst
    num = sum(float(((grads[k].double() - g32[k].double()) ** 2).sum()) for k in grads)
    den = sum(float((g32[k].double() ** 2).sum()) for k in grads)
    assert (num / den) ** 0.5 < 2e-4
    # the dropout changes the result (it is not silently off), and the seed decides it
    model.dropout_seed = 7
    out2 = model(nodes, edges)
    assert rel(out2, out) > 1e-2
    model.dropout_seed = 20250925
    assert torch.equal(model(nodes, edges), out)                       # same seed: bit-identical


def test_eval_mode_ignores_dropout_and_default_seed_follows_torch_manual_seed():
    cfg_p = O.make_config(**TINY, **DROPS)
    cfg_0 = O.make_config(**TINY)
    P = O.init_params(cfg_0, seed=3)
    n8, e8, _ = _live_inputs()
    nodes, edges = (torch.from_numpy(x).float().to(DEV) for x in (n8, e8))
    with_p, without = _model(mpnn.GGNN, cfg_p, P).eval(), _model(mpnn.GGNN, cfg_0, P).eval()
    with torch.no_grad():
        assert torch.equal(with_p(nodes, edges), without(nodes, edges))
    with_p.train()
    torch.manual_seed(123)
    a = with_p(nodes, edges).detach().clone()
    seed_a = with_p.last_dropout_seed
    b = with_p(nodes, edges).detach().clone()
    assert with_p.last_dropout_seed != seed_a and not torch.equal(a, b)
    torch.manual_seed(123)
    assert torch.equal(with_p(nodes, edges).detach(), a)


def test_training_mode_at_default_dims_statistics():
    """GDB-13 default dims, B = 64: finite outputs / gradients, logits spread like the p = 0 model."""
    from graphinvent_amd import synthetic
    cfg = O.make_config(**DROPS)
    P = O.init_params(cfg, seed=2)
    n8, e8, a8 = synthetic.make_batch(64, **synthetic.SHAPES["gdb13"], seed=9)
    nodes, edges, tgt = (torch.from_numpy(x).float().to(DEV) for x in (n8, e8, a8))
    model = _model(mpnn.GGNN, cfg, P).train()
    out = model(nodes, edges)
    loss = O.kl_loss(out, tgt)
    loss.backward()
    assert torch.isfinite(out).all() and torch.isfinite(loss)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    base = _model(mpnn.GGNN, O.make_config(), P).eval()
    with torch.no_grad():
        ref = base(nodes, edges)
    # AlphaDropout keeps mean and variance of SELU activations: the logits stay on the same scale
    assert 0.5 < float(out.std() / ref.std()) < 2.0
