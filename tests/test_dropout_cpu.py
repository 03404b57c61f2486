"""CPU: the oracle's AlphaDropout restatement against torch's own, and the no-row-sharing graph model."""
import numpy as np
import torch

from oracle import ggnn_oracle as O
from tests import ref_dataflow as D
from tests.golden.spec import tiny_inputs


def test_oracle_alpha_dropout_with_a_given_mask_is_torch_alpha_dropout():
    """Recover the mask torch drew from its output, feed it to the oracle: bit-identical result."""
    for p in (0.05, 0.3, 0.7):
        x = torch.selu(torch.randn(200, 64, generator=torch.Generator().manual_seed(1)) * 2)
        torch.manual_seed(42)
        want = torch.nn.functional.alpha_dropout(x, p, training=True)
        alpha = O.ALPHA_DROPOUT_ALPHA
        a = 1.0 / np.sqrt((alpha * alpha * p + 1) * (1 - p))
        b_drop = torch.zeros(1).add(-1).mul_(alpha * a).add_(alpha * a * p)      # ATen's op sequence
        keep = want != b_drop
        assert abs(keep.float().mean().item() - (1 - p)) < 0.03
        O.DROPOUT_HOOK = lambda prefix, layer, t: (p, keep)
        try:
            got = O._alpha_dropout(x, "any", 0)
        finally:
            O.DROPOUT_HOOK = None
        assert torch.equal(got, want)
    assert O._alpha_dropout(x, "any", 0) is x                # hook off: identity (eval / p = 0)


def test_oracle_mlp_applies_dropout_after_every_layer_including_the_last():
    cfg = O.make_config(mlp2_dropout_p=0.5)
    P = O.init_params(cfg, seed=0)
    x = torch.randn(4, cfg["gather_width"])
    seen = []

    def hook(prefix, layer, t):
        seen.append((prefix, layer, tuple(t.shape)))
        return 0.5, torch.zeros_like(t, dtype=torch.bool)    # drop everything
    O.DROPOUT_HOOK = hook
    try:
        y = O.mlp(P, "APDReadout.fTermNet2", x)
    finally:
        O.DROPOUT_HOOK = None
    assert [s[1] for s in seen] == list(range(cfg["mlp2_depth"] + 1))
    assert torch.unique(y).numel() == 1                       # every element = the dropped constant


def test_graph_model_without_row_sharing():
    n8, e8, _ = tiny_inputs()
    g, d = D.compact(n8, e8, nodedup=True), D.compact(n8, e8)
    B, N, _ = n8.shape
    assert g["S"] == B * N and g["U"] == g["E"] == d["E"] and g["D0"] == 0 and d["U"] < d["E"]
    assert np.array_equal(g["cidx"], np.arange(B * N))
    assert np.array_equal(np.sort(g["in_perm"]), np.arange(g["E"]))          # one row per edge
    assert np.array_equal(g["mu_off"], np.arange(g["E"] + 1))
    assert np.array_equal(g["mu_slot"], np.argsort(g["in_perm"], kind="stable"))
    # the same edges, the same per-type counts, the same sources as the shared-row layout
    assert np.array_equal(np.diff(g["type_off"]), np.bincount(
        np.repeat(np.arange(len(d["type_off"]) - 1), np.diff(d["type_off"]))[d["in_perm"]],
        minlength=len(d["type_off"]) - 1))
    src_shared = d["slot_of"][d["u_src"][d["in_perm"]]]                     # source slot per edge
    assert np.array_equal(g["u_src"][g["in_perm"]], src_shared)
