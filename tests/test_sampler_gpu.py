"""-m gpu: the fused sampling step (gi_sample_actions / graphinvent_amd.sampler) against the sampler
oracle (pinned to the reference's get_actions by tests/test_sampler_cpu.py).
  * the draw: the kernel's index brackets u in the fp64 CDF of the fp64 softmax (fp32 cumulative sums
    may move a draw across a boundary only when u sits within 1e-5 of it);
  * everything after the draw (index tuples, invalid set, reset rule): bit-exact vs the oracle given
    the kernel's own drawn index; likelihoods to 1e-5;
  * statistics: empirical action frequencies vs the softmax (chi-square)."""
import os

import numpy as np
import pytest
import torch

from graphinvent_amd import sampler
from oracle import sampler_oracle as SO

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _flat_index(action, N, A, Fe):
    kind, node, rem = action[:, 0], action[:, 1], action[:, 2]
    return np.where(kind == 0, node * A + rem, np.where(kind == 1, N * A + node * Fe + rem,
                                                        N * A + N * Fe))


@pytest.mark.parametrize("edge_dtype", [torch.float32, torch.int8])
def test_sampler_matches_oracle_on_reference_fixture(golden_dir, edge_dtype):
    g = np.load(os.path.join(golden_dir, "golden_sampler.npz"))
    dim_f_add, dim_f_conn = g["dim_f_add"].tolist(), g["dim_f_conn"].tolist()
    N, Fe = dim_f_conn
    A = int(np.prod(dim_f_add[1:]))
    logits = torch.from_numpy(g["logits"]).to(DEV)
    n_nodes = torch.from_numpy(g["n_nodes"]).to(DEV)                    # int8 like the reference
    edges = torch.from_numpy(g["edges"]).to(DEV).to(edge_dtype)
    B, W = g["logits"].shape
    p64 = SO.softmax_rows(g["logits"])
    cdf = np.cumsum(p64, axis=1)
    gen = torch.Generator(device=DEV).manual_seed(5)
    for trial in range(8):
        u = torch.rand(B, device=DEV, generator=gen)
        if trial == 0:
            u[:4] = torch.tensor([0.0, 0.999999, 0.5, 1e-7], device=DEV)
        action, like, flags = sampler.sample_actions_raw(logits, n_nodes, edges, A, uniform=u)
        a = action.cpu().numpy()
        idx = _flat_index(a, N, A, Fe)
        un = u.cpu().numpy().astype(np.float64)
        lo = np.where(idx > 0, cdf[np.arange(B), np.maximum(idx - 1, 0)], 0.0)
        hi = cdf[np.arange(B), idx]
        assert np.all(lo - 1e-5 <= un) and np.all(un < hi + 1e-5), "draw is not the inverse CDF of u"
        assert np.array_equal(idx, SO.draw_inverse_cdf(p64, un)) or \
            np.mean(idx != SO.draw_inverse_cdf(p64, un)) < 0.02            # boundary ties only
        # everything after the draw, in the reference's return format
        out = sampler.sample_actions(logits, n_nodes, edges, dim_f_add, dim_f_conn, uniform=u)
        ref = SO.get_actions(p64, idx, g["n_nodes"], g["edges"], dim_f_add, dim_f_conn)
        f_add, f_conn, f_term, invalid, likelihoods = out
        assert len(f_add) == len(ref["add"]) == 6 and len(f_conn) == len(ref["conn"]) == 4
        for k in range(6):
            assert np.array_equal(f_add[k].cpu().numpy(), ref["add"][k]), f"add[{k}]"
        for k in range(4):
            assert np.array_equal(f_conn[k].cpu().numpy(), ref["conn"][k]), f"conn[{k}]"
        assert np.array_equal(f_term.cpu().numpy(), ref["term"])
        assert np.array_equal(invalid.cpu().numpy(), ref["invalid"])
        assert np.max(np.abs(likelihoods.cpu().numpy() - ref["likelihoods"])) < 1e-5
        fl = flags.cpu().numpy()
        reset_graphs = f_add[0].cpu().numpy()[ref["needs_reset"]]
        assert np.array_equal(np.nonzero(fl & 2)[0], np.sort(reset_graphs))


def test_sampler_reproduces_the_reference_draws_when_given_their_uniforms(golden_dir):
    """Feed uniforms that land inside the probability interval of the action the reference fixture
    drew: the kernel must return exactly the reference's tuples."""
    g = np.load(os.path.join(golden_dir, "golden_sampler.npz"))
    p64 = SO.softmax_rows(g["logits"])
    cdf = np.cumsum(p64, axis=1)
    B = p64.shape[0]
    idx = g["idx"]
    lo = np.where(idx > 0, cdf[np.arange(B), np.maximum(idx - 1, 0)], 0.0)
    u = torch.from_numpy((lo + 0.5 * p64[np.arange(B), idx]).astype(np.float32)).to(DEV)
    out = sampler.sample_actions(torch.from_numpy(g["logits"]).to(DEV),
                                 torch.from_numpy(g["n_nodes"]).to(DEV),
                                 torch.from_numpy(g["edges"]).to(DEV).float(),
                                 g["dim_f_add"].tolist(), g["dim_f_conn"].tolist(), uniform=u)
    f_add, f_conn, f_term, invalid, likelihoods = out
    for k in range(6):
        assert np.array_equal(f_add[k].cpu().numpy(), g[f"add{k}"]), f"add[{k}]"
    for k in range(4):
        assert np.array_equal(f_conn[k].cpu().numpy(), g[f"conn{k}"]), f"conn[{k}]"
    assert np.array_equal(f_term.cpu().numpy(), g["term"])
    assert np.array_equal(invalid.cpu().numpy(), g["invalid"])
    assert np.max(np.abs(likelihoods.cpu().numpy() - g["likelihoods"])) < 1e-6


def test_sampler_statistics_and_large_rows():
    """ChEMBL-shaped APD rows (W = 9769): frequencies of 4096 x 64 draws from one distribution."""
    N, A, Fe = 88, 108, 3
    W = N * A + N * Fe + 1
    gen = torch.Generator(device=DEV).manual_seed(1)
    row = torch.randn(W, device=DEV, generator=gen) * 3
    B = 4096
    logits = row.repeat(B, 1)
    n_nodes = torch.full((B,), 5, dtype=torch.int8, device=DEV)
    edges = torch.zeros((B, N, N, Fe), dtype=torch.int8, device=DEV)
    counts = np.zeros(W)
    for _ in range(64):
        action, like, flags = sampler.sample_actions_raw(logits, n_nodes, edges, A, generator=gen)
        counts += np.bincount(_flat_index(action.cpu().numpy(), N, A, Fe), minlength=W)
    p = SO.softmax_rows(row.cpu().numpy()[None])[0]
    n = counts.sum()
    big = n * p >= 5                                                   # chi-square on well-filled bins
    chi2 = float((((counts - n * p) ** 2)[big] / (n * p[big])).sum())
    dof = int(big.sum())
    assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof), (chi2, dof)
    assert counts[~big].sum() <= 3 * n * p[~big].sum() + 50
    with pytest.raises(RuntimeError):
        sampler.sample_actions_raw(logits.cpu(), n_nodes, edges, A)


def test_sampler_with_five_add_dimensions():
    """dim_f_add with implicit-H / chirality style extra factors (parameters/constants.py:56-89): the
    unravelling of the per-node index must follow torch.nonzero's row-major order."""
    N, dims, Fe = 9, [4, 3, 2, 3, 3], 3                     # (atom, charge, imp_H, chirality, bond)
    dim_f_add, dim_f_conn = [N, *dims], [N, Fe]
    A = int(np.prod(dims))
    W = N * A + N * Fe + 1
    B = 200
    gen = torch.Generator(device=DEV).manual_seed(2)
    logits = torch.randn(B, W, device=DEV, generator=gen)
    n_nodes = torch.randint(0, N + 1, (B,), device=DEV, generator=gen).to(torch.int8)
    edges = torch.zeros((B, N, N, Fe), dtype=torch.int8, device=DEV)
    u = torch.rand(B, device=DEV, generator=gen)
    action, like, flags = sampler.sample_actions_raw(logits, n_nodes, edges, A, uniform=u)
    idx = _flat_index(action.cpu().numpy(), N, A, Fe)
    out = sampler.sample_actions(logits, n_nodes, edges, dim_f_add, dim_f_conn, uniform=u)
    ref = SO.get_actions(SO.softmax_rows(logits.cpu().numpy()), idx, n_nodes.cpu().numpy(),
                         edges.cpu().numpy(), dim_f_add, dim_f_conn)
    assert len(out[0]) == len(ref["add"]) == 8
    for k in range(8):
        assert np.array_equal(out[0][k].cpu().numpy(), ref["add"][k]), k
    for k in range(4):
        assert np.array_equal(out[1][k].cpu().numpy(), ref["conn"][k]), k
    assert np.array_equal(out[2].cpu().numpy(), ref["term"])
    assert np.array_equal(out[3].cpu().numpy(), ref["invalid"])
