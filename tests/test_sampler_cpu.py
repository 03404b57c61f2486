"""The sampler oracle against the outputs of the unmodified reference ``GraphGenerator.get_actions``
(tests/golden/golden_sampler.npz, draw fixed), and the inverse-CDF draw's statistics."""
import os

import numpy as np

from oracle import sampler_oracle as SO


def test_oracle_reproduces_reference_get_actions(golden_dir):
    g = np.load(os.path.join(golden_dir, "golden_sampler.npz"))
    out = SO.get_actions(g["apds"], g["idx"], g["n_nodes"], g["edges"], g["dim_f_add"].tolist(),
                         g["dim_f_conn"].tolist())
    assert len(out["add"]) == 6 and len(out["conn"]) == 4
    for k in range(6):
        assert np.array_equal(out["add"][k], g[f"add{k}"]), f"add[{k}]"
    for k in range(4):
        assert np.array_equal(out["conn"][k], g[f"conn{k}"]), f"conn[{k}]"
    assert np.array_equal(out["term"], g["term"])
    assert np.array_equal(out["invalid"], g["invalid"])
    assert np.array_equal(out["likelihoods"], g["likelihoods"])
    # every validity class of get_invalid_actions is present in the fixture
    assert 0 not in out["invalid"] and {1, 2, 4, 5} <= set(out["invalid"].tolist())
    # the softmax restatement is what produced `apds`
    assert np.max(np.abs(SO.softmax_rows(g["logits"]) - g["apds"])) < 1e-7


def test_inverse_cdf_draw_follows_the_distribution():
    rng = np.random.default_rng(3)
    p = rng.random(40)
    p[[3, 17]] = 0.0
    p /= p.sum()
    n = 200_000
    idx = SO.draw_inverse_cdf(np.tile(p, (n, 1)), rng.random(n))
    counts = np.bincount(idx, minlength=40)
    assert counts[3] == 0 and counts[17] == 0                      # zero-probability actions never drawn
    nz = p > 0
    chi2 = float((((counts - n * p) ** 2)[nz] / (n * p[nz])).sum())
    assert chi2 < 80                                                # 37 dof: p(chi2 > 80) < 1e-4
