"""
ORACLE — test infrastructure only.  NOT part of the product path.

CPU (numpy) restatement of the reference's sampling step, ``GraphGenerator.get_actions`` and
``GraphGenerator.get_invalid_actions`` (GraphGenerator.py:467-657; file:line relative to
``/root/reference/graphinvent/``), for checking ``graphinvent_amd.sampler`` / ``gi_sample_actions``.

The reference draws the action with ``torch.distributions.Multinomial(1, probs).sample()`` (:533-537),
whose random stream cannot be reproduced by another implementation.  The draw is therefore a
parameter here: ``draw_inverse_cdf`` restates the categorical draw as the inverse CDF of a supplied
uniform per graph (what the HIP kernel does), and everything after the draw — reshaping the one-hot
into the add / connect / terminate segments, the index tuples, likelihoods and the invalid-action
rules — follows the reference line by line on the drawn index.

Parity pinning: ``tests/golden/golden_sampler.npz`` holds the outputs of the UNMODIFIED reference
methods with the draw fixed (``tests/golden/make_golden_sampler.py``); ``tests/test_sampler_cpu.py``
replays them against this file.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np


def softmax_rows(logits: np.ndarray) -> np.ndarray:
    """``torch.nn.Softmax(dim=1)`` of GraphGenerator.py:109,121 in float64."""
    z = logits.astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def draw_inverse_cdf(apds: np.ndarray, uniform: np.ndarray) -> np.ndarray:
    """One categorical draw per row: the first index whose cumulative probability exceeds
    u * total (stands in for Multinomial(1, probs).sample(), :533-537)."""
    cdf = np.cumsum(apds.astype(np.float64), axis=1)
    target = uniform.astype(np.float64) * cdf[:, -1]
    idx = (cdf <= target[:, None]).sum(axis=1)
    return np.minimum(idx, apds.shape[1] - 1)


def get_actions(apds: np.ndarray, idx: np.ndarray, n_nodes: np.ndarray, edges: np.ndarray,
                dim_f_add: Sequence[int], dim_f_conn: Sequence[int]) -> Dict[str, object]:
    """Everything ``get_actions`` does after the draw (:500-570), with ``get_invalid_actions``
    (:573-657).  Returns dict(add=tuple, conn=tuple, term, invalid, likelihoods, needs_reset)."""
    B, W = apds.shape
    n_max_nodes = dim_f_add[0]
    f_add_size = int(np.prod(dim_f_add))                                         # :519
    one_hot = np.zeros((B, W), dtype=np.int8)
    one_hot[np.arange(B), idx] = 1                                               # the draw
    f_add = one_hot[:, :f_add_size].reshape((B, *dim_f_add))                     # :522
    f_conn = one_hot[:, f_add_size:-1].reshape((B, *dim_f_conn))                 # :523
    f_term = one_hot[:, -1]                                                      # :524
    likelihoods = apds[one_hot == 1]                                             # :541
    add = list(np.nonzero(f_add))                                                # :543
    conn = list(np.nonzero(f_conn))                                              # :544
    term = np.nonzero(f_term)[0]                                                 # :545
    nn = n_nodes.astype(np.int64)
    add.append(nn[add[0]])                                                       # :556-557 f_add_from
    conn.append(nn[conn[0]] - 1)                                                 # :560-561 f_conn_from
    last = len(add) - 1                                                          # the reference's [5]

    # ---- get_invalid_actions (:573-657) ----
    add_empty = np.nonzero(nn[add[0]] == 0)[0]                                   # :602
    tmp = np.nonzero(add[1] >= nn[add[0]])[0]                                    # :605
    invalid_add = np.setxor1d(tmp, add_empty)                                    # :606-608 counts == 1
    tmp2 = np.nonzero(add[1] != nn[add[0]])[0]                                   # :611
    invalid_add_empty = np.intersect1d(tmp2, add_empty)                          # :612-614 counts > 1
    invalid_madd = np.nonzero(add[last] >= n_max_nodes)[0]                       # :617
    invalid_conn = np.nonzero(conn[1] >= nn[conn[0]])[0]                         # :620
    invalid_conn_nonex = np.nonzero(nn[conn[0]] == 0)[0]                         # :623
    invalid_sconn = np.nonzero(conn[1] == conn[3])[0]                            # :626
    adjacency = edges.astype(np.int64).sum(-1)
    invalid_dconn = np.nonzero(adjacency[conn[0], conn[1], conn[-1]] == 1)[0]    # :629-633 (from = -1 wraps)
    invalid = np.unique(np.concatenate([                                         # :636-646
        add[0][invalid_add], add[0][invalid_add_empty], conn[0][invalid_conn],
        conn[0][invalid_conn_nonex], conn[0][invalid_sconn], conn[0][invalid_dconn],
        add[0][invalid_madd]]))
    needs_reset = np.unique(np.concatenate([invalid_madd, add_empty]))           # :650-654
    add[last] = add[last].copy()
    add[last][needs_reset] = 0                                                   # :567
    return dict(add=tuple(add), conn=tuple(conn), term=term, invalid=invalid,
                likelihoods=likelihoods, needs_reset=needs_reset)
