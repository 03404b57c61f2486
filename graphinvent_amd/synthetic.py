"""
Seeded synthetic molecular-graph minibatches in the reference's preprocessed-HDF contract.

The reference stores every training example as three int8 rows (``DataProcesser.py:157-161,
273-289``): ``nodes[N, Fn]`` (one-hot atom type ++ one-hot formal charge per occupied slot, zero
rows for padding), ``edges[N, N, Fe]`` (symmetric, one-hot bond type per bonded pair, no
self-loops) and ``APDs[N*A + N*Fe + 1]`` (non-negative counts, ``MolecularGraph.py:463-530``).
``BlockDatasetLoader.HDFDataset.__getitem__`` (``BlockDatasetLoader.py:135-143``) hands the model
fp32 copies of those rows.  There is no network for GDB-13/ZINC/ChEMBL here, so the benchmark and
the parity tests draw graphs of the same *shape* from this generator:

* node count per subgraph: ~2 % empty graphs, ~3 % single-atom graphs (every molecule's decoding
  route contributes one of each, ``MolecularGraph.py:676-689``), otherwise uniform on 2..N;
* bonds: random spanning tree (parent index < child index, as a BFS decoding route produces) plus
  floor(n/6) ring closures; bond type ~ Cat(0.85, 0.14, 0.01, ...) as in the shipped
  ``train.csv`` ``edge_feature_hist``;
* targets: sparse non-negative int8 rows with at least one non-zero entry.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def bond_type_probs(n_edge_features: int) -> np.ndarray:
    base = np.array([0.85, 0.14, 0.01] + [0.0] * max(0, n_edge_features - 3), dtype=np.float64)
    base = base[:n_edge_features]
    return base / base.sum()


def make_batch(batch_size: int, max_n_nodes: int, n_atom_types: int, n_formal_charge: int,
               n_edge_features: int = 3, seed: int = 0,
               frac_empty: float = 0.02, frac_single: float = 0.03,
               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Returns int8 ``(nodes[B,N,Fn], edges[B,N,N,Fe], apds[B, N*A+N*Fe+1])``."""
    rng = np.random.default_rng(seed)
    B, N, Fe = batch_size, max_n_nodes, n_edge_features
    Fn = n_atom_types + n_formal_charge
    A = n_atom_types * n_formal_charge * Fe
    apd_len = N * A + N * Fe + 1
    nodes = np.zeros((B, N, Fn), dtype=np.int8)
    edges = np.zeros((B, N, N, Fe), dtype=np.int8)
    probs = bond_type_probs(Fe)

    u = rng.random(B)
    sizes = np.where(u < frac_empty, 0,
                     np.where(u < frac_empty + frac_single, 1,
                              rng.integers(2, N + 1, size=B) if N >= 2 else 1))
    for b in range(B):
        n = int(sizes[b])
        if n == 0:
            continue
        atom = rng.integers(0, n_atom_types, size=n)
        charge = rng.integers(0, n_formal_charge, size=n)
        nodes[b, np.arange(n), atom] = 1
        nodes[b, np.arange(n), n_atom_types + charge] = 1
        if n == 1:
            continue
        child = np.arange(1, n)
        parent = (rng.random(n - 1) * child).astype(np.int64)      # parent index < child index
        btype = rng.choice(Fe, size=n - 1, p=probs)
        edges[b, child, parent, btype] = 1
        edges[b, parent, child, btype] = 1
        for _ in range(n // 6):
            i, j = rng.integers(0, n, size=2)
            if i != j and not edges[b, i, j].any():
                t = rng.choice(Fe, p=probs)
                edges[b, i, j, t] = 1
                edges[b, j, i, t] = 1

    apds = np.zeros((B, apd_len), dtype=np.int8)
    n_hot = rng.integers(1, 5, size=B)
    for b in range(B):
        idx = rng.integers(0, apd_len, size=int(n_hot[b]))
        np.add.at(apds[b], idx, 1)
    return nodes, edges, apds


# dataset shapes used by BASELINE.json's configs
SHAPES = {
    "gdb13": dict(max_n_nodes=13, n_atom_types=5, n_formal_charge=3, n_edge_features=3),
    "zinc": dict(max_n_nodes=38, n_atom_types=9, n_formal_charge=3, n_edge_features=3),
    "chembl": dict(max_n_nodes=88, n_atom_types=12, n_formal_charge=3, n_edge_features=3),
}
