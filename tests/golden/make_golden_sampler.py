"""Golden vectors for the sampling step from the UNMODIFIED reference
``GraphGenerator.get_actions`` / ``get_invalid_actions`` (GraphGenerator.py:467-657).

Runs only in the build container (needs /root/reference).  ``GraphGenerator.py`` imports rdkit, tqdm,
``parameters.constants`` and ``MolecularGraph`` at module level — none importable here — so empty stub
modules are injected for those names (they are not touched by the two methods under test), the
``constants`` the methods read (dim_nodes, dim_edges, dim_f_add, dim_f_conn) are supplied as a
namedtuple, and the one random draw, ``torch.distributions.Multinomial(1, probs).sample()``, is
replaced by a fixed one-hot so the outputs are reproducible.  The methods themselves run unchanged
on a bare instance (``object.__new__``) carrying ``batch_size``, ``n_nodes`` and ``edges``."""
import os
import sys
import types
from collections import namedtuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/graphinvent"

N, ATOMS, CHARGES, BONDS = 13, 5, 3, 3
DIMS = dict(dim_nodes=[N, ATOMS + CHARGES], dim_edges=[N, N, BONDS],
            dim_f_add=[N, ATOMS, CHARGES, BONDS], dim_f_conn=[N, BONDS])


def load_reference():
    for name in ("rdkit", "tqdm", "MolecularGraph", "parameters", "parameters.constants"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["tqdm"].tqdm = lambda *a, **k: None
    sys.modules["MolecularGraph"].GenerationGraph = object
    sys.modules["parameters.constants"].constants = namedtuple("C", sorted(DIMS))(**DIMS)
    sys.path.insert(0, REF)
    import GraphGenerator
    assert GraphGenerator.__file__.startswith(REF)
    return GraphGenerator


def make_inputs(seed=0, B=96):
    rng = np.random.default_rng(seed)
    n_nodes = rng.integers(0, N + 1, size=B).astype(np.int8)
    n_nodes[:6] = [0, 0, N, N, 1, 2]
    edges = np.zeros((B, N, N, BONDS), dtype=np.float32)
    for b in range(B):
        nn = int(n_nodes[b])
        for i in range(1, nn):                       # random tree + a few extra bonds
            j = int(rng.integers(0, i))
            t = int(rng.integers(0, BONDS))
            edges[b, i, j, t] = edges[b, j, i, t] = 1
        for _ in range(nn // 4):
            i, j = rng.integers(0, nn, size=2)
            if i != j and edges[b, i, j].sum() == 0:
                t = int(rng.integers(0, BONDS))
                edges[b, i, j, t] = edges[b, j, i, t] = 1
    A = ATOMS * CHARGES * BONDS
    W = N * A + N * BONDS + 1
    logits = rng.normal(size=(B, W)).astype(np.float32) * 2
    apds = torch.softmax(torch.from_numpy(logits), dim=1).numpy()
    # sampled flat indices: a mix of add / connect / terminate with every validity class present
    idx = np.empty(B, dtype=np.int64)
    for b in range(B):
        r = rng.random()
        if r < 0.45:
            idx[b] = rng.integers(0, N * A)
        elif r < 0.9:
            idx[b] = N * A + rng.integers(0, N * BONDS)
        else:
            idx[b] = W - 1
    nn = n_nodes.astype(int)
    idx[0] = 0 * A + 7                 # add to empty graph at node 0: valid
    idx[1] = 3 * A + 2                 # add to empty graph at node 3: invalid
    idx[2] = 5 * A                     # add to a full graph: invalid (max nodes)
    idx[3] = N * A + 4 * BONDS + 1     # connect in a full graph
    idx[4] = N * A + 0 * BONDS         # connect node 0 to itself (n_nodes = 1): self-loop
    idx[5] = N * A + 0 * BONDS + 2     # connect 0 - 1 in a 2-node graph: duplicate edge
    idx[6] = N * A + 0 * BONDS         # whatever graph 6 is
    return n_nodes, edges, logits, apds, idx


def main():
    GG = load_reference()
    n_nodes, edges, logits, apds, idx = make_inputs()
    B, W = apds.shape
    one_hot = torch.zeros(B, W)
    one_hot[torch.arange(B), torch.from_numpy(idx)] = 1

    class FixedMultinomial:                     # the one random draw, pinned
        def __init__(self, total_count, probs):
            assert total_count == 1 and probs.shape == one_hot.shape

        def sample(self):
            return one_hot.clone()

    torch.distributions.Multinomial = FixedMultinomial
    gen = object.__new__(GG.GraphGenerator)
    gen.batch_size = B
    gen.n_nodes = torch.from_numpy(n_nodes.copy())
    gen.edges = torch.from_numpy(edges.copy())
    add, conn, term, invalid, like = gen.get_actions(torch.from_numpy(apds))
    blob = dict(n_nodes=n_nodes, edges=edges.astype(np.int8), logits=logits, apds=apds, idx=idx,
                term=term.numpy(), invalid=invalid.numpy(), likelihoods=like.numpy(),
                dim_f_add=np.array(DIMS["dim_f_add"]), dim_f_conn=np.array(DIMS["dim_f_conn"]))
    for k, t in enumerate(add):
        blob[f"add{k}"] = t.numpy()
    for k, t in enumerate(conn):
        blob[f"conn{k}"] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "golden_sampler.npz"), **blob)
    print("adds", len(add[0]), "conns", len(conn[0]), "terms", len(term), "invalid", len(invalid),
          "add tuple len", len(add), "conn tuple len", len(conn))


if __name__ == "__main__":
    main()
