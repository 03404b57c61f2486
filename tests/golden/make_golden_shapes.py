"""
More golden fixtures from the UNMODIFIED reference (build container only, needs /root/reference):

    python tests/golden/make_golden_shapes.py

make_golden.py pins the oracle at the tiny configs and at the GGNN / GDB-13 default dimensions; this
script adds the other shapes BASELINE.json names and the variants the preprocessing can produce, each as
inputs + logits + loss + per-parameter gradient digests (weights are regenerated from the seed):

  golden_zinc.npz         gnn.mpnn.GGNN, reference default hyper-parameters on ZINC-shaped graphs
                          (max_n_nodes 38, 9 atom types x 3 charges: BASELINE configs[2])
  golden_att_gdb13.npz    gnn.mpnn.AttentionGGNN, default hyper-parameters, the shipped gdb13_1K-debug rows +
                          synthetic GDB-13-shaped graphs
  golden_att_chembl.npz   gnn.mpnn.AttentionGGNN on ChEMBL-shaped graphs (max_n_nodes 88, 12 atom types:
                          BASELINE configs[4]), small batch (the reference loops over nodes in Python)
  golden_aromatic.npz     gnn.mpnn.GGNN with n_edge_features = 4 (`use_aromatic_bonds` preprocessing,
                          parameters/constants.py:159-166): four message MLPs
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/graphinvent"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import ggnn_oracle as O                      # noqa: E402
from graphinvent_amd import synthetic                    # noqa: E402
from tests.golden.spec import digest                     # noqa: E402
from tests.golden.make_golden import reference_run       # noqa: E402  (imports the reference)
import gnn.mpnn as ref_mpnn                              # noqa: E402

assert ref_mpnn.__file__.startswith(REF), ref_mpnn.__file__


def save(name, cfg_overrides, seed, model, n8, e8, a8):
    kind = "AttGGNN" if model == "AttGGNN" else "GGNN"
    cfg = O.make_config(**cfg_overrides)
    P = O.init_params(cfg, seed=seed, model=kind)
    cls = ref_mpnn.AttentionGGNN if kind == "AttGGNN" else ref_mpnn.GGNN
    nodes, edges, target = (torch.from_numpy(x).float() for x in (n8, e8, a8))
    out, loss, grads = reference_run(cfg, P, nodes, edges, target, cls=cls)
    blob = dict(nodes=n8, edges=e8, apds=a8, logits=out.numpy(), loss=loss.numpy(), seed=np.asarray(seed),
                model=np.asarray(kind))
    blob.update({"cfg." + k: np.asarray(v) for k, v in cfg_overrides.items()})
    blob.update({"gdigest." + k: digest(v) for k, v in grads.items()})
    np.savez_compressed(f"{HERE}/{name}.npz", **blob)
    print(name, "loss", float(loss), "logits", tuple(out.shape), "params", sum(v.numel() for v in P.values()))


def shape_overrides(shape, n_edge_features=3):
    sh = synthetic.SHAPES[shape]
    na, nc, N = sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"]
    return dict(n_node_features=na + nc, n_edge_features=n_edge_features, max_n_nodes=N,
                len_f_add_per_node=na * nc * n_edge_features, len_f_conn_per_node=n_edge_features)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # GGNN, ZINC shape
    n8, e8, a8 = synthetic.make_batch(24, **synthetic.SHAPES["zinc"], seed=21)
    save("golden_zinc", shape_overrides("zinc"), 5, "GGNN", n8, e8, a8)
    # AttentionGGNN, GDB-13 default dims: shipped rows + synthetic
    fx = np.load(f"{HERE}/gdb13_1K-debug_test.npz")
    sn, se, sa = synthetic.make_batch(12, **synthetic.SHAPES["gdb13"], seed=23)
    n8 = np.concatenate([fx["nodes"][:20], sn]); e8 = np.concatenate([fx["edges"][:20], se])
    a8 = np.concatenate([fx["APDs"][:20], sa])
    save("golden_att_gdb13", {}, 6, "AttGGNN", n8, e8, a8)
    # AttentionGGNN, ChEMBL shape
    n8, e8, a8 = synthetic.make_batch(6, **synthetic.SHAPES["chembl"], seed=25)
    save("golden_att_chembl", shape_overrides("chembl"), 7, "AttGGNN", n8, e8, a8)
    # GGNN with aromatic bonds as a fourth edge feature
    n8, e8, a8 = synthetic.make_batch(20, 20, 6, 3, n_edge_features=4, seed=27)
    rng = np.random.default_rng(2)
    b, i, j = np.nonzero(np.triu(e8[..., 0], 1))
    pick = rng.random(b.size) < 0.25
    for bb, ii, jj in zip(b[pick], i[pick], j[pick]):
        e8[bb, ii, jj, 0] = e8[bb, jj, ii, 0] = 0
        e8[bb, ii, jj, 3] = e8[bb, jj, ii, 3] = 1
    assert e8[..., 3].any()
    ov = dict(n_node_features=9, n_edge_features=4, max_n_nodes=20, len_f_add_per_node=6 * 3 * 4,
              len_f_conn_per_node=4)
    save("golden_aromatic", ov, 8, "GGNN", n8, e8, a8)


if __name__ == "__main__":
    main()
