"""End-to-end use of the MI355X path in the shape of the reference's training loop
(Workflow.train_epoch, Workflow.py:766-798) on the reference's shipped preprocessed data
(data/pre-training/gdb13_1K-debug/train.h5; here the committed .npz conversion of it):

    HDF / int8 rows -> BlockStreamLoader (block-wise, like BlockDatasetLoader.py:77-99) -> gnn.mpnn.GGNN(constants)
    -> apd_kl_loss -> FusedAdam

    python examples/train_fixture.py [--epochs 30] [--batch 32] [--model GGNN|AttGGNN] [--h5 path.h5]

With --h5 the three datasets are STREAMED from a GraphINVENT .h5 file through libhdf5 (ctypes), a block at a time."""
import argparse
import os
import sys
from collections import namedtuple

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphinvent_amd import dp                                    # noqa: E402
from graphinvent_amd.gnn import mpnn                              # noqa: E402
from graphinvent_amd.loader import ArraySource, BlockStreamLoader, HDFSource   # noqa: E402
from graphinvent_amd.loss import apd_kl_loss                      # noqa: E402
from graphinvent_amd.optim import FusedAdam                       # noqa: E402


def constants_for(nodes, edges, apds):
    """The fields of parameters/constants.py the model reads, derived from the data's shapes and the
    reference's default hyper-parameters (parameters/defaults.py:280-363)."""
    N, Fn = nodes.shape[1:]
    Fe = edges.shape[3]
    A = (apds.shape[1] - 1 - N * Fe) // N
    c = dict(device="cuda", big_positive=1e6, big_negative=-1e6, n_node_features=Fn, n_edge_features=Fe,
             max_n_nodes=N, len_f_add_per_node=A, len_f_conn_per_node=Fe, hidden_node_features=100,
             message_size=100, message_passes=3, enn_depth=4, enn_hidden_dim=250, enn_dropout_p=0.0,
             gather_width=100, gather_att_depth=4, gather_att_hidden_dim=250, gather_att_dropout_p=0.0,
             gather_emb_depth=4, gather_emb_hidden_dim=250, gather_emb_dropout_p=0.0, mlp1_depth=4,
             mlp1_hidden_dim=500, mlp1_dropout_p=0.0, mlp2_depth=4, mlp2_hidden_dim=500,
             mlp2_dropout_p=0.0, msg_depth=4, msg_hidden_dim=250, msg_dropout_p=0.0, att_depth=4,
             att_hidden_dim=250, att_dropout_p=0.0)
    return namedtuple("CONSTANTS", sorted(c))(**c)


def train(epochs=30, batch=32, model_name="GGNN", h5=None, seed=0, verbose=True):
    if h5:
        source = HDFSource(h5)
        nodes, edges, apds = (np.empty((1,) + tuple(shp), dtype=np.int8) for shp in source.row_shapes)   # shapes only
    else:
        d = np.load(os.path.join(ROOT, "tests", "golden", "gdb13_1K-debug_train.npz"))
        nodes, edges, apds = d["nodes"], d["edges"], d["APDs"]
        source = ArraySource(nodes, edges, apds)
    torch.manual_seed(seed)
    cls = mpnn.GGNN if model_name == "GGNN" else mpnn.AttentionGGNN
    model = cls(constants_for(nodes, edges, apds)).to("cuda").train()
    opt = FusedAdam(model.parameters(), lr=1e-4)                  # defaults.py:120 init_lr
    # trailing all-zero target rows (dataset-size padding) are trimmed; the ragged last minibatch is kept
    loader = BlockStreamLoader(source, batch, block_size=max(batch, 10000), seed=seed)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=epochs * len(loader) + 1)
    trainer = dp.DataParallel(model, opt, sched, loss_fn=apd_kl_loss)
    history = []
    for epoch in range(epochs):
        loader.set_epoch(epoch)
        total = torch.zeros((), device="cuda")
        for nb, eb, ab in loader:
            total += trainer.step(nb, eb, ab)
        history.append(float(total) / len(loader))
        if verbose:
            print(f"epoch {epoch:3d}  mean training loss {history[-1]:.4f}")
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--model", default="GGNN", choices=["GGNN", "AttGGNN"])
    ap.add_argument("--h5", default=None)
    a = ap.parse_args()
    train(a.epochs, a.batch, a.model, a.h5)
