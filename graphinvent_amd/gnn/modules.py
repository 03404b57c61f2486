"""
Parameter containers with the reference's names, registration order and ``state_dict`` keys
(gnn/modules.py:12-52 ``GraphGather``, :111-170 ``MLP``, :173-281 ``GlobalReadout``).

They hold weights only.  All arithmetic of the GGNN hot path runs in the hand-written HIP kernels
behind ``graphinvent_amd.lib`` (one fused call per forward / backward, see ``gnn/mpnn.py``), so
these classes have no per-module ``forward``.
"""
from __future__ import annotations

from typing import List

import torch


class _LinearStack(torch.nn.Module):
    """Holds the Linear layers of an MLP under the child names 0, 3, 6, ... — the positions the
    Linear modules occupy inside the reference's ``Sequential(Linear, SELU, AlphaDropout, ...)``
    (gnn/modules.py:130-142), which is what makes the checkpoint keys ``seq.{0,3,..}.weight``."""

    def __init__(self, sizes: List[int]):
        super().__init__()
        for layer, (fan_in, fan_out) in enumerate(zip(sizes, sizes[1:])):
            lin = torch.nn.Linear(fan_in, fan_out, bias=True)
            torch.nn.init.xavier_uniform_(lin.weight)       # gnn/modules.py:163
            self.add_module(str(3 * layer), lin)

    def linears(self) -> List[torch.nn.Linear]:
        return list(self.children())


class MLP(torch.nn.Module):
    """depth+1 Linear layers, SELU after every one including the last (gnn/modules.py:111-170).
    ``dropout_p`` > 0: ``torch.nn.AlphaDropout`` behind every Linear + SELU in ``train()`` mode, on the HIP path
    (gi_alpha_dropout_fwd); the shipped default p = 0.0 and ``eval()`` are the identity."""

    def __init__(self, in_features: int, hidden_layer_sizes: list, out_features: int,
                 dropout_p: float) -> None:
        super().__init__()
        self.dropout_p = float(dropout_p)
        self.in_features = in_features
        self.out_features = out_features
        self.hidden_layer_sizes = list(hidden_layer_sizes)
        self.seq = _LinearStack([in_features, *hidden_layer_sizes, out_features])

    def forward(self, *_):
        raise RuntimeError("graphinvent_amd MLP is a parameter container; the fused HIP model "
                           "(gnn.mpnn.GGNN.forward) runs it")


class GraphGather(torch.nn.Module):
    def __init__(self, node_features: int, hidden_node_features: int, out_features: int,
                 att_depth: int, att_hidden_dim: int, att_dropout_p: float, emb_depth: int,
                 emb_hidden_dim: int, emb_dropout_p: float, big_positive: float) -> None:
        super().__init__()
        self.big_positive = big_positive
        self.att_nn = MLP(node_features + hidden_node_features, [att_hidden_dim] * att_depth,
                          out_features, att_dropout_p)
        self.emb_nn = MLP(hidden_node_features, [emb_hidden_dim] * emb_depth, out_features,
                          emb_dropout_p)


class GlobalReadout(torch.nn.Module):
    def __init__(self, f_add_elems: int, f_conn_elems: int, f_term_elems: int, mlp1_depth: int,
                 mlp1_dropout_p: float, mlp1_hidden_dim: int, mlp2_depth: int,
                 mlp2_dropout_p: float, mlp2_hidden_dim: int, graph_emb_size: int,
                 max_n_nodes: int, node_emb_size: int, device: str) -> None:
        super().__init__()
        self.device = device
        self.fAddNet1 = MLP(node_emb_size, [mlp1_hidden_dim] * mlp1_depth, f_add_elems,
                            mlp1_dropout_p)
        self.fConnNet1 = MLP(node_emb_size, [mlp1_hidden_dim] * mlp1_depth, f_conn_elems,
                             mlp1_dropout_p)
        self.fAddNet2 = MLP(max_n_nodes * f_add_elems + graph_emb_size,
                            [mlp2_hidden_dim] * mlp2_depth, f_add_elems * max_n_nodes,
                            mlp2_dropout_p)
        self.fConnNet2 = MLP(max_n_nodes * f_conn_elems + graph_emb_size,
                             [mlp2_hidden_dim] * mlp2_depth, f_conn_elems * max_n_nodes,
                             mlp2_dropout_p)
        self.fTermNet2 = MLP(graph_emb_size, [mlp2_hidden_dim] * mlp2_depth, f_term_elems,
                             mlp2_dropout_p)
