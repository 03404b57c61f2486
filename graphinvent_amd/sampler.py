"""
Sampling step of graph generation on the device: what
``GraphGenerator.build_graphs`` does between the model call and ``apply_actions``
(GraphGenerator.py:121-124, 467-657) — softmax of the APD logits, one categorical draw per graph,
splitting the draw into "add" / "connect" / "terminate" index tuples, likelihoods, and the
invalid-action rules — as ONE HIP launch (``gi_sample_actions``) plus a handful of tiny index ops to
lay the result out in the reference's return format.

``sample_actions`` returns exactly what ``GraphGenerator.get_actions(apds)`` returns:
``(f_add_idc, f_conn_idc, f_term_idc, invalid_idc, likelihoods)`` with
``f_add_idc = (graph, node_to, *add sub-indices (atom type, formal charge, ..., bond type), from)`` and
``f_conn_idc = (graph, node_to, bond type, from)``, graphs ascending.  The draw itself is an
inverse-CDF draw from uniforms of a ``torch.Generator`` (same distribution as
``torch.distributions.Multinomial(1, probs).sample()``, not the same random stream).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import lib as L


def sample_actions_raw(logits: torch.Tensor, n_nodes: torch.Tensor, edges: torch.Tensor,
                       n_add_per_node: int, uniform: Optional[torch.Tensor] = None,
                       generator: Optional[torch.Generator] = None):
    """One launch; returns (action[B,4] int32 = kind, node_to, rem, from; likelihood[B]; flags[B])."""
    lib = L.load()
    if not logits.is_cuda:
        raise RuntimeError("sample_actions needs CUDA (ROCm) tensors: the MI355X HIP path has no CPU "
                           "fallback")
    logits = logits.float().contiguous()
    B, W = logits.shape
    N, Fe = edges.shape[1], edges.shape[3]
    if W != N * n_add_per_node + N * Fe + 1:
        raise ValueError(f"APD width {W} does not match N={N}, A={n_add_per_node}, Fe={Fe}")
    if edges.dtype == torch.int8:
        edges, dt = edges.contiguous(), L.DTYPE_I8
    else:
        edges, dt = edges.float().contiguous(), L.DTYPE_F32
    if uniform is None:
        uniform = torch.rand(B, device=logits.device, generator=generator)
    uniform = uniform.float().contiguous()
    nn32 = n_nodes.to(device=logits.device, dtype=torch.int32).contiguous()
    action = torch.empty((B, 4), dtype=torch.int32, device=logits.device)
    like = torch.empty(B, dtype=torch.float32, device=logits.device)
    flags = torch.empty(B, dtype=torch.int32, device=logits.device)
    with torch.cuda.device(logits.device):       # launch on the logits' device, whichever is current
        L.check(lib.gi_sample_actions(logits.data_ptr(), logits.stride(0), uniform.data_ptr(),
                                      nn32.data_ptr(), edges.data_ptr(), dt, B, N, n_add_per_node, Fe,
                                      action.data_ptr(), like.data_ptr(), flags.data_ptr(),
                                      torch.cuda.current_stream(logits.device).cuda_stream),
                "gi_sample_actions")
    return action, like, flags


def sample_actions(logits: torch.Tensor, n_nodes: torch.Tensor, edges: torch.Tensor,
                   dim_f_add: Sequence[int], dim_f_conn: Sequence[int],
                   uniform: Optional[torch.Tensor] = None,
                   generator: Optional[torch.Generator] = None) -> Tuple:
    """Drop-in for ``GraphGenerator.get_actions`` taking the model's raw logits (the softmax of
    GraphGenerator.py:121 is fused in).  ``dim_f_add`` / ``dim_f_conn`` = ``constants.dim_f_add`` /
    ``constants.dim_f_conn`` (parameters/constants.py:43-95)."""
    sub = [int(x) for x in dim_f_add[1:]]
    A = 1
    for x in sub:
        A *= x
    if int(dim_f_conn[1]) != edges.shape[3] or int(dim_f_add[0]) != edges.shape[1]:
        raise ValueError("dim_f_add / dim_f_conn do not match the edges tensor")
    action, like, flags = sample_actions_raw(logits, n_nodes, edges, A, uniform, generator)
    kind, node, rem, frm = (action[:, k].long() for k in range(4))
    graphs = torch.arange(action.shape[0], device=action.device)
    is_add, is_conn = kind == 0, kind == 1
    g_add, rem_add = graphs[is_add], rem[is_add]
    parts = []
    for size in reversed(sub):                       # unravel rem over the add sub-dimensions
        parts.append(rem_add % size)
        rem_add = rem_add // size
    f_add_idc = (g_add, node[is_add], *reversed(parts), frm[is_add])
    f_conn_idc = (graphs[is_conn], node[is_conn], rem[is_conn], frm[is_conn])
    f_term_idc = graphs[kind == 2]
    invalid_idc = graphs[(flags & 1) != 0]
    return f_add_idc, f_conn_idc, f_term_idc, invalid_idc, like
