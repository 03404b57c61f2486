"""CPU model of the NEXT row reduction: Weisfeiler-Lehman-style
de-duplication of the message passes beyond pass 0.

After pass p the hidden state of a node depends only on its colour c_p:
    c_0(v) = feature pattern of v                      (the pass-0 classes the library already uses)
    c_p(v) = (c_{p-1}(v), multiset {(bond type, c_{p-1}(u)) : u -> v})
(gnn/summation_mpnn.py:128-144: m_e = MLP_type(e)(h_src(e)), a_v = sum of incoming m_e, h_v <- GRU(a_v, h_v) — the
same function of the same arguments for every node of a colour, in whatever graph of the batch it sits).  So
pass p needs one message row per distinct (bond type, source colour) pair, one aggregation + GRU row per colour,
and `a = C_p . m` with the colour-level edge-count matrix C_p — the pass-0 construction (`cmat`) at every pass.
The readout then gathers h through the last colour map.  Exact in exact arithmetic; on the device the summation
order inside a colour must be canonical (sorted pairs) for the rows to be bitwise equal to what any member would
have computed.

`wl_plan` = the index arrays a device implementation has to reproduce bit for bit (this is their numpy
reference); `forward` = the logits through them (fp64 proof against the oracle in tests/test_wl_cpu.py)."""
from typing import Dict, List

import numpy as np
import torch

from tests import ref_dataflow as D


def wl_plan(g: Dict[str, np.ndarray], x_rows: np.ndarray, passes: int) -> List[dict]:
    """g: tests.ref_dataflow.compact(...) of the batch; x_rows [R, Fn]: feature row of every compact row (row S = 0).
    Returns one dict per pass p = 0 .. passes-1 describing the pass that maps colours c_p to c_{p+1}:
      cls      [R]   compact row -> colour c_p (ids in order of first appearance by ascending row)
      ncls           number of colours c_p
      rep      [ncls] lowest compact row of every colour
      mpairs   [nm, 2] distinct (bond type, source colour) pairs with at least one edge, sorted
      cnt      [ncls_next, nm] edge counts: how many edges of pair j enter a node of next colour k
      nxt      [R]   compact row -> colour c_{p+1};  ncls_next, rep_next like above
    """
    S, E, R = g["S"], g["E"], g["S"] + 1
    in_perm, seg_off, u_src, type_off = g["in_perm"], g["seg_off"], g["u_src"], g["type_off"]
    Fe = len(type_off) - 1
    u_type = np.zeros(len(u_src), dtype=np.int64)
    for t in range(Fe):
        u_type[type_off[t]:type_off[t + 1]] = t

    def ids_of(keys):
        table, out = {}, np.empty(len(keys), dtype=np.int64)
        for r, k in enumerate(keys):
            out[r] = table.setdefault(k, len(table))
        return out, len(table)

    cls, ncls = ids_of([tuple(row.tolist()) for row in x_rows])
    plans = []
    for _ in range(passes):
        rep = np.full(ncls, R, dtype=np.int64)
        np.minimum.at(rep, cls, np.arange(R))
        # incoming (bond type, source colour) pairs of every compact row, in dst-CSR order
        pair_of_edge = [(int(u_type[in_perm[e]]), int(cls[u_src[in_perm[e]]])) for e in range(E)]
        mpairs = sorted(set(pair_of_edge))
        mid = {p: j for j, p in enumerate(mpairs)}
        keys = []
        for r in range(R):
            inc = sorted(pair_of_edge[e] for e in range(seg_off[r], seg_off[r + 1])) if r < S else []
            keys.append((int(cls[r]), tuple(inc)))
        nxt, ncls_next = ids_of(keys)
        rep_next = np.full(ncls_next, R, dtype=np.int64)
        np.minimum.at(rep_next, nxt, np.arange(R))
        cnt = np.zeros((ncls_next, len(mpairs)), dtype=np.int64)
        for k in range(ncls_next):
            r = rep_next[k]
            if r < S:
                for e in range(seg_off[r], seg_off[r + 1]):
                    cnt[k, mid[pair_of_edge[e]]] += 1
        plans.append(dict(cls=cls, ncls=ncls, rep=rep, mpairs=np.array(mpairs, dtype=np.int64).reshape(-1, 2),
                          cnt=cnt, nxt=nxt, ncls_next=ncls_next, rep_next=rep_next))
        cls, ncls = nxt, ncls_next
    return plans


def forward(P, cfg, nodes: torch.Tensor, edges: torch.Tensor):
    """GGNN logits through the colour-level dataflow (sum-aggregating model).  Returns (logits, stats) with the
    row counts per pass: colours / message pairs against the compact rows / message rows the library runs today."""
    dtype = nodes.dtype
    B, N, Fn = nodes.shape
    H, M = cfg["hidden_node_features"], cfg["message_size"]
    Fe, A, C = cfg["n_edge_features"], cfg["len_f_add_per_node"], cfg["len_f_conn_per_node"]
    g = D.compact(nodes.numpy(), edges.numpy())
    assert g["err"] == 0
    S, R = g["S"], g["S"] + 1
    T = {k: torch.from_numpy(v) for k, v in g.items() if isinstance(v, np.ndarray)}
    x = torch.zeros(R, Fn, dtype=dtype)
    x[:S] = nodes.reshape(B * N, Fn)[T["slot_of"].long()]
    plans = wl_plan(g, x.numpy(), cfg["message_passes"])
    stats = []
    # state per colour c_0: [x | 0]
    p0 = plans[0] if plans else None
    if plans:
        Hc = torch.zeros(p0["ncls"], H, dtype=dtype)
        Hc[:, :Fn] = x[torch.from_numpy(p0["rep"])]
    for pl in plans:
        nm = len(pl["mpairs"])
        m = torch.zeros(nm, M, dtype=dtype)
        for t in range(Fe):
            sel = np.nonzero(pl["mpairs"][:, 0] == t)[0]
            if len(sel):
                src = torch.from_numpy(pl["mpairs"][sel, 1])
                m[torch.from_numpy(sel)] = D.mlp_fwd(P, f"msg_nns.{t}", Hc, idx=src)[-1]
        agg = torch.from_numpy(pl["cnt"]).to(dtype) @ m                       # [ncls_next, M]
        prev = Hc[torch.from_numpy(pl["cls"][pl["rep_next"]])]                # previous state of every next colour
        gi = D.linear(agg, P["gru.weight_ih"], P["gru.bias_ih"], False)
        gh = D.linear(prev, P["gru.weight_hh"], P["gru.bias_hh"], False)
        has_edge = torch.from_numpy(pl["cnt"].sum(1) > 0)
        Hc, _ = D.gru_gates(gi, gh, prev, has_edge)
        stats.append(dict(colours_in=pl["ncls"], message_pairs=nm, colours_out=pl["ncls_next"], rows=R, message_rows=g["U"]))
    h = Hc[torch.from_numpy(plans[-1]["nxt"])] if plans else torch.cat([x, torch.zeros(R, H - Fn, dtype=dtype)], 1)
    # readout exactly as tests.ref_dataflow.forward
    hx = torch.cat([h, x], 1)
    att = D.mlp_fwd(P, "gather.att_nn", hx)[-1]
    emb = D.mlp_fwd(P, "gather.emb_nn", h)[-1]
    gemb, _ = D.gather_readout(att, emb, T["cidx"], T["node_mask"], B, N, cfg["big_positive"])
    add1 = D.mlp_fwd(P, "APDReadout.fAddNet1", h)[-1]
    conn1 = D.mlp_fwd(P, "APDReadout.fConnNet1", h)[-1]
    c = T["cidx"].view(B, N).long()
    cat_add = torch.cat([add1[c].reshape(B, N * A), gemb], 1)
    cat_conn = torch.cat([conn1[c].reshape(B, N * C), gemb], 1)
    out = torch.cat([D.mlp_fwd(P, "APDReadout.fAddNet2", cat_add)[-1], D.mlp_fwd(P, "APDReadout.fConnNet2", cat_conn)[-1],
                     D.mlp_fwd(P, "APDReadout.fTermNet2", gemb)[-1]], 1)
    return out, stats
