"""
Test infrastructure: feed the ORACLE the AlphaDropout masks the HIP path drew.

In training mode with dropout_p > 0 the reference applies ``torch.nn.AlphaDropout`` behind every
Linear + SELU (gnn/modules.py:130-142).  The HIP path draws its keep bits from a counter-based hash
of (seed, site id, row, column) — site id = 16 * (index of the layer's weight in the parameter table)
+ message pass; rows = message rows (one per edge, no row sharing in this mode), node slots, or
graphs.  ``gi_dropout_mask`` exports the bits of one site; this module maps them into the oracle's
layouts (every bond-type MLP on every edge in nonzero order; [B, N, .] over all padded slots;
AttentionGGNN: neighbour-padded [V, maxdeg, .]) and installs them as ``oracle.ggnn_oracle.DROPOUT_HOOK``,
so that the oracle's fp32 forward/backward and the HIP path see the SAME masks and can be compared at
the usual tolerances.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np
import torch

MAXP = 16      # gi_model.hip: site id = weight index * MAXP + pass


def export_mask(seed: int, site: int, p: float, rows: int, cols: int) -> torch.Tensor:
    """bool [rows, cols] keep bits of one site (gi_dropout_setup + gi_dropout_mask)."""
    from graphinvent_amd import lib as L
    lib = L.load()
    q = L.DropoutParams()
    L.check(lib.gi_dropout_setup(float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(site), C.byref(q)),
            "gi_dropout_setup")
    buf = torch.empty((rows, cols), dtype=torch.uint8, device="cuda")
    L.check(lib.gi_dropout_mask(C.byref(q), rows, cols, buf.data_ptr(), cols,
                                torch.cuda.current_stream().cuda_stream), "gi_dropout_mask")
    return buf.bool().cpu()


def family_p(cfg: dict, prefix: str, model: str) -> float:
    if prefix.startswith("msg_nns."):
        return cfg["msg_dropout_p"] if model == "AttGGNN" else cfg["enn_dropout_p"]
    if prefix.startswith("att_nns."):
        return cfg["att_dropout_p"]
    return {"gather.att_nn": cfg["gather_att_dropout_p"], "gather.emb_nn": cfg["gather_emb_dropout_p"],
            "APDReadout.fAddNet1": cfg["mlp1_dropout_p"], "APDReadout.fConnNet1": cfg["mlp1_dropout_p"],
            "APDReadout.fAddNet2": cfg["mlp2_dropout_p"], "APDReadout.fConnNet2": cfg["mlp2_dropout_p"],
            "APDReadout.fTermNet2": cfg["mlp2_dropout_p"]}[prefix]


class OracleDropout:
    """Callable for ``oracle.ggnn_oracle.DROPOUT_HOOK`` around ONE oracle forward.

    keys: state_dict keys in order (= the HIP parameter table); g: index arrays of the graph compacted
    WITHOUT row sharing (tests/pins.graph_arrays or tests/ref_dataflow.compact(nodedup=True))."""

    NODE = ("gather.att_nn", "gather.emb_nn", "APDReadout.fAddNet1", "APDReadout.fConnNet1")

    def __init__(self, cfg: dict, keys: List[str], seed: int, g: dict, nodes: np.ndarray,
                 edges: np.ndarray, model: str = "GGNN"):
        self.cfg, self.keys, self.seed, self.model = cfg, list(keys), seed, model
        B, N, _ = nodes.shape
        self.B, self.N = B, N
        adj = edges.sum(3) != 0
        eb, ei, ej = np.nonzero(adj)
        self.E = eb.size
        self.etype = edges[eb, ei, ej, :].argmax(1)
        assert int(g["U"]) == self.E and int(g["S"]) == B * N and int(g["D0"]) == 0
        self.edge_row = torch.from_numpy(np.asarray(g["in_perm"]).astype(np.int64))
        if model == "AttGGNN":
            deg = adj.sum(2)[adj.sum(2) > 0].astype(np.int64)
            self.node_of_edge = torch.from_numpy(np.repeat(np.arange(deg.size), deg))
            self.slot_in_node = torch.from_numpy(
                np.concatenate([np.arange(k) for k in deg]) if deg.size else np.zeros(0, np.int64))
        self.calls: Dict[tuple, int] = {}
        self.sites = 0
        self.kept = 0
        self.drawn = 0

    def __call__(self, prefix: str, layer: int, x: torch.Tensor):
        p = family_p(self.cfg, prefix, self.model)
        k = (prefix, layer)
        call = self.calls.get(k, 0)
        self.calls[k] = call + 1
        if p == 0:
            return None
        edge_level = prefix.startswith("msg_nns.") or prefix.startswith("att_nns.")
        base = prefix.split(".")[0] + ".0" if edge_level else prefix
        widx = self.keys.index(f"{base}.seq.{3 * layer}.weight")
        cols = x.shape[-1]
        if edge_level:
            keep = export_mask(self.seed, widx * MAXP + call, p, self.E, cols)[self.edge_row]
            t = int(prefix.split(".")[1])
            sel = torch.from_numpy(self.etype == t)
            if self.model == "AttGGNN":                  # x: [V, maxdeg, cols]
                mask = torch.ones(x.shape, dtype=torch.bool)
                idx = sel.nonzero(as_tuple=True)[0]
                mask[self.node_of_edge[idx], self.slot_in_node[idx]] = keep[idx]
            else:                                        # x: [E, cols]; other types' rows are gated to 0
                mask = torch.where(sel[:, None], keep, torch.ones_like(keep))
            self.kept += int(keep[sel].sum()); self.drawn += int(sel.sum()) * cols
        elif prefix in self.NODE:                        # x: [B, N, cols] over every padded slot
            keep = export_mask(self.seed, widx * MAXP, p, self.B * self.N + 1, cols)
            mask = keep[:self.B * self.N].view(self.B, self.N, cols)
            self.kept += int(mask.sum()); self.drawn += mask.numel()
        else:                                            # x: [B, cols]
            mask = export_mask(self.seed, widx * MAXP, p, self.B, cols)
            self.kept += int(mask.sum()); self.drawn += mask.numel()
        self.sites += 1
        return p, mask
