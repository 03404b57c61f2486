"""debug: which buffer of a wrong replay differs first from an eager bounded forward of the same batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphinvent_amd import ops, synthetic, lib as L
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
sh = synthetic.SHAPES["gdb13"]
cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"])
P = O.init_params(cfg, seed=3, model="GGNN")
def dev(*a): return [torch.from_numpy(np.ascontiguousarray(x)).float().cuda() for x in a]
B = 256
m = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda"))); m.load_state_dict(P); m = m.cuda().eval()
m.cache_pass0 = False
params = m._params()
bounds = ops.default_bounds(B, 13, 3)
def raw(n, e):
    return mpnn.ggnn_forward_raw(m.constants, n, e, params, m._KIND, None, bounds, None)
b0 = synthetic.make_batch(B, **sh, seed=1)
nodes, edges = dev(b0[0], b0[1])
def views(tape, out):
    dims, graph, ws = tape
    lay = graph.layout
    cnt = graph.gfix[lay.counts:lay.counts + 24].tolist()
    S, E, U, D0 = cnt[0], cnt[1], cnt[3], cnt[20]
    R = S + 1
    v = {"counts": graph.gfix[lay.counts:lay.counts + 24], "gfix_fixed": graph.gfix[:lay.scratch], "gfix_all": graph.gfix,
         "cmat": graph.cmat[:R, :D0]}
    offs = graph._offs
    names = ["u_src", "in_perm", "mu_off", "mu_dst", "mu_slot", "out_perm", "d_src"]
    lens = [U, E, U + 1, E, E, U, D0]
    for nm, o, ln in zip(names, offs[:7], lens):
        v[nm] = graph.gvar[o:o + ln]
    for p in range(4):
        v[f"hx{p}"] = ops.ws_view(ws, dims, graph, "hx", R, p)
    for p in range(3):
        v[f"m{p}"] = ops.ws_view(ws, dims, graph, "m", D0 if p == 0 else U, p)[:, :100]
        v[f"agg{p}"] = ops.ws_view(ws, dims, graph, "agg", R, p)[:, :100]
        v[f"gi{p}"] = ops.ws_view(ws, dims, graph, "gi", R, p)[:, :300]
        v[f"gh{p}"] = ops.ws_view(ws, dims, graph, "gh", R, p)[:, :300]
    for nm in ("en", "emb", "add1", "conn1"):
        v[nm] = ops.ws_view(ws, dims, graph, nm, R)
    for nm in ("cat_add", "cat_conn", "gemb"):
        v[nm] = ops.ws_view(ws, dims, graph, nm, B)
    v["out"] = out
    return v, (S, E, U, D0)
with torch.no_grad():
    m.sync_free = True
    raw(nodes, edges); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, tape = raw(nodes, edges)
    for seed in (2, 3, 1, 2):
        nb = synthetic.make_batch(B, **sh, seed=seed)
        nk, ek = dev(nb[0], nb[1])
        nodes.copy_(nk); edges.copy_(ek)
        g.replay(); torch.cuda.synchronize()
        gv, sizes = views(tape, out)
        gv = {k: t.clone() for k, t in gv.items()}
        eout, etape = raw(nk, ek)
        torch.cuda.synchronize()
        ev, esizes = views(etape, eout)
        line = [f"seed {seed} sizes replay {sizes} eager {esizes}:"]
        for k in gv:
            a, b_ = gv[k], ev[k]
            if a.shape != b_.shape:
                line.append(f"{k}: SHAPE {tuple(a.shape)} vs {tuple(b_.shape)}"); continue
            if k == "gfix_all":
                ne = (a != b_).nonzero().flatten(); 
                if len(ne): line.append(f"gfix_all: {len(ne)} ints differ, first at {int(ne[0])} (scratch starts {tape[1].layout.scratch})")
                continue
            d = (a.float() - b_.float()).abs().max().item() if a.numel() else 0.0
            if d != 0: line.append(f"{k}: {d:.3g}")
        print(" ".join(line), flush=True)
        m.sync_free = False
        m(nk, ek); m(nk, ek)
        m.sync_free = True
        m(nk, ek)
