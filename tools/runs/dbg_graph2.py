"""debug: captured bounded forward replays vs eager (cache off), B=256"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphinvent_amd import ops, synthetic
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
sh = synthetic.SHAPES["gdb13"]
cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"])
P = O.init_params(cfg, seed=3, model="GGNN")
def dev(*a): return [torch.from_numpy(np.ascontiguousarray(x)).float().cuda() for x in a]
B = 256
m = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda"))); m.load_state_dict(P); m = m.cuda().eval()
m.cache_pass0 = False; m.sync_free = True
b0 = synthetic.make_batch(B, **sh, seed=1)
nodes, edges = dev(b0[0], b0[1])
res = []
with torch.no_grad():
    m(nodes, edges); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(nodes, edges)
    for seed in (2, 3, 1, 2, 3, 3):
        nb = synthetic.make_batch(B, **sh, seed=seed)
        nk, ek = dev(nb[0], nb[1])
        nodes.copy_(nk); edges.copy_(ek)
        g.replay(); torch.cuda.synchronize()
        got = out.clone()
        gr = m._last_bounded_graph
        c = gr.layout.counts
        counts = gr.gfix[c:c + 24].tolist()
        m.sync_free = False
        ref = m(nk, ek)
        m.sync_free = True
        res.append(f"{float((got-ref).abs().max()):.3g} S,E,U,D0={counts[0]},{counts[1]},{counts[3]},{counts[20]}")
print("GI_DBG_COMPACT", os.environ.get("GI_DBG_COMPACT"), " | ".join(res))
