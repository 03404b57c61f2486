"""debug: which interleaving breaks replays.  MODE bits: 1 = hold the captured CompactGraph, 2 = synchronize before replay,
4 = skip the eager bounded forward, 8 = skip the eager unbounded forwards"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphinvent_amd import ops, synthetic
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
MODE = int(os.environ.get("MODE", "0"))
sh = synthetic.SHAPES["gdb13"]
cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"])
P = O.init_params(cfg, seed=3, model="GGNN")
def dev(*a): return [torch.from_numpy(np.ascontiguousarray(x)).float().cuda() for x in a]
B = 256
m = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda"))); m.load_state_dict(P); m = m.cuda().eval()
m.cache_pass0 = False
seeds = (2, 3, 1, 2, 3)
data = {s: dev(*synthetic.make_batch(B, **sh, seed=s)[:2]) for s in (1, 2, 3)}
with torch.no_grad():
    refs = {s: m(*data[s]).clone() for s in data}
    m.sync_free = True
    nodes, edges = data[1][0].clone(), data[1][1].clone()
    m(nodes, edges); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(nodes, edges)
    cg = m._last_bounded_graph if MODE & 1 else None
    res = []
    for seed in seeds:
        nk, ek = data[seed]
        nodes.copy_(nk); edges.copy_(ek)
        if MODE & 2: torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        got = out.clone()
        d = {s: float((got - refs[s]).abs().max()) for s in refs}
        cnt = ""
        if cg is not None:
            c = cg.layout.counts; cc = cg.gfix[c:c + 24].tolist(); cnt = f" S,E,U,D0,err={cc[0]},{cc[1]},{cc[3]},{cc[20]},{cc[2]}"
        res.append(f"seed{seed}: " + " ".join(f"d{s}={v:.3g}" for s, v in d.items()) + cnt)
        if not MODE & 8:
            m.sync_free = False
            m(nk, ek); m(nk, ek)
            m.sync_free = True
        if not MODE & 4:
            m(nk, ek)
print("MODE", MODE, "\n   " + "\n   ".join(res))
