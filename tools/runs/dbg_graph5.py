"""debug: the failing flow of dbg_graph.py case 1 with input integrity checks"""
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
for trial in range(2):
    m = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda"))); m.load_state_dict(P); m = m.cuda().eval()
    m.cache_pass0 = False; m.sync_free = True
    b0 = synthetic.make_batch(B, **sh, seed=1)
    nodes, edges = dev(b0[0], b0[1])
    with torch.no_grad():
        m(nodes, edges); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(nodes, edges)
        for seed in (2, 3, 1, 2):
            nb = synthetic.make_batch(B, **sh, seed=seed)
            nk, ek = dev(nb[0], nb[1])
            nodes.copy_(nk); edges.copy_(ek)
            g.replay(); torch.cuda.synchronize()
            got = out.clone()
            in_ok = (torch.equal(nodes, nk), torch.equal(edges, ek))
            err = ops.bounded_error(m._last_bounded_graph)
            m.sync_free = False
            ref = m(nk, ek)
            refc = m(nk, ek)
            m.sync_free = True
            eager = m(nk, ek)
            nk2, ek2 = dev(nb[0], nb[1])
            m.sync_free = False
            ref2 = m(nk2, ek2)
            m.sync_free = True
            g.replay(); torch.cuda.synchronize()
            print(f"trial {trial} seed {seed}: inputs intact {in_ok} {torch.equal(nk, nk2)} {torch.equal(ek, ek2)} |got-ref|={float((got-ref).abs().max()):.3g} "
                  f"|got-ref2|={float((got-ref2).abs().max()):.3g} |ref-ref2|={float((ref-ref2).abs().max()):.3g} |replay again-ref2|={float((out-ref2).abs().max()):.3g} "
                  f"stats={m.pass0_cache_stats()}", flush=True)
