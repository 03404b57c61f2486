"""debug: do graph replays clobber eager tensors allocated after the capture?"""
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
with torch.no_grad():
    m(nodes, edges); torch.cuda.synchronize()
    print("before capture: allocated %.1f MB reserved %.1f MB" % (torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(nodes, edges)
    cg = m._last_bounded_graph
    print("after capture: allocated %.1f MB reserved %.1f MB" % (torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6))
    print("graph tensors: gfix %x +%d, gvar %x +%d, cmat %x +%d, out %x +%d" % (cg.gfix.data_ptr(), cg.gfix.nbytes, cg.gvar.data_ptr(), cg.gvar.nbytes, cg.cmat.data_ptr(), cg.cmat.nbytes, out.data_ptr(), out.nbytes))
    for seed in (2, 3, 1, 2):
        nb = synthetic.make_batch(B, **sh, seed=seed)
        nk, ek = dev(nb[0], nb[1])
        extra = [torch.full((n,), 7.0, device="cuda") for n in (1 << 20, 1 << 22, 1 << 24, 3 << 20)]     # eager tensors of assorted sizes
        s0 = (float(nk.sum()), float(ek.sum()))
        nodes.copy_(nk); edges.copy_(ek)
        g.replay(); torch.cuda.synchronize()
        s1 = (float(nk.sum()), float(ek.sum()))
        bad = [int((e != 7.0).sum()) for e in extra]
        print(f"seed {seed}: nk {nk.data_ptr():x} ek {ek.data_ptr():x} sums before {s0} after {s1} clobbered extra elements {bad} ptrs {[hex(e.data_ptr()) for e in extra]}")
        m.sync_free = False
        ref = m(nk, ek); m(nk, ek)
        m.sync_free = True
        m(nk, ek)
        print("   |got-ref|", float((out - ref).abs().max()))
