"""debug: the pytest flow (test_bounded_forward_is_capturable_as_one_hip_graph) with knobs.
CACHE=0|1, HOLD=0|1 (keep the captured forward's tape alive), TWICE=0|1 (replay twice, report both)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphinvent_amd import ops, synthetic
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
CACHE, HOLD, TWICE = (os.environ.get(k, "0") == "1" for k in ("CACHE", "HOLD", "TWICE"))
sh = synthetic.SHAPES["gdb13"]
cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"])
P = O.init_params(cfg, seed=3, model="GGNN")
def dev(*a): return [torch.from_numpy(np.ascontiguousarray(x)).float().cuda() for x in a]
B = 256
stash = []
if HOLD:
    orig = mpnn.ggnn_forward_raw
    def keep(*a, **k):
        r = orig(*a, **k); stash.append(r[1]); return r
    mpnn.ggnn_forward_raw = keep
m = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda"))); m.load_state_dict(P); m = m.cuda().eval()
truth = {}
with torch.no_grad():
    m.cache_pass0 = False
    for s in (1, 2, 3):
        nb = synthetic.make_batch(B, **sh, seed=s)
        truth[s] = m(*dev(nb[0], nb[1])).cpu()
    del stash[:]
    m.cache_pass0 = CACHE
    b0 = synthetic.make_batch(B, **sh, seed=1)
    nodes, edges = dev(b0[0], b0[1])
    m.sync_free = True
    m(nodes, edges); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = m(nodes, edges)
    res = []
    for seed in (2, 3, 1, 2):
        nb = synthetic.make_batch(B, **sh, seed=seed)
        nk, ek = dev(nb[0], nb[1])
        nodes.copy_(nk); edges.copy_(ek)
        if os.environ.get("THRASH", "0") == "1":
            junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda"); junk.fill_(1); del junk
        if os.environ.get("THRASH", "0") == "2":
            torch.cuda.synchronize()
            tiny = torch.zeros(4, device="cuda"); tiny += 1
        graph.replay(); torch.cuda.synchronize()
        got = out.clone()
        got2 = None
        if TWICE:
            graph.replay(); torch.cuda.synchronize(); got2 = out.clone()
        m.sync_free = False
        ref = m(nk, ek)
        m.sync_free = True
        t = truth[seed].cuda()
        res.append(f"s{seed}: got {float((got-t).abs().max()):.3g}" + (f" got2 {float((got2-t).abs().max()):.3g}" if TWICE else "") + f" ref {float((ref-t).abs().max()):.3g}")
print(f"CACHE={int(CACHE)} HOLD={int(HOLD)} TWICE={int(TWICE)}: " + " | ".join(res), m.pass0_cache_stats())
