"""debug: captured bounded forward replays vs eager, cache on/off, warm-up count"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphinvent_amd import ops, synthetic
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
sh = synthetic.SHAPES["gdb13"]
cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"])
P = O.init_params(cfg, seed=3, model="GGNN")
def dev(*a): return [torch.from_numpy(np.ascontiguousarray(x)).float().cuda() for x in a]
for cache in (False, True):
    for B, nwarm, f32 in ((256, 1, True), (256, 2, True), (256, 1, False), (1000, 1, True)):
        m = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda"))); m.load_state_dict(P); m = m.cuda().eval()
        m.cache_pass0 = cache; m.sync_free = True
        b0 = synthetic.make_batch(B, **sh, seed=1)
        nodes, edges = dev(b0[0], b0[1]) if f32 else [torch.from_numpy(x).cuda() for x in b0[:2]]
        with torch.no_grad():
            for _ in range(nwarm): m(nodes, edges)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = m(nodes, edges)
            for seed in (2, 3, 1, 2):
                nb = synthetic.make_batch(B, **sh, seed=seed)
                nk, ek = dev(nb[0], nb[1]) if f32 else [torch.from_numpy(x).cuda() for x in nb[:2]]
                nodes.copy_(nk); edges.copy_(ek)
                g.replay(); torch.cuda.synchronize()
                got = out.clone()
                err = ops.bounded_error(m._last_bounded_graph)
                m.sync_free = False; m.cache_pass0 = False
                ref = m(nk, ek)
                m.cache_pass0 = cache
                refc = m(nk, ek)
                m.sync_free = True
                eager = m(nk, ek)
                print(f"cache={cache} B={B} warm={nwarm} f32={f32} seed={seed}: err={err} |got-ref|={float((got-ref).abs().max()):.3g} "
                      f"|refc-ref|={float((refc-ref).abs().max()):.3g} |eager_bounded-ref|={float((eager-ref).abs().max()):.3g} stats={m.pass0_cache_stats()}", flush=True)
            if B == 1000:
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(20): g.replay()
                torch.cuda.synchronize(); print("  replay ms", (time.perf_counter() - t0) / 20 * 1e3)
