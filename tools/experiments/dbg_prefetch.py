import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
from graphinvent_amd import ops, synthetic, lib as L
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
cfg = O.make_config(device="cuda")
model = mpnn.GGNN(O.as_constants(cfg)); model.load_state_dict(O.init_params(cfg, seed=2)); model = model.cuda().eval()
n8, e8, _ = synthetic.make_batch(64, **synthetic.SHAPES["gdb13"], seed=21)
nodes = torch.from_numpy(n8).float().cuda(); edges = torch.from_numpy(e8).float().cuda()
with torch.no_grad():
    ref = model(nodes, edges); torch.cuda.synchronize()
    ref2 = model(nodes, edges); torch.cuda.synchronize()
    print("plain repeat equal", torch.equal(ref, ref2))
    ops.prefetch_compact(nodes, edges); torch.cuda.synchronize()
    out = model(nodes, edges); torch.cuda.synchronize()
    print("prefetched equal (with syncs)", torch.equal(ref, out), float((ref-out).abs().max()))
    params = list(model.parameters())
    ops.prefetch_compact(nodes, edges); torch.cuda.synchronize()
    o2, (dims, graph, ws) = mpnn.ggnn_forward_raw(model.constants, nodes, edges, params)
    o1, (dims1, graph1, ws1) = mpnn.ggnn_forward_raw(model.constants, nodes, edges, params)
    torch.cuda.synchronize()
    print("raw equal", torch.equal(o1, o2), "gvar equal", torch.equal(graph.gvar, graph1.gvar), "gfix equal", torch.equal(graph.gfix, graph1.gfix))
    R = graph.S + 1
    for name in ("hx", "m", "agg", "gi", "gh"):
        a = ops.ws_view(ws, dims, graph.S, graph.E, graph.U, name, R if name != "m" else graph.U, 0)
        b = ops.ws_view(ws1, dims1, graph.S, graph.E, graph.U, name, R if name != "m" else graph.U, 0)
        print(name, torch.equal(a, b), float((a-b).abs().max()))
