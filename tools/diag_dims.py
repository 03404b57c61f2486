"""Diagnostic (GPU box): one case of tests/test_dims_gpu.py stage by stage against the CPU dataflow model
(tests/ref_dataflow.py), then logits / loss / gradients against the fp32 oracle.  python tools/diag_dims.py <case> [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphinvent_amd import lib as L, ops
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O
from tests import ref_dataflow as D
from tests.test_dims_gpu import _case, _set_mode
from tests.test_model_gpu import make_model, to_dev, fully_masked_rows

name = sys.argv[1]; mode = sys.argv[2] if len(sys.argv) > 2 else "fp16x2"
cfg, (n8, e8, a8) = _case(name)
P = O.init_params(cfg, seed=31)
lib = L.load(); _set_mode(lib, mode)
rel = lambda a, b: float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max() / max(float(torch.as_tensor(b).double().abs().max()), 1e-30))
ref_out, tape = D.forward(P, cfg, torch.from_numpy(n8).float(), torch.from_numpy(e8).float(), keep=True)
model = make_model(cfg, P)
nodes, edges, tgt = to_dev(n8, e8, a8)
params = list(model.parameters())
out, (dims, graph, ws) = mpnn.ggnn_forward_raw(model.constants, nodes, edges, params)
S, E, U = graph.S, graph.E, graph.U
print(name, mode, "S E U D0", S, E, U, graph.D0, "ref", tape["g"]["S"], tape["g"]["E"], tape["g"]["U"], tape["g"]["D0"])
R, B = S + 1, n8.shape[0]
H, M, G, Fn = dims.H, dims.M, dims.G, dims.Fn
view = lambda nm, rows, i=0, j=0: ops.ws_view(ws, dims, graph, nm, rows, i, j)
for p, ps in enumerate(tape["passes"]):
    rows = graph.D0 if ps["p0"] else U
    print(f"pass {p}: hx {rel(view('hx', R, p)[:, :H], ps['h_prev']):.2e}", end=" ")
    for l in range(dims.enn_depth):
        want = torch.cat([ps["acts_t"][t][l] for t in range(dims.Fe)], 0)
        print(f"eact{l} {rel(view('eact', rows, p, l)[:, :dims.enn_hidden], want):.2e}", end=" ")
    print(f"m {rel(view('m', rows, p)[:, :M], ps['m']):.2e} agg {rel(view('agg', R, p)[:, :M], ps['agg']):.2e}")
hxP = view("hx", R, dims.passes)
print(f"h {rel(hxP[:, :H], tape['h']):.2e} en {rel(view('en', R)[:, :G], tape['att_acts'][-1]):.2e} emb {rel(view('emb', R)[:, :G], tape['emb_acts'][-1]):.2e} "
      f"add1 {rel(view('add1', R)[:, :dims.A], tape['add1'][-1]):.2e} conn1 {rel(view('conn1', R)[:, :dims.C], tape['conn1'][-1]):.2e}")
live = np.setdiff1d(np.arange(B), fully_masked_rows(e8))
print(f"logits live {rel(out[live], ref_out[live]):.2e} all {rel(out, ref_out):.2e}")
from tests import pins
signs = pins.signs_from_hip(dims, graph, ws, out, attn=False)
mask_pin = pins.mask_pin_from_hip(dims, graph, ws, n8.shape[0], cfg["big_positive"])
garr = pins.graph_arrays(graph)
o_leaf = out.detach().clone().requires_grad_(True)
loss = O.kl_loss(o_leaf, tgt); loss.backward()
grads, _ = mpnn.ggnn_backward_raw((dims, graph, ws), out, o_leaf.grad, params)
t = lambda x: torch.from_numpy(x).float()
o32, l32, g32, flipped, total = pins.oracle_pinned(O, P, cfg, t(n8), t(e8), t(a8), signs, garr, 'GGNN', mask_pin=mask_pin)
print('pinned: flipped', flipped, 'of', total)
print(f"loss hip {float(loss):.6f} oracle {float(l32):.6f}; logits vs oracle {rel(out, o32):.2e}")
for (k, _), g in zip(model.named_parameters(), grads):
    e = rel(g, g32[k])
    if e > 3e-5: print(f"  grad {k:45s} {e:.2e}  max|g| {float(g32[k].abs().max()):.2e}")
