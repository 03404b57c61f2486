"""Diagnostic (GPU box): per-parameter gradient error of the HIP path and of the fp32 oracle, both
measured against the fp64 oracle, on BASELINE config-2-shaped input."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphinvent_amd import synthetic
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
shape = sys.argv[2] if len(sys.argv) > 2 else "gdb13"
sh = synthetic.SHAPES[shape]
over = dict(hidden_node_features=128, message_size=128) if shape == "gdb13" else {}
cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"], **over)
P = O.init_params(cfg, seed=4)
n8, e8, a8 = synthetic.make_batch(B, **sh, seed=6)
keep = np.nonzero(e8.reshape(B, -1).any(1))[0]
n8, e8, a8 = n8[keep], e8[keep], a8[keep]
model = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda"))); model.load_state_dict(P); model = model.cuda().train()
nodes, edges, tgt = (torch.from_numpy(x).float().cuda() for x in (n8, e8, a8))
out = model(nodes, edges); loss = O.kl_loss(out, tgt); loss.backward()
gh = {k: p.grad.double().cpu() for k, p in model.named_parameters()}
t = lambda x, dt: torch.from_numpy(x).to(dt)
t0 = time.time()
o32, l32, g32 = O.forward_backward(P, cfg, t(n8, torch.float32), t(e8, torch.float32), t(a8, torch.float32))
t1 = time.time()
o64, l64, g64 = O.forward_backward({k: v.double() for k, v in P.items()}, cfg, t(n8, torch.float64), t(e8, torch.float64), t(a8, torch.float64))
print(f"oracle fp32 {t1-t0:.2f}s fp64 {time.time()-t1:.2f}s threads {torch.get_num_threads()}")
rel = lambda a, b: float((a.double() - b.double()).abs().max() / max(float(b.abs().max()), 1e-30))
print("logits: hip-vs-64 %.2e  o32-vs-64 %.2e" % (rel(out.cpu(), o64), rel(o32, o64)))
print("loss  : hip %.8f o32 %.8f o64 %.8f" % (float(loss), float(l32), float(l64)))
for k in g64:
    print("%-40s hip-vs-64 %.2e   o32-vs-64 %.2e   hip-vs-o32 %.2e  max|g| %.2e" % (k, rel(gh[k], g64[k]), rel(g32[k], g64[k]), rel(gh[k], g32[k]), float(g64[k].abs().max())))
