"""Host cost of the individual pieces of one step (no GPU sync inside the timed calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from graphinvent_amd import ops
from graphinvent_amd.gnn import mpnn
from graphinvent_amd.loss import apd_kl_loss
from graphinvent_amd.optim import FusedAdam

cfg, constants = bench.workload_constants("cuda")
model = mpnn.GGNN(constants).cuda().train()
params = list(model.parameters())
b = bench.make_batches(0, torch.device("cuda"))[0]
opt = FusedAdam(model.parameters(), lr=1e-4)
sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=5000)
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
N = 40
for it in range(N + 5):
    if it == 5:
        T.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ops.prefetch_compact(b[0], b[1]); tick("prefetch_compact", t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); out, tape = mpnn.ggnn_forward_raw(constants, b[0], b[1], params); tick("forward_raw (compact_fill + C forward + python)", t0)
    torch.cuda.synchronize()
    o = out.detach().requires_grad_(True)
    t0 = time.perf_counter(); loss = apd_kl_loss(o, b[2]); tick("fused KL fwd", t0)
    t0 = time.perf_counter(); loss.backward(); tick("fused KL bwd (autograd)", t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); grads, gflat = mpnn.ggnn_backward_raw(tape, out, o.grad, params); tick("backward_raw (C backward + python)", t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); out2 = model(b[0], b[1]); tick("model.forward via autograd.Function", t0)
    torch.cuda.synchronize()
    l2 = apd_kl_loss(out2, b[2])
    t0 = time.perf_counter(); opt.zero_grad(set_to_none=True); tick("zero_grad", t0)
    t0 = time.perf_counter(); l2.backward(); tick("loss.backward via autograd", t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); opt.step(); tick("FusedAdam.step", t0)
    t0 = time.perf_counter(); sched.step(); tick("OneCycleLR.step", t0)
for k, v in T.items():
    print(f"{k:55s} {1e3 * v / N:7.3f} ms")
