"""Would two half-batches on two streams beat one batch on one stream?  (Fixed per-launch latency of
one half hiding behind the other half's main loops.)  Same total work: 1000 graphs per 'step'."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from graphinvent_amd import ops
from graphinvent_amd.gnn import mpnn
from graphinvent_amd.loss import apd_kl_loss
from graphinvent_amd.optim import FusedAdam

cfg, constants = bench.workload_constants("cuda")
torch.manual_seed(0)
model = mpnn.GGNN(constants).cuda().train()
opt = FusedAdam(model.parameters(), lr=1e-4)
batches = bench.make_batches(0, torch.device("cuda"))
halves = [[tuple(t[:500].contiguous() for t in b), tuple(t[500:].contiguous() for t in b)] for b in batches]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def full_step(i):
    b = batches[i % 4]
    ops.prefetch_compact(*batches[(i + 1) % 4][:2])
    out = model(b[0], b[1])
    for p in model.parameters(): p.grad = None
    apd_kl_loss(out, b[2]).backward()
    opt.step()

def half_step(i):
    h = halves[i % 4]
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    for p in model.parameters(): p.grad = None
    outs = []
    for s, part in ((sA, h[0]), (sB, h[1])):
        with torch.cuda.stream(s):
            outs.append((s, model(part[0], part[1]), part[2]))
    for s, out, tgt in outs:
        with torch.cuda.stream(s):
            (0.5 * apd_kl_loss(out, tgt)).backward()      # second backward accumulates into .grad
    cur.wait_stream(sA); cur.wait_stream(sB)
    opt.step()

for name, fn in (("one batch of 1000, one stream", full_step), ("two halves of 500, two streams", half_step)):
    for i in range(5): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(30): fn(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host {1e3 * (t1 - t0) / 30:.3f} ms, total {1e3 * (t2 - t0) / 30:.3f} ms per 1000 graphs")
