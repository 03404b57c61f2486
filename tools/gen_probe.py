"""generation-loop probe: MODE=sync_free|blocking python tools/gen_probe.py  (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from graphinvent_amd import ops
from graphinvent_amd.gnn import mpnn
from graphinvent_amd.sampler import sample_actions_raw
cfg, constants = bench.workload_constants("cuda")
torch.manual_seed(0)
model = mpnn.GGNN(constants).cuda().eval()
batches = bench.make_batches(0, "cuda")
nodes, edges = batches[0][0].clone(), batches[0][1].clone()
model.sync_free = os.environ.get("MODE", "sync_free") == "sync_free"
model.cache_pass0 = os.environ.get("CACHE", "1") == "1"
A = cfg["len_f_add_per_node"]
with torch.no_grad():
    for i in range(int(os.environ.get("ROUNDS", "23"))):
        if i == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        src = batches[i % 4]
        nodes.copy_(src[0]); edges.copy_(src[1])
        logits = model(nodes, edges)
        n_nodes = (nodes.sum(2) != 0).sum(1).int()
        sample_actions_raw(logits, n_nodes, edges, A)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(os.environ.get("MODE"), "CACHE", os.environ.get("CACHE", "1"), "ms/round", dt / (int(os.environ.get("ROUNDS", "23")) - 3) * 1e3, ops.READBACKS)
