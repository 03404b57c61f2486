"""PCIe-inclusive training rate: the same step as bench.py, but every batch comes from HOST memory
through graphinvent_amd.loader.ShardedBlockLoader (int8 block in pinned memory -> vectorised gather
-> async H2D one batch ahead -> graph compaction of that batch on the copy stream -> int8 tensors
straight into the model and the fused loss).  BASELINE config 2 shapes, synthetic block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from graphinvent_amd import dp, synthetic
from graphinvent_amd.loader import ShardedBlockLoader
from graphinvent_amd.gnn import mpnn
from graphinvent_amd.loss import apd_kl_loss
from graphinvent_amd.optim import FusedAdam

N_GRAPHS, B = 40000, 1000
sh = synthetic.SHAPES["gdb13"]
parts = [synthetic.make_batch(4000, **sh, seed=s) for s in range(N_GRAPHS // 4000)]
nodes, edges, apds = (np.concatenate([p[i] for p in parts]) for i in range(3))
print(f"block: {N_GRAPHS} graphs, {(nodes.nbytes + edges.nbytes + apds.nbytes) / N_GRAPHS:.0f} bytes/graph (int8)")
cfg, constants = bench.workload_constants("cuda")
torch.manual_seed(0)
model = mpnn.GGNN(constants).cuda().train()
opt = FusedAdam(model.parameters(), lr=1e-4)
tr = dp.DataParallel(model, opt, loss_fn=apd_kl_loss)
loader = ShardedBlockLoader(nodes, edges, apds, B, seed=0, device="cuda")
for epoch in range(3):
    loader.set_epoch(epoch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for nb, eb, ab in loader:
        loss = tr.step(nb, eb, ab)
        n += nb.shape[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"epoch {epoch}: {n} graphs in {dt * 1e3:.1f} ms -> {n / dt:,.0f} graphs/s, {dt / (n / B) * 1e3:.3f} ms/step, loss {float(loss):.4f}")

if os.environ.get("GI_PROFILE_LOADER"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for nb, eb, ab in loader:
        tr.step(nb, eb, ab)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
