"""PCIe-inclusive training rate: the same step as bench.py, but every batch comes from HOST memory through
graphinvent_amd.loader.BlockStreamLoader (block-wise: a block of 10 000 int8 rows is read into pinned memory
by a background thread while the previous one is consumed -> vectorised gather -> async H2D one batch ahead
-> graph compaction of that batch on the copy stream -> int8 tensors straight into the model and the fused
loss).  BASELINE config 2 shapes, synthetic 40 000-graph dataset; prints the resident-input rate of the same
process next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from graphinvent_amd import dp, synthetic
from graphinvent_amd.loader import ArraySource, BlockStreamLoader
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
loader = BlockStreamLoader(ArraySource(nodes, edges, apds), B, block_size=10000, seed=0, device="cuda")
print(f"pinned host memory: {loader.pinned_bytes / 1e6:.1f} MB (2 blocks + 2 staging batches) for a {(nodes.nbytes + edges.nbytes + apds.nbytes) / 1e6:.0f} MB dataset")
for epoch in range(3):
    loader.set_epoch(epoch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for nb, eb, ab in loader:
        loss = tr.step(nb, eb, ab)
        n += nb.shape[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"epoch {epoch}: {n} graphs in {dt * 1e3:.1f} ms -> {n / dt:,.0f} graphs/s, {dt / (n / B) * 1e3:.3f} ms/step, loss {float(loss):.4f}")

# the same step with the inputs already resident in HBM (what bench.py times), same process, same model
res = [tuple(torch.from_numpy(x[i * B:(i + 1) * B]).cuda() for x in (nodes, edges, apds)) for i in range(4)]
from graphinvent_amd import ops
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40):
        ops.prefetch_compact(*res[(i + 1) % 4][:2])
        tr.step(*res[i % 4])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"resident inputs (int8): {40 * B / dt:,.0f} graphs/s, {dt / 40 * 1e3:.3f} ms/step")

if os.environ.get("GI_PROFILE_LOADER"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for nb, eb, ab in loader:
        tr.step(nb, eb, ab)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
