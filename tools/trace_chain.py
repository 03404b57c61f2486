"""Per-workgroup timeline of one gi_mlp_chain launch on the headline message-row shapes: where each
row block ran (XCD / CU) and for how long (GI_CHAIN_TRACE measurement aid in gi_chain.hip)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
trace = torch.zeros(16 * 4096, dtype=torch.int64, device="cuda")
os.environ["GI_CHAIN_TRACE"] = str(trace.data_ptr())
from tools import bench_chain  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 8400
bench_chain.main(U=U, reps=1)
torch.cuda.synchronize()
t = trace.cpu().view(-1, 16)
t = t[t[:, 1] > 0]
t0 = int(t[:, 0].min())
start, end = (t[:, 0] - t0).float() / 100.0, (t[:, 1] - t0).float() / 100.0     # us (100 MHz)
dur = end - start
hw, xcc = t[:, 2] >> 8, t[:, 2] & 0xff
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
print(f"workgroups {t.shape[0]}  kernel span {float(end.max()):.1f} us  duration min/mean/max "
      f"{float(dur.min()):.1f}/{float(dur.mean()):.1f}/{float(dur.max()):.1f} us")
print("started after 5 us:", int((start > 5).sum()), " distinct CUs:", len(set(cu.tolist())))
per_cu = collections.Counter(cu.tolist())
print("workgroups per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
late = (start > 5).nonzero().flatten().tolist()[:12]
for i in late:
    print(f"  wg {i}: start {float(start[i]):.1f} end {float(end[i]):.1f} rows {int(t[i, 3])} cu {int(cu[i]):#x}")
q = torch.tensor([0.1, 0.5, 0.9])
print("end-time quantiles (us):", [round(float(x), 1) for x in torch.quantile(end, q)])

ph = (t[:, 4:4 + 7] - t[:, 0:1]).float() / 100.0
first = (start < 5)
print("phase end times of the first-round workgroups, mean us (prologue, then after each layer):",
      [round(float(x), 1) for x in ph[first].mean(0)])

# ---- fused GRU kernel phases on the same shapes ---------------------------------------------------
from graphinvent_amd import ops, synthetic
import numpy as np
n8, e8, _ = synthetic.make_batch(1000, **synthetic.SHAPES["gdb13"], seed=1)
g, _ = ops.compact(torch.from_numpy(n8).float().cuda(), torch.from_numpy(e8).float().cuda(), 128)
R, H, M = g.S + 1, 128, 128
gt = torch.zeros(4 * 1024, dtype=torch.int64, device="cuda")
os.environ["GI_GRU_TRACE"] = str(gt.data_ptr())
m = torch.randn(g.U, M, device="cuda"); hx = torch.randn(R, 136, device="cuda")
W_ih = torch.randn(3 * H, M, device="cuda") / 11; W_hh = torch.randn(3 * H, H, device="cuda") / 11
b = torch.randn(3 * H, device="cuda")
agg = torch.empty(R, M, device="cuda"); hn = torch.empty(R, 136, device="cuda")
gi = torch.empty(R, 3 * H, device="cuda"); gh = torch.empty(R, 3 * H, device="cuda")
for _ in range(3):
    ops.gru_fused_fwd(m, g.in_perm, g.seg_off, agg, False, hx, hn, W_ih, W_hh, b, b, gi, gh, R, H, M)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gru_fused_fwd(m, g.in_perm, g.seg_off, agg, False, hx, hn, W_ih, W_hh, b, b, gi, gh, R, H, M)
e1.record(); torch.cuda.synchronize()
x = gt.cpu().view(-1, 4); x = x[x[:, 3] > 0]
d = (x[:, 1:] - x[:, 0:1]).float() / 100.0
print(f"gru fused: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch, {x.shape[0]} workgroups; mean us since "
      f"workgroup start at: prologue done {float(d[:, 0].mean()):.1f}, main loop done {float(d[:, 1].mean()):.1f}, "
      f"end {float(d[:, 2].mean()):.1f}; start spread {float((x[:, 0] - x[:, 0].min()).max()) / 100:.1f} us")
