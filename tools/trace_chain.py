"""Per-workgroup timeline of one gi_mlp_chain launch on the headline message-row shapes: where each
row block ran (XCD / CU) and for how long (gi_mlp_chain_config's trace buffer)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphinvent_amd import lib as L  # noqa: E402
from tools import bench_chain  # noqa: E402

trace = torch.zeros(16 * 4096, dtype=torch.int64, device="cuda")
L.check(L.load().gi_mlp_chain_config(0, -1, 2, trace.data_ptr()), "gi_mlp_chain_config")

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
