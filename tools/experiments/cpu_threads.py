import sys; sys.path.insert(0, "/root/repo")
import bench
cfg, _ = bench.workload_constants("cpu")
for t in (4, 8, 12, 16, 24):
    r = bench.cpu_baseline(cfg, t, timed_steps=2)
    print(t, r["value"], flush=True)
