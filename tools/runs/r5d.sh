cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5d; mkdir -p $O
A="--shape chembl --model attggnn --batch 250 --steps 10 --warmup 3"
tools/ab.sh -r 1 -o /root/repo/$O/abc -a "$A" "new" "norecord GI_DBG_NORECORD=1" "main GI_PREPACK=0" "nox2chain GI_CHAIN_X2=0" > $O/abc.log 2>&1; cat $O/abc/summary.txt
python - <<'P'
import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from graphinvent_amd import ops
dev = torch.device('cuda', 0)
wl = bench.Workload('chembl', 'attggnn', 250, 0, dev, 100)
for i in range(3): wl.run_step()
torch.cuda.synchronize()
for i in range(6):
    s0 = torch.cuda.memory_stats()
    t0 = time.perf_counter(); wl.run_step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    print("step host %.2f ms, total %.2f ms; reserved %.2f GB, new segments %d, retries %d" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, s1['reserved_bytes.all.current'] / 1e9, s1['segment.all.allocated'] - s0['segment.all.allocated'], s1['num_alloc_retries'] - s0['num_alloc_retries']))
P
