#!/bin/bash
# Runs on the GPU box: A/B of the resident-activation chain kernel (GI_CHAIN=0/1) on the headline
# bench + a rocprofv3 kernel-stats pass with it on.  Output -> gpurun_out/<tag>/
TAG=${1:-ab_chain}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-probe"
for rep in 1 2; do
  for v in 0 1; do
    GI_CHAIN=$v $B 2>/dev/null | tail -1 > $OUT/bench_chain${v}_$rep.json
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B --steps 20 > $OUT/rocprof.log 2>&1
rm -f $OUT/stats/*kernel_trace.csv
python3 - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_chain*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms  fwd", d["forward_only"]["ms_per_step"], "ms  launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
