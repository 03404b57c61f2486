#!/bin/bash
# workgroups per weight-gradient problem (slab count), re-measured on the steady-loop GEMM
OUT=/root/repo/gpurun_out/run28; mkdir -p $OUT; cd /root/repo
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream"
run() { name=$1; shift; for rep in 1 2; do env "$@" $B 2>/dev/null | tail -1 > $OUT/bench_${name}_$rep.json; done; }
run w256 GI_WGRAD_WGS=256
run w128 GI_WGRAD_WGS=128
run w192 GI_WGRAD_WGS=192
run w384 GI_WGRAD_WGS=384
run w096 GI_WGRAD_WGS=96
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/summary.txt
