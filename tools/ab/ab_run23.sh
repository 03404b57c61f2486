#!/bin/bash
# weight-gradient side stream: off / lowest priority (default) / confined to a CU subset
OUT=/root/repo/gpurun_out/run23; mkdir -p $OUT; cd /root/repo
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only"
run() { name=$1; shift; for rep in 1 2; do env "$@" $B 2>/dev/null | tail -1 > $OUT/bench_${name}_$rep.json; done; }
run base GI_NOP=1
run noside GI_WGRAD_SIDE_STREAM=0
run m55 GI_SIDE_CU_MASK=0x55555555
run m77 GI_SIDE_CU_MASK=0x77777777
run m0f GI_SIDE_CU_MASK=0x0f0f0f0f
run m3f GI_SIDE_CU_MASK=0x3f3f3f3f
run mff00 GI_SIDE_CU_MASK=0xffff0000
run mall GI_SIDE_CU_MASK=0xffffffff
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
