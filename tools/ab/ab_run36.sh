#!/bin/bash
# where does the weight-gradient side stream stop paying?  (rows S grow with the batch)
OUT=/root/repo/gpurun_out/run36; mkdir -p $OUT; cd /root/repo
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 12"
for cfg in "gdb13 2000 ggnn" "gdb13 3000 ggnn" "zinc 500 ggnn" "zinc 700 ggnn" "chembl 500 attggnn"; do
  set -- $cfg
  for v in 1 0; do
    GI_WGRAD_SIDE_STREAM=$v $B --shape $1 --batch $2 --model $3 2>/dev/null | tail -1 > $OUT/bench_$1_$2_side$v.json
  done
done
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms", "frac", r["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/summary.txt
