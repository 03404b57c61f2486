#!/bin/bash
OUT=/root/repo/gpurun_out/run16; mkdir -p $OUT; cd /root/repo
for cfg in "attggnn chembl 250" "ggnn zinc 1000"; do
  set -- $cfg
  for v in 0 1; do
    B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --steps 20 --warmup 5 --model $1 --shape $2 --batch $3"
    GI_CHAIN_XROWS=$v $B 2>/dev/null | tail -1 > $OUT/bench_$2_x$v.json
  done
done
python3 - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["ms_per_step"], "ms launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
PY
