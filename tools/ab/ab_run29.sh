#!/bin/bash
# chain kernel vs layer-by-layer message stacks at the larger shapes, after the GEMM steady-state loop
OUT=/root/repo/gpurun_out/run29; mkdir -p $OUT; cd /root/repo
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 15"
for v in 1 0; do
  GI_CHAIN=$v $B --shape zinc --batch 1000 2>/dev/null | tail -1 > $OUT/bench_zinc_chain$v.json
  GI_CHAIN=$v $B --shape chembl --batch 250 --model attggnn 2>/dev/null | tail -1 > $OUT/bench_chembl_chain$v.json
  GI_CHAIN=$v $B --shape chembl --batch 250 2>/dev/null | tail -1 > $OUT/bench_chemblggnn_chain$v.json
done
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
