#!/bin/bash
OUT=/root/repo/gpurun_out/run18; mkdir -p $OUT; cd /root/repo
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only"
for rep in 1 2; do for v in 0 1; do
  GI_CHAIN_EXCLUSIVE=$v $B 2>/dev/null | tail -1 > $OUT/bench_excl${v}_$rep.json
done; done
for v in 0 1; do GI_CHAIN_EXCLUSIVE=$v $B --model attggnn --shape chembl --batch 250 --steps 20 2>/dev/null | tail -1 > $OUT/bench_chembl_excl$v.json; done
python3 - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["ms_per_step"], "ms launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
PY
