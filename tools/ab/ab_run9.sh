#!/bin/bash
OUT=/root/repo/gpurun_out/run9; mkdir -p $OUT; cd /root/repo
(timeout 300 python -m pytest "tests/test_model_gpu.py::test_full_size_parity_vs_fp32_and_fp64_oracle" -q -x 2>&1 | grep -E "assert|Error|hip_l2|worst|passed|failed" | head -20) > $OUT/tests.log
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only"
for rep in 1 2; do for v in 0 1; do
  GI_WGRAD_TILE2=$v $B 2>/dev/null | tail -1 > $OUT/bench_tile2_${v}_$rep.json
done; done
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/tests.log $OUT/summary.txt
