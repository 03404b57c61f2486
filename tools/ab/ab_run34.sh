#!/bin/bash
# split-K for the skinny long-reduction layers of the graph-level stacks
OUT=/root/repo/gpurun_out/run34; mkdir -p $OUT; cd /root/repo
(timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_attggnn_gpu.py tests/test_model_gpu.py tests/test_dropout_gpu.py -q -x 2>&1 | grep -E "passed|failed|error|Error|FAILED|assert" | tail -8) > $OUT/tests.log
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream"
for rep in 1 2; do
  $B 2>/dev/null | tail -1 > $OUT/bench_gdb13_$rep.json
  $B --shape chembl --batch 250 --model attggnn --steps 15 2>/dev/null | tail -1 > $OUT/bench_chembl_$rep.json
  $B --shape zinc --batch 1000 --steps 15 2>/dev/null | tail -1 > $OUT/bench_zinc_$rep.json
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
cat $OUT/tests.log $OUT/summary.txt
