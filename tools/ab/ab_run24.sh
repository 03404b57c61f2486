#!/bin/bash
# early kick of the deferred weight gradients at the last pass + one-stream roofline leg
OUT=/root/repo/gpurun_out/run24; mkdir -p $OUT; cd /root/repo
(timeout 900 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py -q -x 2>&1 | tail -5) > $OUT/tests.log
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only"
for rep in 1 2; do $B 2>/dev/null | tail -1 > $OUT/bench_$rep.json; done
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"], r.get("one_stream"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/tests.log $OUT/summary.txt
