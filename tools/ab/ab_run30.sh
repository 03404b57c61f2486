#!/bin/bash
# in-kernel slab reduction (GI_WGRAD_REDUCE=1): parity + determinism tests, then the step time
OUT=/root/repo/gpurun_out/run30; mkdir -p $OUT; cd /root/repo
(GI_WGRAD_REDUCE=1 timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py tests/test_dp_gpu.py tests/test_dropout_gpu.py -q -x 2>&1 | grep -E "passed|failed|error|Error|FAILED|assert" | tail -12) > $OUT/tests.log
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream"
for rep in 1 2; do for v in 0 1; do GI_WGRAD_REDUCE=$v $B 2>/dev/null | tail -1 > $OUT/bench_red${v}_$rep.json; done; done
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"], "loss", d["config"]["loss"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/tests.log $OUT/summary.txt
