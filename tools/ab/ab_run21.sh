#!/bin/bash
# GEMM steady-loop (straight-line main loop) check: kernel tests + default bench twice + microbench
OUT=/root/repo/gpurun_out/run21; mkdir -p $OUT; cd /root/repo
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -5) > $OUT/tests.log
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream"
for rep in 1 2; do $B 2>/dev/null | tail -1 > $OUT/bench_$rep.json; done
python bench.py --no-cpu-baseline --no-probe 2>/dev/null | tail -1 > $OUT/bench_full.json
(timeout 300 python tools/bench_gemm.py 2>&1 | tail -40) > $OUT/bench_gemm.txt
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"], [ (e["config"]["workload"][:30], e["ms_per_step"]) for e in d.get("extra_configs", [])])
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/tests.log $OUT/summary.txt $OUT/bench_gemm.txt
