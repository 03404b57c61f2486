#!/bin/bash
OUT=/root/repo/gpurun_out/run12; mkdir -p $OUT; cd /root/repo
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "not bench_batch and not full_size and not fixture" 2>&1 | tail -6) > $OUT/tests.log
python tools/trace_chain.py 8400 2>&1 | grep -v amdgpu.ids | head -8 > $OUT/micro.txt
python tools/bench_chain.py both 8400 2>&1 | grep -v amdgpu.ids >> $OUT/micro.txt
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe"
for rep in 1 2; do for v in 0 1; do
  GI_CHAIN_XROWS=$v $B 2>/dev/null | tail -1 > $OUT/bench_xrows${v}_$rep.json
done; done
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms fwd", d["forward_only"]["ms_per_step"], "launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/tests.log $OUT/micro.txt $OUT/summary.txt
