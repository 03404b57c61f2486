#!/bin/bash
OUT=/root/repo/gpurun_out/run14; mkdir -p $OUT; cd /root/repo
(timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "chain" 2>&1 | tail -3) > $OUT/tests.log
for v in 0 1; do echo "== XROWS=$v" >> $OUT/micro.txt; GI_CHAIN_XROWS=$v python tools/trace_chain.py 8400 2>&1 | grep -v amdgpu.ids | grep -v "gru fused" | grep -E "duration|phase|chain " >> $OUT/micro.txt; done
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
