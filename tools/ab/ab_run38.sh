#!/bin/bash
# 64-row chain workgroups (<2, 2>): tests under a timeout, then A/B at the shapes with several rounds of row blocks
OUT=/root/repo/gpurun_out/run38; mkdir -p $OUT; cd /root/repo
(timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "chain" 2>&1 | grep -E "passed|failed|error|Error|FAILED|assert" | tail -8) > $OUT/tests.log
cat $OUT/tests.log
grep -q failed $OUT/tests.log && exit 0
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 12"
for cfg in "zinc 1000 ggnn" "chembl 250 attggnn" "gdb13 1000 ggnn" "gdb13 3000 ggnn"; do
  set -- $cfg
  for v in 0 -1; do
    GI_CHAIN_ROWS64=$v timeout 120 $B --shape $1 --batch $2 --model $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(\"$1 $2 rows64=$v:\", d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"config\"][\"loss\"])"
  done
done 2>&1 | tee $OUT/summary.txt
