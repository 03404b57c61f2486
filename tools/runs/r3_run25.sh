#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run25; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python -m pytest tests/test_syncfree_gpu.py tests/test_p0cache_gpu.py -m gpu -q -x 2>&1 | tail -1; done
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension" | tail -40) > $OUT/gpu_suite_final.log; tail -5 $OUT/gpu_suite_final.log | cut -c1-200
if grep -q " passed" $OUT/gpu_suite_final.log && ! grep -q "failed\|Aborted\|error" $OUT/gpu_suite_final.log; then
  bash tools/collect_profiles.sh r03 > $OUT/collect.log 2>&1; tail -30 $OUT/collect.log
fi
