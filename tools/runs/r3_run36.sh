#!/bin/bash
# final state of the round: full GPU suite, traces + reports, profile collection
OUT=/root/repo/gpurun_out/r3_run36; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension" | tail -40) > $OUT/gpu_suite_final.log; grep "passed\|failed" $OUT/gpu_suite_final.log
if grep -q " passed" $OUT/gpu_suite_final.log && ! grep -q "failed\|Aborted\|error" $OUT/gpu_suite_final.log; then
  bash tools/collect_traces.sh r03 2>&1 | tail -14
  bash tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1; tail -2 gpurun_out/r03/collect.log
fi
