#!/bin/bash
# final state: suite, lab check of the dgrad fp32 form, traces + reports, profile collection
OUT=/root/repo/gpurun_out/r3_run44; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in "dgrad3 1 1 0" "dgrad3f 1 1 0"; do timeout 60 tools/gemm_lab $c 2>&1 | tail -1; done | tee $OUT/lab.txt
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension" | tail -40) > $OUT/gpu_suite_final.log; grep "passed\|failed" $OUT/gpu_suite_final.log
if grep -q " passed" $OUT/gpu_suite_final.log && ! grep -q "failed\|Aborted\|error" $OUT/gpu_suite_final.log; then
  bash tools/collect_traces.sh r03 2>&1 | grep "bf16x3\|ms per step\|wgrad \|gi_gemm /\|step (from"
  bash tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1; tail -1 gpurun_out/r03/collect.log
fi
