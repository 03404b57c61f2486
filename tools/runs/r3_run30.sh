#!/bin/bash
# final state of the round: suite (bf16x3 launches on = default), model tests with GI_BF3=0, traces + reports, profile collection
OUT=/root/repo/gpurun_out/r3_run30; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension" | tail -40) > $OUT/gpu_suite_final.log; tail -4 $OUT/gpu_suite_final.log | cut -c1-200
(GI_BF3=0 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py tests/test_golden_shapes_gpu.py -m gpu -q 2>&1 | tail -3) > $OUT/gpu_model_tests_fp32_mfma_only.log; tail -2 $OUT/gpu_model_tests_fp32_mfma_only.log
if grep -q " passed" $OUT/gpu_suite_final.log && ! grep -q "failed\|Aborted\|error" $OUT/gpu_suite_final.log; then
  bash tools/collect_traces.sh r03 2>&1 | tail -45
  bash tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1; tail -2 gpurun_out/r03/collect.log
  cat gpurun_out/r03/traffic.json | head -8
fi
