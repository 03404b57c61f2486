#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run31; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(GI_BF3=0 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py tests/test_golden_shapes_gpu.py -m gpu -q 2>&1 | grep "passed\|failed") | tee $OUT/gpu_model_tests_fp32_mfma_only.log
bash tools/collect_traces.sh r03 2>&1 | tail -48
bash tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1; tail -2 gpurun_out/r03/collect.log
cat gpurun_out/r03/traffic.json | head -8
