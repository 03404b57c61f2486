#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run43; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in "dgrad3 1 1 0" "dgrad3f 1 1 0"; do timeout 60 tools/gemm_lab $c 2>&1 | tail -1; done | tee $OUT/lab.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k bf16x3 2>&1 | grep "passed\|failed\|Error" | head -3
bash tools/ab.sh -r 2 -o $OUT/ab "fp32 GI_BF3=0" "bf3 GI_BF3=1" | tail -4
