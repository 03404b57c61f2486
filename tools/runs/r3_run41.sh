#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run41; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k bf16x3 2>&1 | grep "passed\|failed\|Error" | head -3
bash tools/ab.sh -r 2 -o $OUT/ab "fp32 GI_BF3=0" "bf3 GI_BF3=1" | tail -4
for b in 0 1; do GI_BF3=$b CACHE=1 python tools/runs/fwd_probe.py 2>&1 | tail -1; done
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension" | tail -40) > $OUT/gpu_suite_final.log; grep "passed\|failed" $OUT/gpu_suite_final.log
