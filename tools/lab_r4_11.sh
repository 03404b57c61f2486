#!/bin/bash
# round 4: fp16x2 with amax cells (64 slots per tensor instead of one word): kernel tests, A/B, trace
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "fp16x2 or bf16x3" 2>&1 | tail -8 > $O/pytest_k.txt
tools/ab.sh -r 2 -o $O/ab_head "x2" "bf16x3 GI_X2=0" > $O/ab_head.txt 2>&1
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "x2" "bf16x3 GI_X2=0" > $O/ab_zinc.txt 2>&1
tools/collect_traces.sh r4k > $O/traces.txt 2>&1
cat $O/pytest_k.txt $O/ab_head.txt $O/ab_zinc.txt
