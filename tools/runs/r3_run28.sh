#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run28; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k bf16x3 2>&1 | tail -5
(GI_BF3=1 timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File\|^Extension" | tail -30) > $OUT/suite_bf3.log; tail -6 $OUT/suite_bf3.log | cut -c1-250
bash tools/ab.sh -r 2 -o $OUT/ab "fp32 GI_BF3=0" "bf3 GI_BF3=1" | tail -5
bash tools/ab.sh -r 1 -o $OUT/abz -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "zinc_fp32 GI_BF3=0" "zinc_bf3 GI_BF3=1" | tail -3
bash tools/ab.sh -r 1 -o $OUT/abc -a "--shape chembl --batch 250 --model attggnn --steps 10 --warmup 3" "chembl_fp32 GI_BF3=0" "chembl_bf3 GI_BF3=1" | tail -3
for b in 0 1; do GI_BF3=$b CACHE=1 python tools/runs/fwd_probe.py 2>&1 | tail -1; done
