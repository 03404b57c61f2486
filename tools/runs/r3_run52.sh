#!/bin/bash
cd /root/repo
bash tools/ab.sh -r 2 -o gpurun_out/r3_run52z -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "zinc_fp32 GI_BF3=0" "zinc_bf3 GI_BF3=1" | tail -4
bash tools/ab.sh -r 2 -o gpurun_out/r3_run52c -a "--shape chembl --batch 250 --model attggnn --steps 10 --warmup 3" "chembl_fp32 GI_BF3=0" "chembl_bf3 GI_BF3=1" | tail -4
