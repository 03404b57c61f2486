#!/bin/bash
cd /root/repo
bash tools/ab.sh -r 3 -o gpurun_out/r3_run32 "hold" "nohold GI_DBG_HOLD=0" | tail -7
bash tools/ab.sh -r 1 -o gpurun_out/r3_run32z -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "zhold" "znohold GI_DBG_HOLD=0" | tail -3
