#!/bin/bash
cd /root/repo
bash tools/ab.sh -r 4 -o gpurun_out/r3_run47 -a "--steps 30" "f32wt" "planes GI_DBG_DGRAD_PLANES=1" | tail -9
