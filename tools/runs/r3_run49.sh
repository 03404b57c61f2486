#!/bin/bash
cd /root/repo
for m in 2048 3000 4096 5500; do for c in "fwd 1 1 0" "fwd3f 1 1 0" "dgrad 1 1 0" "dgrad3f 1 1 0"; do echo -n "M=$m "; GI_LAB_M=$m timeout 60 tools/gemm_lab $c 2>&1 | tail -1 | cut -c1-110; done; done
