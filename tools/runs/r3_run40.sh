#!/bin/bash
cd /root/repo
for r in 0 1; do for c in "fwd3f 1 1 0" "dgrad3 1 1 0"; do echo -n "remap=$r "; GI_LAB_B3_REMAP=$r timeout 60 tools/gemm_lab $c 2>&1 | tail -1; done; done
