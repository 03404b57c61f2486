#!/bin/bash
cd /root/repo
for c in "fwd3 1 1 0" "fwd3a 1 1 0" "dgrad3 1 1 0" "dgrad3a 1 1 0" "fwd13a 1 1 0"; do timeout 60 tools/gemm_lab $c 2>&1 | tail -1; done
GI_LAB_FILL=0 timeout 60 tools/gemm_lab fwd3a 1 1 0 2>&1 | tail -1
