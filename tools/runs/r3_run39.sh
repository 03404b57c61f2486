#!/bin/bash
cd /root/repo
for c in "fwd3 1 1 0" "fwd3f 1 1 0" "fwd13 1 1 0" "fwd13f 1 1 0"; do timeout 60 tools/gemm_lab $c 2>&1 | tail -1; done
GI_LAB_FILL=0 timeout 60 tools/gemm_lab fwd3f 1 1 0 2>&1 | tail -1
