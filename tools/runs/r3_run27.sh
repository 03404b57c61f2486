#!/bin/bash
cd /root/repo
for c in "fwd 1 1 0" "fwd3 1 1 0" "dgrad 1 1 0" "dgrad3 1 1 0" "fwd1 1 1 0" "fwd13 1 1 0" "tier2 1 1 0" "tier23 1 1 0"; do timeout 60 tools/gemm_lab $c 2>&1 | tail -2; done
