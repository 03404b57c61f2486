#!/bin/bash
cd /root/repo
for f in 1 0; do for c in "fwd3 1 1 0" "fwd13 1 1 0" "fwd 1 1 0"; do echo -n "fill=$f "; GI_LAB_FILL=$f timeout 60 tools/gemm_lab $c 2>&1 | tail -1; done; done
