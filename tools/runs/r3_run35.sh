#!/bin/bash
cd /root/repo
for b in gemm_lab gemm_lab_noslp; do for c in "fwd3 1 1 0" "dgrad3 1 1 0" "fwd13 1 1 0"; do echo -n "$b "; timeout 60 tools/$b $c 2>&1 | tail -1; done; done
