#!/bin/bash
cd /root/repo
for c in "wgrad 1 1 0" "wgrad 1 2 0" "wgrad 2 2 0" "wgrad 1 1 11" "wgrad 1 2 11" "wgrad 2 2 11"; do timeout 60 tools/gemm_lab $c 2>&1 | tail -1; done
