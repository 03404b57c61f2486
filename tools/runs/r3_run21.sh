#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run21; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in 1 0; do for h in 0 1; do for t in 0 1; do CACHE=$c HOLD=$h TWICE=$t timeout 300 python tools/runs/dbg_graph8.py 2>&1 | grep -v "^Extension\|amdgpu.ids" | tail -1; done; done; done | tee $OUT/dbg.txt | cut -c1-300
