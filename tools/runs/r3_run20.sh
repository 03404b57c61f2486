#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run20; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for h in 0 1; do HOLD=$h timeout 300 python tools/runs/dbg_graph7.py 2>&1 | grep -v "^Extension\|amdgpu.ids"; done | tee $OUT/dbg.txt | cut -c1-400
