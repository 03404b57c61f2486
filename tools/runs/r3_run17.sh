#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run19; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 300 python tools/runs/dbg_graph6.py 2>&1 | grep -v "^Extension\|amdgpu.ids" | tee $OUT/dbg.txt | cut -c1-400
