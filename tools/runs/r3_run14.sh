#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run14; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for d in 0 7 1 2 4; do GI_DBG_COMPACT=$d timeout 300 python tools/runs/dbg_graph2.py 2>&1 | grep -v "^Extension\|amdgpu.ids" | tail -2; done | tee $OUT/dbg.txt | cut -c1-400
