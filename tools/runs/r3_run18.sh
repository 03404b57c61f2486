#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run18; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 300 python tools/runs/dbg_graph.py 2>&1 | grep -v "^Extension\|amdgpu.ids" | tee $OUT/dbg.txt | cut -c1-200 | grep -v "ref|=0 |refc" 
echo ---
timeout 300 python -m pytest tests/test_syncfree_gpu.py -q -x 2>&1 | tail -3
for d in 0 7; do GI_DBG_COMPACT=$d timeout 300 python -m pytest tests/test_syncfree_gpu.py -q -x -k capturable 2>&1 | tail -2; done
