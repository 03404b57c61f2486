#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run15; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for d in 0 1 2 4 8 12; do MODE=$d timeout 300 python tools/runs/dbg_graph3.py 2>&1 | grep -v "^Extension\|amdgpu.ids" | tail -6; done | tee $OUT/dbg.txt | cut -c1-300
