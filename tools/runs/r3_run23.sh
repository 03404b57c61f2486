#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run23; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
run() { timeout 300 python tools/runs/dbg_graph8.py 2>&1 | grep -v "^Extension\|amdgpu.ids" | tail -1 | cut -c1-220; }
echo base; CACHE=0 run
echo THRASH=1; CACHE=0 THRASH=1 run
echo THRASH=2; CACHE=0 THRASH=2 run
echo DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; CACHE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run
echo DEBUG_HIP_GRAPH_DOT_PRINT; CACHE=0 HIP_GRAPH_... 2>/dev/null
echo AMD_SERIALIZE_KERNEL=3; CACHE=0 AMD_SERIALIZE_KERNEL=3 run
echo HSA_ENABLE_SDMA=0; CACHE=0 HSA_ENABLE_SDMA=0 run
