#!/bin/bash
OUT=/root/repo/gpurun_out/run5; mkdir -p $OUT; cd /root/repo
for kt in 32; do echo "== KT=$kt" >> $OUT/trace.txt; GI_CHAIN_KT=$kt python tools/trace_chain.py 8400 >> $OUT/trace.txt 2>&1; done
cat $OUT/trace.txt
