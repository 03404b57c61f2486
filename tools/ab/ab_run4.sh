#!/bin/bash
OUT=/root/repo/gpurun_out/run4; mkdir -p $OUT; cd /root/repo
for kt in 16 32; do echo "== KT=$kt" >> $OUT/trace.txt; GI_CHAIN_KT=$kt python tools/trace_chain.py 8400 >> $OUT/trace.txt 2>&1; done
cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-probe --steps 20"
GI_CHAIN=1 GI_GRU_FUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B > $OUT/rocprof.log 2>&1
rm -f $OUT/stats/*kernel_trace.csv
cat $OUT/trace.txt; head -14 $OUT/stats/*kernel_stats.csv | cut -c1-150
