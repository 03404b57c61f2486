#!/bin/bash
OUT=/root/repo/gpurun_out/run17; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --steps 6 --warmup 3"
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o bench -- $B > $OUT/log.txt 2>&1
python /root/repo/tools/timeline.py $OUT/tr/*kernel_trace.csv > $OUT/timeline.txt 2>&1
rm -rf $OUT/tr
tail -150 $OUT/timeline.txt
