#!/bin/bash
OUT=/root/repo/gpurun_out/run11; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for cfg in "attggnn chembl 250" "ggnn zinc 1000"; do
  set -- $cfg
  B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --steps 10 --warmup 3 --model $1 --shape $2 --batch $3"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$2 -o bench -- $B > $OUT/$2.log 2>&1
  rm -f $OUT/$2/*kernel_trace.csv
  grep '^{"metric"' $OUT/$2.log | cut -c1-400
  head -16 $OUT/$2/*kernel_stats.csv | cut -c1-170
done
