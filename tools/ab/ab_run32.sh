#!/bin/bash
# kernel breakdown of the AttentionGGNN / ChEMBL-shape and the ZINC-shape steps
OUT=/root/repo/gpurun_out/run32; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for cfg in "chembl 250 attggnn" "zinc 1000 ggnn"; do
  set -- $cfg
  rm -rf /tmp/st_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$1 -o b -- python /root/repo/bench.py --shape $1 --batch $2 --model $3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream > /tmp/st_$1.log 2>&1
  cp /tmp/st_$1/b_kernel_stats.csv $OUT/kernel_stats_$1.csv
  python3 /root/repo/tools/timeline.py /tmp/st_$1/b_kernel_trace.csv > $OUT/timeline_$1.txt 2>&1
done
ls -la $OUT
