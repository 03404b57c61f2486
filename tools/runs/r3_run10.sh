#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run10; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for t in test_bounded_forward_equals_the_ordinary_forward test_generation_style_loop_never_reads_back test_bounded_forward_is_capturable_as_one_hip_graph test_bound_violations_are_flagged_not_fatal; do
  (timeout 300 python -m pytest tests/test_syncfree_gpu.py -q -x -k $t 2>&1 | grep -v "^  File\|^Extension" | tail -25) > $OUT/$t.log; echo "== $t"; tail -12 $OUT/$t.log | cut -c1-220
done
cd /tmp
for mode in sync_free blocking; do
  MODE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$mode -o p -- python /root/repo/tools/runs/gen_probe.py 2>&1 | grep "ms/round"
  python - <<PY
import csv,glob
f=glob.glob("$OUT/prof_$mode/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("$mode total kernel ms per round", tot/23/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print("  %-70s calls %5s avg %8.1f us total/round %7.1f us"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/23/1e3))
PY
  rm -f $OUT/prof_$mode/*kernel_trace.csv
done
