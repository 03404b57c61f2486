#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of the default bench command + separate PMC passes
# (HBM traffic and MFMA utilisation of the GEMM family / seg_sum).  Output -> gpurun_out/<tag>/
TAG=${1:-r01}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BENCH="python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
rm -f $OUT/stats/*kernel_trace.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_$name.log 2>&1
  python3 - "$name" <<'PY' > $OUT/pmc_$name.txt
import csv, collections, glob, sys, re
f = glob.glob("/tmp/pmc_%s/*counter_collection.csv" % sys.argv[1])
if not f: print("no counter file"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    m = re.search(r"(gi_gemm_batch_kernel<[^>]*>|gi_gemm_kernel<[^>]*>|\w+_kernel)", k)
    key = m.group(1) if m else k[:40]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
print("per-dispatch averages (rocprofv3 --pmc %s), kernel: {counter: avg} dispatches" % sys.argv[1])
for key in sorted(agg, key=lambda k: -sum(agg[k].values())):
    print(key, {c: round(v / n[key][c], 1) for c, v in agg[key].items()}, max(n[key].values()))
PY
done
ls -la $OUT $OUT/stats
