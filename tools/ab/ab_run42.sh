#!/bin/bash
# PMC passes of the default bench command with GI_FUSE_DEFAULT = 15 (the part of tools/collect_profiles.sh that
# fits the GPU minutes left in the round): MFMA busy, FETCH_SIZE, WRITE_SIZE, L2 hit / fabric requests.
OUT=/root/repo/gpurun_out/run42; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  timeout 100 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-extra-configs --no-one-stream --no-forward-only > /tmp/pmc_$name.log 2>&1
  python3 - "$name" <<'PY' > $OUT/pmc_$name.txt
import csv, collections, glob, sys, re
f = glob.glob("/tmp/pmc_%s/*counter_collection.csv" % sys.argv[1])
if not f: print("no counter file"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    m = re.search(r"(gi_gemm_batch_kernel<[^>]*>|gi_gemm_kernel<[^>]*>|gi_chain_kernel<[^>]*>|\w+_kernel)", k)
    key = m.group(1) if m else k[:40]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
print("per-dispatch averages (rocprofv3 --pmc %s), kernel: {counter: avg} dispatches" % sys.argv[1])
for key in sorted(agg, key=lambda k: -sum(agg[k].values())):
    print(key, {c: round(v / n[key][c], 1) for c, v in agg[key].items()}, max(n[key].values()))
PY
  echo "$(date +%H:%M:%S) $name done" >> $OUT/summary.txt
done
ls -la $OUT
