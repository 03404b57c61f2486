#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of the default bench command + separate PMC passes
# (HBM traffic and MFMA utilisation of the GEMM family / seg_sum) + the default bench JSON line.
# Output -> gpurun_out/<tag>/ ; copy into profiles/<tag>/ afterwards.
TAG=${1:-r03}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BENCH="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-forward-only --no-probe --no-one-stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
rm -f $OUT/stats/*kernel_trace.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-extra-configs --no-one-stream --no-forward-only > /tmp/pmc_$name.log 2>&1
  python3 - "$name" <<'PY' > $OUT/pmc_$name.txt
import csv, collections, glob, sys, re
f = glob.glob("/tmp/pmc_%s/*counter_collection.csv" % sys.argv[1])
if not f: print("no counter file"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    m = re.search(r"(gi_gemm\w*_kernel<[^>]*>|gi_b3[pv]_kernel<[^>]*>|gi_chain\w*_kernel<[^>]*>|\w+_kernel)", k)
    key = m.group(1) if m else k[:40]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
print("per-dispatch averages (rocprofv3 --pmc %s), kernel: {counter: avg} dispatches" % sys.argv[1])
for key in sorted(agg, key=lambda k: -sum(agg[k].values())):
    print(key, {c: round(v / n[key][c], 1) for c, v in agg[key].items()}, max(n[key].values()))
PY
done
# HBM-side bytes per GEMM launch from the FETCH_SIZE / WRITE_SIZE passes (what bench.py reports as
# roofline.traffic): KB counters; FETCH x2 on gfx950 for 16 B/lane reads (MI355X_MICROARCH.md).
python3 - $OUT <<'PY'
import json, re, sys, ast
out = sys.argv[1]
def gemm_avg(path):
    tot = cnt = 0.0
    for line in open(path):
        m = re.match(r"(gi_(?:gemm|chain|b3p|b3v)\S*<[^>]*>) (\{.*\}) (\d+)$", line.strip())
        if not m: continue
        vals = ast.literal_eval(m.group(2)); k = int(m.group(3))
        tot += list(vals.values())[0] * k; cnt += k
    return (tot / cnt if cnt else 0.0), int(cnt)
f, nf = gemm_avg(out + "/pmc_FETCH_SIZE.txt"); w, nw = gemm_avg(out + "/pmc_WRITE_SIZE.txt")
json.dump({"kernel": "gi_gemm family (all dispatches of the profiled steps)", "dispatches": nf,
           "FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
           "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE taken as is",
           "hbm_side_bytes_per_launch": int(f * 1024 * 2 + w * 1024),
           "note": "fabric-side counters: Infinity-Cache hits are included; TCC_EA0_RDREQ (pmc_TCC_HIT_sum.txt) shows the real DRAM reads",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-extra-configs --no-one-stream --no-forward-only` (training steps only), tools/collect_profiles.sh"},
          open(out + "/traffic.json", "w"), indent=1)
PY
# the aggregation kernel beyond the Infinity Cache (567 MB of message rows): counter bytes behind the
# GB/s that bench.py derives from HIP-event time x algorithmic bytes (separate --pmc passes)
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmcp_$name
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmcp_$name -o p -- python /root/repo/bench.py --probe-only > /tmp/pmcp_$name.log 2>&1
  python3 - "$name" <<'PY' >> $OUT/pmc_seg_sum_probe.txt
import csv, collections, glob, sys
f = glob.glob("/tmp/pmcp_%s/*counter_collection.csv" % sys.argv[1])
if not f: print("no counter file"); sys.exit(0)
rows = [r for r in csv.DictReader(open(f[0])) if "seg_sum_kernel" in r["Kernel_Name"]]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for r in rows: agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("seg_sum_kernel, probe dispatches (rocprofv3 --pmc %s): per-dispatch average" % sys.argv[1],
      {k: round(v / n[k], 1) for k, v in agg.items()}, "dispatches", max(n.values()) if n else 0)
PY
done
grep aggregation_probe /tmp/pmcp_FETCH_SIZE.log >> $OUT/pmc_seg_sum_probe.txt
python /root/repo/bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json
ls -la $OUT $OUT/stats
