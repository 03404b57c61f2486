#!/bin/bash
# On the GPU box: raw rocprofv3 kernel traces (start / end of every launch, queue ids) of the training step for
# tools/critical_path.py and tools/gemm_class_report.py — default schedule, every launch alone on the device
# (bench.py --one-stream), ZINC shape, AttentionGGNN / ChEMBL shape; the last ~6 steps of each are kept.  Plus the
# GI_GEMM_LOG launch log of the one-stream run (GI_TRACE_ALL=1: the ZINC / ChEMBL shapes too).  Output -> gpurun_out/<tag>/ ; copy into profiles/<tag>/.
TAG=${1:-r03}
OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-forward-only --no-probe --no-one-stream --steps 12 --warmup 4"
: > $OUT/trace_summary.txt
tr() {  # tag, bench args...
  local tag=$1; shift
  rm -rf /tmp/tr_$tag
  timeout 100 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o t -- $B "$@" > /tmp/tr_$tag.log 2>&1
  python3 - /tmp/tr_$tag/t_kernel_trace.csv $OUT/trace_$tag.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "compact_count_kernel" in r["Kernel_Name"]]
lo = idx[-8] if len(idx) >= 8 else 0
keep = ["Kernel_Name", "Queue_Id", "Stream_Id", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count"]
keep = [k for k in keep if k in rows[0]]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=keep)
w.writeheader()
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:]:
    r = {k: r[k] for k in keep}
    r["Start_Timestamp"] = int(r["Start_Timestamp"]) - t0; r["End_Timestamp"] = int(r["End_Timestamp"]) - t0
    r["Kernel_Name"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    w.writerow(r)
print(sys.argv[2], len(rows) - lo, "rows")
PY
  grep "^{\"metric\"" /tmp/tr_$tag.log | python3 -c "import json,sys; d=json.load(sys.stdin); print('$tag', d['ms_per_step'], 'ms per step under rocprofv3 --kernel-trace')" >> $OUT/trace_summary.txt 2>&1
}
tr default
tr onestream --one-stream
[ -n "$GI_TRACE_ALL" ] && tr zinc --shape zinc --batch 1000 --model ggnn
[ -n "$GI_TRACE_ALL" ] && tr chembl --shape chembl --batch 250 --model attggnn
rm -f $OUT/gemm_launch_log.txt
GI_GEMM_LOG=$OUT/gemm_launch_log.txt timeout 60 $B --one-stream --steps 2 --warmup 1 > /dev/null 2>&1
cd /root/repo
python3 tools/critical_path.py $OUT/trace_default.csv > $OUT/critical_path_default.txt 2>&1
python3 tools/critical_path.py $OUT/trace_onestream.csv > $OUT/critical_path_onestream.txt 2>&1
python3 tools/gemm_class_report.py $OUT/trace_onestream.csv $OUT/gemm_launch_log.txt > $OUT/gemm_class_report.txt 2>&1
cat $OUT/trace_summary.txt; tail -30 $OUT/gemm_class_report.txt; head -30 $OUT/critical_path_default.txt
