#!/bin/bash
# Raw rocprofv3 kernel traces (start / end of every launch, queue ids) of the training step for offline
# critical-path analysis: default, weight gradients on the main stream (uncontended kernel durations),
# ZINC shape, AttentionGGNN / ChEMBL shape.  The last ~6 steps of each trace are kept.
OUT=/root/repo/gpurun_out/run41; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-forward-only --no-probe --no-one-stream --steps 12 --warmup 4"
tr() {  # tag, env..., -- args
  local tag=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  rm -rf /tmp/tr_$tag
  env "${envs[@]}" timeout 100 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o t -- $B "$@" > /tmp/tr_$tag.log 2>&1
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
  grep "^{\"metric\"" /tmp/tr_$tag.log | python3 -c "import json,sys; d=json.load(sys.stdin); print('$tag', d['ms_per_step'])" >> $OUT/summary.txt 2>&1
}
tr default --
tr onestream GI_WGRAD_SIDE_STREAM=0 --
tr zinc -- --shape zinc --batch 1000 --model ggnn
tr chembl -- --shape chembl --batch 250 --model attggnn
GI_GEMM_LOG=$OUT/gemm_log_default.txt timeout 60 $B --steps 2 --warmup 1 > /dev/null 2>&1
ls -la $OUT; cat $OUT/summary.txt
