#!/bin/bash
OUT=/root/repo/gpurun_out/run20; mkdir -p $OUT; cd /root/repo
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_attggnn_gpu.py -q -x 2>&1 | tail -15) > $OUT/tests.log
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --model attggnn --steps 20"
for v in 0 1; do GI_ATT_PASS0=$v $B --shape chembl --batch 250 2>/dev/null | tail -1 > $OUT/bench_chembl_p0_$v.json; GI_ATT_PASS0=$v $B --shape gdb13 --batch 1000 2>/dev/null | tail -1 > $OUT/bench_gdb13att_p0_$v.json; done
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/tests.log $OUT/summary.txt
