#!/bin/bash
# knobs re-measured on the steady-loop GEMM + a fresh step timeline
OUT=/root/repo/gpurun_out/run22; mkdir -p $OUT; cd /root/repo
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only"
run() { name=$1; shift; for rep in 1 2; do env "$@" $B 2>/dev/null | tail -1 > $OUT/bench_${name}_$rep.json; done; }
run base GI_NOP=1
run chain0 GI_CHAIN=0
run remap0 GI_GEMM_XCD_REMAP=0
run remap700 GI_GEMM_XCD_REMAP=700
run remap2500 GI_GEMM_XCD_REMAP=2500
run excl GI_CHAIN_EXCLUSIVE=1
run gru GI_GRU_FUSED=1
run wt GI_DGRAD_WT=1
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only > /tmp/tl.log 2>&1
python3 /root/repo/tools/timeline.py $(ls /tmp/tl/*kernel_trace.csv | head -1) > $OUT/timeline.txt 2>&1
cat $OUT/summary.txt
