#!/bin/bash
# GPU box: unit tests of the new kernels, chain micro-benchmark (tile depth 16 / 32), step A/B, PMC.
OUT=/root/repo/gpurun_out/run3; mkdir -p $OUT; cd /root/repo
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "not bench_batch and not full_size and not fixture" 2>&1 | tail -15) > $OUT/tests.log
for kt in 16 32; do
  echo "== GI_CHAIN_KT=$kt" >> $OUT/micro.txt
  GI_CHAIN_KT=$kt python tools/bench_chain.py both 8400 >> $OUT/micro.txt 2>&1
  GI_CHAIN_KT=$kt python tools/bench_chain.py both 26000 >> $OUT/micro.txt 2>&1
done
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe"
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  GI_CHAIN=$1 GI_GRU_FUSED=$2 $B 2>/dev/null | tail -1 > $OUT/bench_chain$1_gru$2.json
done
GI_CHAIN=1 GI_GRU_FUSED=1 GI_CHAIN_KT=32 $B 2>/dev/null | tail -1 > $OUT/bench_chain1_gru1_kt32.json
python3 - $OUT <<'PY' > $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "ms  fwd", d["forward_only"]["ms_per_step"], "ms  launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "frac", r["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
bash tools/pmc_kernel.sh gi_chain_kernel $OUT/pmc_chain_kt16.txt -- env GI_CHAIN_KT=16 python /root/repo/tools/bench_chain.py both 8400 > /dev/null 2>&1
cat $OUT/tests.log $OUT/micro.txt $OUT/summary.txt $OUT/pmc_chain_kt16.txt
