#!/bin/bash
# round 3, GPU call 5: flattened k-tile stream (+ deferred epilogue) against the round-2 kernel; unit tests; step A/B
OUT=/root/repo/gpurun_out/r3_run5; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | tail -15) > $OUT/gemm_tests.log; tail -3 $OUT/gemm_tests.log
L=$OUT/lab.txt; : > $L
for cls in "fwd 1 1" "fwd 1 2" "fwd 2 2" "dgrad 1 1" "dgrad 1 2" "wgrad 1 1" "tier2 1 1"; do
  echo -n "old     " >> $L; timeout 60 tools/gemm_lab_old $cls 0 >> $L 2>&1
  for ps in 0 11; do
    echo -n "nodefer " >> $L; timeout 60 tools/gemm_lab_nodefer $cls $ps >> $L 2>&1
    echo -n "defer   " >> $L; timeout 60 tools/gemm_lab $cls $ps >> $L 2>&1
  done
done
timeout 60 tools/gemm_lab_trace fwd 1 1 11 $OUT/trace_fwd_11_p11.csv >> $L 2>&1
timeout 60 tools/gemm_lab_trace fwd 1 2 11 $OUT/trace_fwd_12_p11.csv >> $L 2>&1
cat $L
S=$OUT/summary.txt; : > $S
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5"
run() {
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 90 $B "$@" 2>$OUT/err.txt | tail -1 | python -c "
import json, sys
try:
    d = json.load(sys.stdin); print('$label:', d['ms_per_step'], 'frac', d['roofline']['frac'], 'loss', d['config']['loss'])
except Exception as e:
    print('$label: FAILED', repr(e), open('$OUT/err.txt').read()[-600:])" >> $S 2>&1
}
for rep in 1 2; do
  run "persist0" GI_GEMM_PERSIST=0 GI_CHAIN_RING=2 --
  run "persist11" GI_GEMM_PERSIST=11 GI_CHAIN_RING=2 --
done
run "zinc persist0" GI_GEMM_PERSIST=0 GI_CHAIN_RING=2 -- --steps 10 --warmup 3 --shape zinc --batch 1000 --model ggnn
run "zinc persist11" GI_GEMM_PERSIST=11 GI_CHAIN_RING=2 -- --steps 10 --warmup 3 --shape zinc --batch 1000 --model ggnn
cat $S
