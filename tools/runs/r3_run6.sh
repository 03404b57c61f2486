#!/bin/bash
# round 3, GPU call 6: k-tile stream kernel (final form: persist off by default), unit tests, lab, step A/B, suite
OUT=/root/repo/gpurun_out/r3_run6; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or chain" 2>&1 | tail -15) > $OUT/gemm_tests.log; tail -3 $OUT/gemm_tests.log
L=$OUT/lab.txt; : > $L
for cls in "fwd 1 1" "fwd 2 2" "dgrad 1 1" "wgrad 1 1" "tier2 1 1"; do
  echo -n "old " >> $L; timeout 60 tools/gemm_lab_old $cls 0 >> $L 2>&1
  for ps in 0 11; do echo -n "new " >> $L; timeout 60 tools/gemm_lab $cls $ps >> $L 2>&1; done
done
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
  run "default(persist0,ring2)" X=1 --
  run "persist11" GI_GEMM_PERSIST=11 --
  run "ring3" GI_CHAIN_RING=3 --
done
run "zinc default" X=1 -- --steps 10 --warmup 3 --shape zinc --batch 1000 --model ggnn
run "zinc persist11" GI_GEMM_PERSIST=11 -- --steps 10 --warmup 3 --shape zinc --batch 1000 --model ggnn
run "chembl default" X=1 -- --steps 10 --warmup 3 --shape chembl --batch 250 --model attggnn
run "chembl persist11" GI_GEMM_PERSIST=11 -- --steps 10 --warmup 3 --shape chembl --batch 250 --model attggnn
cat $S
(timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $OUT/suite.log; tail -5 $OUT/suite.log
