#!/bin/bash
# Last GPU call of the round: the whole suite on the final code (incl. tests/test_pipeline_gpu.py), then A/B of the
# opt-in pipelined readout update (GI_PIPELINE_READOUT=1) against the default trainer.
OUT=/root/repo/gpurun_out/run43; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
S=$OUT/summary.txt; : > $S
(timeout 400 python -m pytest tests -m gpu -q --maxfail=6 2>&1 | tail -60) > $OUT/suite_default.log
echo "$(date +%H:%M:%S) suite: $(grep -E 'passed|failed' $OUT/suite_default.log | tail -1)" >> $S
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5"
run() {
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 60 $B "$@" 2>$OUT/err.txt | tail -1 | python -c "
import json, sys
try:
    d = json.load(sys.stdin); print('$label:', d['ms_per_step'], 'frac', d['roofline']['frac'], 'loss', d['config']['loss'], d['config'].get('pipeline_readout'))
except Exception as e:
    print('$label: FAILED', repr(e), open('$OUT/err.txt').read()[-400:])" >> $S 2>&1
}
for rep in 1 2; do
  run "default" X=1 --
  run "pipelined" GI_PIPELINE_READOUT=1 --
done
cat $S
