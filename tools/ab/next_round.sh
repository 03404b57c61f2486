#!/bin/bash
# First GPU call of the next round (≈4 min): experiments prepared at the end of round 2 without GPU time left.
#   1. GI_CHAIN_RING=2 (gi_chain_kernel<BWD, 1, 2>, 113 KB of LDS: a GEMM workgroup fits beside a chain
#      workgroup) has never run on hardware: its gated tests first (GI_TEST_EXPERIMENTAL=1), then A/B.
#      every overlap schedule so far was bounded by chain workgroups owning their CU.
#   2. the same knob together with the pipelined readout update (GI_PIPELINE_READOUT=1) and with
#      GI_FUSE=31 (aggregation backward inside the dZ chain) — both ties on their own.
#   3. 64-row chain blocks everywhere (GI_CHAIN_ROWS64=1) on the headline: half the CUs stay free for the
#      weight-gradient stream.
# Output: gpurun_out/next/summary.txt.  Adopt only what wins two A/B pairs AND passes `pytest -m gpu`.
OUT=/root/repo/gpurun_out/next; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
S=$OUT/summary.txt; : > $S
(GI_TEST_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_kernels_gpu.py -q --maxfail=3 -k "two_slot_ring" 2>&1 | tail -15) > $OUT/ring2_tests.log
echo "ring2 tests: $(grep -E 'passed|failed' $OUT/ring2_tests.log | tail -1)" >> $S
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5"
run() {
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 90 $B "$@" 2>/dev/null | tail -1 | python -c "
import json, sys
try:
    d = json.load(sys.stdin); print('$label:', d['ms_per_step'], 'frac', d['roofline']['frac'], 'loss', d['config']['loss'])
except Exception as e:
    print('$label: FAILED', repr(e))" >> $S 2>&1
}
for rep in 1 2; do
  run "default" X=1 --
  run "ring2" GI_CHAIN_RING=2 --
  run "ring2 + pipelined readout" GI_CHAIN_RING=2 GI_PIPELINE_READOUT=1 --
  run "ring2 + fuse31" GI_CHAIN_RING=2 GI_FUSE=31 --
  run "rows64" GI_CHAIN_ROWS64=1 --
  run "rows64 + pipelined readout" GI_CHAIN_ROWS64=1 GI_PIPELINE_READOUT=1 --
done
for cfg in "zinc 1000 ggnn" "chembl 250 attggnn"; do
  set -- $cfg
  run "$1 default" X=1 -- --steps 10 --warmup 3 --shape $1 --batch $2 --model $3
  run "$1 ring2" GI_CHAIN_RING=2 -- --steps 10 --warmup 3 --shape $1 --batch $2 --model $3
  run "$1 pipelined readout" GI_PIPELINE_READOUT=1 -- --steps 10 --warmup 3 --shape $1 --batch $2 --model $3
done
cat $S
