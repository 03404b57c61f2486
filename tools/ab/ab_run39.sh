#!/bin/bash
# GI_FUSE launch-count reductions (gi_fuse_flags): the whole GPU suite with every variant on, then A/B of
# the training step (off / the elementwise set / + the chain-input fusion / + node-level weight gradients
# inline), then the kernel stats of the default bench command with the variants on.
# Ordered by importance: the call may be cut short by the round's GPU budget; everything is written
# incrementally under gpurun_out/run39/.
OUT=/root/repo/gpurun_out/run39; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
S=$OUT/summary.txt; : > $S
note() { echo "$(date +%H:%M:%S) $*" >> $S; }
note start
(GI_FUSE=31 timeout 240 python -m pytest tests/test_kernels_gpu.py -q --maxfail=6 \
   -k "gates or cols3 or slot_glue or segmented or chain or seg_sum" 2>&1 | tail -25) > $OUT/units_fuse31.log
note "units(fuse31): $(tail -1 $OUT/units_fuse31.log)"
(GI_FUSE=31 timeout 700 python -m pytest tests -m gpu -q --maxfail=4 2>&1 | tail -40) > $OUT/suite_fuse31.log
note "suite(fuse31): $(tail -1 $OUT/suite_fuse31.log)"
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5"
for rep in 1 2; do
  for cfg in "0 0" "31 0" "15 0" "31 1"; do
    set -- $cfg
    GI_FUSE=$1 GI_WGRAD_INLINE=$2 timeout 120 $B 2>$OUT/bench_$1_$2_$rep.err | tail -1 > $OUT/bench_$1_$2_$rep.json
    python - "$1" "$2" $OUT/bench_$1_$2_$rep.json >> $S <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print("fuse=%s inline=%s: %.3f ms/step  frac %.4f  launches/step %d  loss %s" % (
        sys.argv[1], sys.argv[2], d["ms_per_step"], d["roofline"]["frac"],
        d["roofline"]["launches_per_step"], d["config"]["loss"]))
except Exception as e:
    print("fuse=%s inline=%s: FAILED %r" % (sys.argv[1], sys.argv[2], e))
PY
  done
done
note "A/B done"
# the other two shapes, variants off / on
for cfg in "zinc 1000 ggnn" "chembl 250 attggnn"; do
  set -- $cfg
  for f in 0 31; do
    GI_FUSE=$f timeout 120 $B --steps 10 --warmup 3 --shape $1 --batch $2 --model $3 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.load(sys.stdin); print('$1 $2 $3 fuse=$f:', d['ms_per_step'], d['roofline']['frac'], d['config']['loss'])" >> $S 2>&1
  done
done
note "shapes done"
# kernel stats of the default bench command with the variants on (what profiles/ must agree with)
cd /tmp; rm -rf /tmp/st39
GI_FUSE=31 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st39 -o bench -- \
  python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-forward-only --no-probe --no-one-stream \
  > $OUT/bench_under_rocprof_fuse31.log 2>&1
cp /tmp/st39/bench_kernel_stats.csv $OUT/rocprofv3_kernel_stats_fuse31.csv 2>/dev/null
grep "^{\"metric\"" $OUT/bench_under_rocprof_fuse31.log > $OUT/bench_under_rocprof_fuse31.json
note "rocprof done"
cd /root/repo
(timeout 700 python -m pytest tests -m gpu -q --maxfail=4 2>&1 | tail -15) > $OUT/suite_default.log
note "suite(default): $(tail -1 $OUT/suite_default.log)"
cat $S
