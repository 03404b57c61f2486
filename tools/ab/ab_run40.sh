#!/bin/bash
# Final collection of the round with GI_FUSE_DEFAULT = 15: GPU suite, kernel stats + bench lines for profiles/r02,
# then A/B sweeps of the scheduling knobs around the new default.  Incremental output under gpurun_out/run40/.
OUT=/root/repo/gpurun_out/run40; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
S=$OUT/summary.txt; : > $S
note() { echo "$(date +%H:%M:%S) $*" >> $S; }
note start
(timeout 400 python -m pytest tests -m gpu -q --maxfail=4 2>&1 | tail -40) > $OUT/suite_default.log
note "suite(default=fuse15): $(grep -E 'passed|failed' $OUT/suite_default.log | tail -1)"
cd /tmp; rm -rf /tmp/st40
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st40 -o bench -- \
  python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-forward-only --no-probe --no-one-stream \
  > $OUT/bench_under_rocprof.log 2>&1
cp /tmp/st40/bench_kernel_stats.csv $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
cp /tmp/st40/bench_domain_stats.csv $OUT/rocprofv3_domain_stats.csv 2>/dev/null
grep "^{\"metric\"" $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
note "rocprof done"
cd /root/repo
timeout 300 python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json
note "bench default: $(python -c "import json; d=json.load(open('$OUT/bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], [e['ms_per_step'] for e in d['extra_configs']], d['cpu_baseline']['value'])" 2>&1)"
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5"
run() {   # label, env assignments..., then bench args after --
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 $B "$@" 2>/dev/null | tail -1 | python -c "
import json, sys
try:
    d = json.load(sys.stdin); print('$label:', d['ms_per_step'], 'frac', d['roofline']['frac'], 'loss', d['config']['loss'])
except Exception as e:
    print('$label: FAILED', repr(e))" >> $S 2>&1
}
for rep in 1 2 3; do
  run "fuse0" GI_FUSE=0 --
  run "fuse15" GI_FUSE=15 --
done
for rep in 1 2; do
  run "fuse15 kick4" GI_WGRAD_KICK=4 --
  run "fuse15 kick2" GI_WGRAD_KICK=2 --
  run "fuse15 wgs160" GI_WGRAD_WGS=160 --
  run "fuse15 wgs224" GI_WGRAD_WGS=224 --
  run "fuse15 hiprio" GI_BENCH_HIPRIO=1 --
  run "fuse15 inline2" GI_WGRAD_INLINE=2 --
  run "fuse15 fuse7(no slots)" GI_FUSE=7 --
  run "fuse15 gatesonly" GI_FUSE=3 --
done
note "sweep done"
for rep in 1 2; do
  run "zinc fuse0" GI_FUSE=0 -- --steps 10 --warmup 3 --shape zinc --batch 1000 --model ggnn
  run "zinc fuse15" GI_FUSE=15 -- --steps 10 --warmup 3 --shape zinc --batch 1000 --model ggnn
  run "chembl fuse0" GI_FUSE=0 -- --steps 10 --warmup 3 --shape chembl --batch 250 --model attggnn
  run "chembl fuse15" GI_FUSE=15 -- --steps 10 --warmup 3 --shape chembl --batch 250 --model attggnn
  run "chembl fuse31" GI_FUSE=31 -- --steps 10 --warmup 3 --shape chembl --batch 250 --model attggnn
done
note "shapes done"
cat $S
