#!/bin/bash
# A/B of bench.py variants on the GPU box, one line per run (what tools/ab/ab_run*.sh did one script at a time).
#   tools/ab.sh [-r REPS] [-o OUTDIR] [-a "extra bench args"] "label ENV=1 ENV2=x" "label2 ..." ...
# Every argument is one variant: its first word is the label, the rest are environment assignments.  The
# variants run round-robin REPS times (default 2) so that drift hits all of them alike; summary -> OUTDIR/summary.txt
# e.g.  tools/ab.sh -r 2 "default" "fp32 GI_BF3=0" -a "--shape zinc --batch 1000 --steps 10 --warmup 3"
REPS=2; OUT=/root/repo/gpurun_out/ab; ARGS=""
while getopts "r:o:a:" o; do case $o in r) REPS=$OPTARG;; o) OUT=$OPTARG;; a) ARGS=$OPTARG;; esac; done
shift $((OPTIND - 1))
mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
S=$OUT/summary.txt; : > $S
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5 $ARGS"
VARIANTS=("$@")
for rep in $(seq $REPS); do
  for v in "${VARIANTS[@]}"; do
    read -r -a words <<< "$v"; label=${words[0]}; envs=("${words[@]:1}")
    env "${envs[@]}" X_AB=1 timeout 120 $B 2>$OUT/err.txt | tail -1 | python -c "
import json, sys
try:
    d = json.load(sys.stdin); print('$label:', d['ms_per_step'], 'frac', d['roofline']['frac'], 'loss', d['config']['loss'])
except Exception as e:
    print('$label: FAILED', repr(e), open('$OUT/err.txt').read()[-500:])" >> $S 2>&1
  done
done
cat $S
