#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run11; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File\|^Extension" | tail -30) > $OUT/suite.log; tail -12 $OUT/suite.log | cut -c1-200
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-one-stream --steps 20 --warmup 5 2>$OUT/err.txt | tail -1 > $OUT/bench.json
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3_run11/bench.json'))
g=d['generation_loop']
print(d['ms_per_step'], 'fwd', d['forward_only']['ms_per_step'], 'gen sync_free', g['sync_free']['ms_per_round'], 'blocking', g['blocking_readback']['ms_per_round'])
PY
done
