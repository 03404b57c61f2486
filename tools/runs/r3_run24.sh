#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run24; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File\|^Extension" | tail -30) > $OUT/suite.log; tail -8 $OUT/suite.log | cut -c1-200
for c in 0 1; do CACHE=$c python tools/runs/fwd_probe.py 2>&1 | tail -1; done | tee $OUT/fwd.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-one-stream --steps 20 --warmup 5 2>$OUT/err.txt | tail -1 > $OUT/bench.json
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3_run24/bench.json'))
print(d['ms_per_step'], 'fwd', d['forward_only']['ms_per_step'], 'gen', json.dumps({k:(v.get('ms_per_round') if isinstance(v,dict) else v) for k,v in d['generation_loop'].items() if k!='note'}))
PY
done
CACHE=0 python tools/runs/dbg_graph8.py 2>&1 | tail -1 | cut -c1-200
