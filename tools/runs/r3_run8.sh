#!/bin/bash
# round 3, GPU call 8: sync-free forward tests, generation-loop leg, loader bench, suite
OUT=/root/repo/gpurun_out/r3_run8; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_syncfree_gpu.py -q -x 2>&1 | tail -30) > $OUT/syncfree.log; tail -30 $OUT/syncfree.log
python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-one-stream --steps 20 --warmup 5 2>$OUT/err.txt | tail -1 > $OUT/bench.json
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3_run8/bench.json'))
print(d['ms_per_step'], d['forward_only'], json.dumps(d['generation_loop']))
PY
tail -5 $OUT/err.txt
(timeout 300 python tools/bench_loader.py 2>&1 | tail -12) > $OUT/bench_loader.txt; cat $OUT/bench_loader.txt
