#!/bin/bash
# round 3, GPU call 9: sync-free forward (grid-stride chains), chain tests, generation loop, step A/B vs before
OUT=/root/repo/gpurun_out/r3_run9; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_syncfree_gpu.py tests/test_kernels_gpu.py -q -x -k "syncfree or bounded or generation or chain or violations or capturable" 2>&1 | tail -15) > $OUT/tests.log; tail -6 $OUT/tests.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-one-stream --steps 20 --warmup 5 2>$OUT/err.txt | tail -1 > $OUT/bench.json
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3_run9/bench.json'))
g=d['generation_loop']
print(d['ms_per_step'], 'fwd', d['forward_only']['ms_per_step'], 'gen sync_free', g['sync_free']['ms_per_round'], 'blocking', g['blocking_readback']['ms_per_round'])
PY
done
