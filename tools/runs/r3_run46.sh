#!/bin/bash
# run-to-run / box-to-box spread of the default bench line on the final code
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3_run46
python bench.py 2>/dev/null | tail -1 > gpurun_out/r3_run46/bench_default_$(date +%s).json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r3_run46/bench_default_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], d['value'], 'frac', r['frac'], 'fp32only', d['fp32_mfma_only']['ms_per_step'], 'fwd', d['forward_only']['ms_per_step'], 'gen', d['generation_loop']['sync_free']['ms_per_round'], 'zinc', d['extra_configs'][0]['ms_per_step'], 'chembl', d['extra_configs'][1]['ms_per_step'], 'cpu', d['cpu_baseline']['value'])
PY
