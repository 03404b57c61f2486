#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-extra-configs --no-probe --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print(d['ms_per_step'], d['value'], 'frac', r['frac'], 'traffic', r['traffic'], 'one_stream', r['one_stream']['frac'], 'fp32only', d['fp32_mfma_only']['ms_per_step'], 'fwd', d['forward_only']['ms_per_step'], 'gen', d['generation_loop']['sync_free']['ms_per_round'], d['generation_loop']['sync_free_hipgraph'])"
