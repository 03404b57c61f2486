cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5am; mkdir -p $O
timeout 400 python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('frac_own_pipe'), d['roofline']['traffic'], d['cpu_baseline']['value'], [e['ms_per_step'] for e in d['extra_configs']], d['forward_only']['ms_per_step'])"
