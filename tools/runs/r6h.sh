cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6h; mkdir -p $O
( time timeout 600 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt; tail -3 $O/time.txt; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r6h/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], 'loader', d.get('loader_inclusive'), 'cpu', d.get('cpu_baseline'), 'guard', d.get('x2_guard'))
PY
timeout 600 python -m pytest tests/test_bench_gpu.py -q -x 2>&1 | tail -3
