#!/bin/bash
OUT=/root/repo/gpurun_out/r3_run12; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_p0cache_gpu.py -m gpu -q -x 2>&1 | grep -v "^  File\|^Extension" | tail -30) > $OUT/p0.log; tail -15 $OUT/p0.log | cut -c1-220
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File\|^Extension" | tail -30) > $OUT/suite.log; tail -8 $OUT/suite.log | cut -c1-200
for c in "tier2 1 1 0" ; do tools/gemm_lab $c; done > $OUT/lab.txt 2>&1
for ns in 2 3 4; do GI_LAB_NSPLIT=$ns tools/gemm_lab tier2s 1 1 0; done >> $OUT/lab.txt 2>&1
cat $OUT/lab.txt
for c in 0 1 0 1; do CACHE=$c python tools/runs/fwd_probe.py 2>&1 | tail -1; done | tee $OUT/fwd.txt
for m in sync_free blocking; do for c in 0 1; do MODE=$m CACHE=$c python tools/runs/gen_probe.py 2>&1 | tail -1; done; done | tee $OUT/gen.txt
python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-one-stream --steps 20 --warmup 5 2>$OUT/err.txt | tail -1 > $OUT/bench.json
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3_run12/bench.json'))
print(d['ms_per_step'], 'fwd', d['forward_only'], 'gen', json.dumps(d['generation_loop']))
PY
cd /tmp
for c in 0 1; do
CACHE=$c STEPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd_c$c -o p -- python /root/repo/tools/runs/fwd_probe.py > /dev/null 2>&1
rm -f $OUT/prof_fwd_c$c/*/*kernel_trace.csv $OUT/prof_fwd_c$c/*kernel_trace.csv
done
MODE=sync_free CACHE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_gen -o p -- python /root/repo/tools/runs/gen_probe.py > /dev/null 2>&1
rm -f $OUT/prof_gen/*/*kernel_trace.csv $OUT/prof_gen/*kernel_trace.csv
ls -R $OUT | head -40
