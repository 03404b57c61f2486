cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "chain" 2>&1 | tail -3
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('ring3', d['ms_per_step'], d['config']['loss'])"; done | tee $O/head.txt
python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 10 --warmup 3 --shape zinc --batch 1000 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('ring3 zinc', d['ms_per_step'])"
tools/collect_traces.sh r6l_tr > /dev/null 2>&1; cat gpurun_out/r6l_tr/critical_path_default.txt | grep "message\|step"
