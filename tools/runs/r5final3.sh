# the full -m gpu suite on the final tree (after the fixture-training criterion was made trajectory-robust) + three bench lines
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_x2_trial_gpu.py > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'])"; done > $O/bench_default_repeats_second_box.txt; cat $O/bench_default_repeats_second_box.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
