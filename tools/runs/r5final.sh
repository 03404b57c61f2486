# round 5: the full -m gpu suite, the default bench line, rocprofv3 stats + PMC passes, traces of the four configurations,
# the fp16x2 trial table, the chain kernels' LDS / L2 counters, the model tests in the other two arithmetic modes
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_x2_trial_gpu.py > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
timeout 600 python -m pytest tests/test_x2_trial_gpu.py -q -s > $O/gpu_trained_checkpoint.log 2>&1; echo "trial rc $?"; tail -2 $O/gpu_trained_checkpoint.log
timeout 300 python tests/test_x2_trial_gpu.py > $O/x2_trial.txt 2>&1
timeout 600 python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('frac_own_pipe'), d['cpu_baseline']['value'], d.get('x2_guard'))"
for i in 1 2; do python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'])"; done > $O/bench_default_repeats.txt; cat $O/bench_default_repeats.txt
bash tools/collect_profiles.sh r05 > $O/collect_profiles.log 2>&1; tail -3 $O/collect_profiles.log
GI_TRACE_ALL=1 bash tools/collect_traces.sh r05 > $O/collect_traces.log 2>&1; tail -45 $O/collect_traces.log
bash tools/pmc_kernel.sh gi_chain $O/pmc_chain_kernels.txt -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-extra-configs --no-one-stream --no-forward-only > /dev/null 2>&1
GI_CHAIN_XCD=0 bash tools/pmc_kernel.sh gi_chain $O/pmc_chain_kernels_dispatch_order.txt -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-extra-configs --no-one-stream --no-forward-only > /dev/null 2>&1
GI_X2=0 timeout 600 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py -q > $O/gpu_model_tests_bf16x3_only.log 2>&1; tail -1 $O/gpu_model_tests_bf16x3_only.log
GI_BF3=0 timeout 600 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py -q > $O/gpu_model_tests_fp32_mfma_only.log 2>&1; tail -1 $O/gpu_model_tests_fp32_mfma_only.log
for u in 1 2 4; do echo "U=$u: $(GI_SEGSUM_U=$u python bench.py --probe-only 2>/dev/null | tail -1)"; done > $O/segsum_probe.txt
ls $O
