# Round 6, the collection behind profiles/r06/ (run on the GPU box: gpurun -- 'bash tools/runs/r06_collect.sh'):
# the -m gpu suite, the model tests in the other two arithmetic modes, the default bench line (+ repeats), rocprofv3
# kernel stats + the separate PMC passes (traffic.json), traces of the four configurations with the critical-path and
# GEMM-class reports, the chain kernels' counters.
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
GI_X2=0 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py tests/test_dims_gpu.py -q > $O/gpu_model_tests_bf16x3_only.log 2>&1; tail -1 $O/gpu_model_tests_bf16x3_only.log
GI_BF3=0 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py tests/test_dims_gpu.py -q > $O/gpu_model_tests_fp32_mfma_only.log 2>&1; tail -1 $O/gpu_model_tests_fp32_mfma_only.log
timeout 600 python -m pytest tests/test_dims_gpu.py -q -s 2>&1 | grep "GEMM-family\|passed\|failed" > $O/untuned_dimensions_pipes.txt; cat $O/untuned_dimensions_pipes.txt
timeout 600 python bench.py > $O/bench_first_run.log 2>&1; grep '^{"metric"' $O/bench_first_run.log > $O/bench_first_run.json; python -c "
import json; d=json.load(open('$O/bench_first_run.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('frac_own_pipe'), d['cpu_baseline']['value'], d.get('loader_inclusive',{}).get('value'), d.get('x2_guard'))"
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'])"; done > $O/bench_default_repeats.txt; cat $O/bench_default_repeats.txt
bash tools/collect_profiles.sh r06 > $O/collect_profiles.log 2>&1; tail -3 $O/collect_profiles.log
GI_TRACE_ALL=1 bash tools/collect_traces.sh r06 > $O/collect_traces.log 2>&1; tail -30 $O/collect_traces.log
bash tools/pmc_kernel.sh gi_chain $O/pmc_chain_kernels.txt -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-extra-configs --no-one-stream --no-forward-only > /dev/null 2>&1
ls $O
