# round 5, final collection on the final library: the full -m gpu suite, the fitted-checkpoint trial, the default bench line (+ repeats),
# rocprofv3 stats + PMC passes, traces of the four configurations, the chain kernels alone / against their number of workgroups,
# their LDS / L2 counters, the model tests in the other two arithmetic modes
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_x2_trial_gpu.py > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
timeout 600 python -m pytest tests/test_x2_trial_gpu.py -q -s > $O/gpu_trained_checkpoint.log 2>&1; echo "trial rc $?"; tail -2 $O/gpu_trained_checkpoint.log
timeout 600 python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('frac_own_pipe'), d['cpu_baseline']['value'], d.get('x2_guard'))"
for i in 1 2; do python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'])"; done > $O/bench_default_repeats.txt; cat $O/bench_default_repeats.txt
bash tools/collect_profiles.sh r05 > $O/collect_profiles.log 2>&1; tail -3 $O/collect_profiles.log
GI_TRACE_ALL=1 bash tools/collect_traces.sh r05 > $O/collect_traces.log 2>&1; tail -45 $O/collect_traces.log
bash tools/pmc_kernel.sh gi_chain $O/pmc_chain_kernels.txt -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-extra-configs --no-one-stream --no-forward-only > /dev/null 2>&1
cd /tmp
for v in "0 0" "1 0" "1 1"; do set -- $v; rm -rf /tmp/cb; BENCH_CHAIN_X2=$1 BENCH_CHAIN_ROWS32=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb -o b -- python /root/repo/tools/bench_chain.py both > /dev/null 2>&1; echo "# BENCH_CHAIN_X2=$1 BENCH_CHAIN_ROWS32=$2 (tools/bench_chain.py both: 8 400 rows in three groups; name, launches, total ns, average ns, %, min, max, stddev)"; grep 'gi_chain' /tmp/cb/*kernel_stats.csv | grep -v pack | sed 's/(anonymous namespace):://g'; done > $O/chain_kernels_alone.txt 2>&1; cat $O/chain_kernels_alone.txt
rm -rf /tmp/cs; rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o t -- python /root/repo/tools/chain_scaling.py run > /dev/null 2>&1
python /root/repo/tools/chain_scaling.py report /tmp/cs/*kernel_trace.csv > $O/chain_scaling.txt 2>&1; cat $O/chain_scaling.txt
cd /root/repo
GI_X2=0 timeout 600 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py -q > $O/gpu_model_tests_bf16x3_only.log 2>&1; tail -1 $O/gpu_model_tests_bf16x3_only.log
GI_BF3=0 timeout 600 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py -q > $O/gpu_model_tests_fp32_mfma_only.log 2>&1; tail -1 $O/gpu_model_tests_fp32_mfma_only.log
ls $O
