cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5ad; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "chain" > $O/k.log 2>&1; tail -5 $O/k.log
cd /tmp
for v in "1 0"; do set -- $v; rm -rf /tmp/cb; BENCH_CHAIN_X2=$1 BENCH_CHAIN_ROWS32=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb -o b -- python /root/repo/tools/bench_chain.py both > /dev/null 2>&1; echo "x2=$1 rows32=$2:"; grep 'gi_chain' /tmp/cb/*kernel_stats.csv | grep -v pack | sed 's/(anonymous namespace):://g;s/"void //' | awk -F'"' '{print $1 $2 $3}' | cut -c1-90; done > $O/chain_variants.txt 2>&1; cat $O/chain_variants.txt
rm -rf /tmp/cs; CHAIN_SCALING_KINDS=x2 rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o t -- python /root/repo/tools/chain_scaling.py run > $O/run.log 2>&1
python /root/repo/tools/chain_scaling.py report /tmp/cs/*kernel_trace.csv > $O/chain_scaling.txt; cat $O/chain_scaling.txt
cd /root/repo
tools/ab.sh -r 3 -o $O/ab_default "default" "bwd_fp32 GI_CHAIN_X2=0" > /dev/null 2>&1; cat $O/ab_default/summary.txt
