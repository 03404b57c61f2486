cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5ah; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "chain" > $O/k.log 2>&1; tail -3 $O/k.log
cd /tmp
for v in "0 0" "1 0" "1 1"; do set -- $v; rm -rf /tmp/cb; BENCH_CHAIN_X2=$1 BENCH_CHAIN_ROWS32=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb -o b -- python /root/repo/tools/bench_chain.py both > /dev/null 2>&1; echo "x2=$1 rows32=$2:"; grep 'gi_chain' /tmp/cb/*kernel_stats.csv | grep -v pack | sed 's/(anonymous namespace):://g;s/"void //' | awk -F'"' '{print $1 $2 $3}' | cut -c1-90; done > $O/chain_variants.txt 2>&1; cat $O/chain_variants.txt
