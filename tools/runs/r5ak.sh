cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5ak; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "chain" > $O/k.log 2>&1; tail -3 $O/k.log
cd /tmp
rm -rf /tmp/cs; CHAIN_SCALING_KINDS=x2r rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o t -- python /root/repo/tools/chain_scaling.py run > /dev/null 2>&1
python /root/repo/tools/chain_scaling.py report /tmp/cs/*kernel_trace.csv > $O/chain_scaling.txt; cat $O/chain_scaling.txt
for v in "1 1"; do set -- $v; rm -rf /tmp/cb; BENCH_CHAIN_X2=$1 BENCH_CHAIN_ROWS32=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb -o b -- python /root/repo/tools/bench_chain.py both > /dev/null 2>&1; grep 'gi_chain' /tmp/cb/*kernel_stats.csv | grep -v pack | sed 's/(anonymous namespace):://g' | cut -d, -f1-4; done > $O/chain_variants.txt 2>&1; cat $O/chain_variants.txt
cd /root/repo
tools/ab.sh -r 3 -o $O/ab_default "default" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_default/summary.txt
tools/ab.sh -r 1 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "default" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_zinc/summary.txt
tools/ab.sh -r 1 -o $O/ab_chembl -a "--model attggnn --shape chembl --batch 250 --steps 10 --warmup 3" "default" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_chembl/summary.txt
