cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5u; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "row_independent or rows32" > $O/k.log 2>&1; tail -5 $O/k.log
cd /tmp
for v in "0 0" "1 1"; do set -- $v; rm -rf /tmp/cb; BENCH_CHAIN_X2=$1 BENCH_CHAIN_ROWS32=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb -o b -- python /root/repo/tools/bench_chain.py both > /dev/null 2>&1; echo "x2=$1 rows32=$2:"; grep 'gi_chain' /tmp/cb/*kernel_stats.csv | grep -v pack | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-120; done > $O/chain_variants.txt 2>&1; cat $O/chain_variants.txt
cd /root/repo
tools/ab.sh -r 3 -o $O/ab_default "fwd_x2r" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_default/summary.txt
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "fwd_x2r" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_zinc/summary.txt
tools/ab.sh -r 2 -o $O/ab_chembl -a "--model attggnn --shape chembl --batch 250 --steps 10 --warmup 3" "fwd_x2r" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_chembl/summary.txt
timeout 500 python -m pytest tests/test_model_gpu.py -q -x > $O/model.log 2>&1; tail -8 $O/model.log
timeout 300 python -m pytest tests/test_x2_trial_gpu.py -q -x -k "trained or guard" > $O/trial.log 2>&1; tail -8 $O/trial.log
