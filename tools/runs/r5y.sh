cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5y; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "row_independent or rows32" > $O/k.log 2>&1; tail -5 $O/k.log
GI_CHAIN_X2R_DUAL=0 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "row_independent or rows32" > $O/k0.log 2>&1; tail -2 $O/k0.log
cd /tmp
for d in 1 0; do
rm -rf /tmp/cs; GI_CHAIN_X2R_DUAL=$d rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o t -- python /root/repo/tools/chain_scaling.py run > $O/run.log 2>&1
echo "GI_CHAIN_X2R_DUAL=$d"; python /root/repo/tools/chain_scaling.py report /tmp/cs/*kernel_trace.csv; done > $O/chain_scaling.txt 2>&1; cat $O/chain_scaling.txt
cd /root/repo
tools/ab.sh -r 3 -o $O/ab_default "dual" "ext GI_CHAIN_X2R_DUAL=0" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_default/summary.txt
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "dual" "ext GI_CHAIN_X2R_DUAL=0" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_zinc/summary.txt
tools/ab.sh -r 2 -o $O/ab_chembl -a "--model attggnn --shape chembl --batch 250 --steps 10 --warmup 3" "dual_forced GI_CHAIN_FWD_X2=2" "fwd_fp32" > /dev/null 2>&1; cat $O/ab_chembl/summary.txt
