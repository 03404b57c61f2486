cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_x2_trial_gpu.py -q -s > $O/gpu_trained_checkpoint.log 2>&1; echo "trial rc $?"; tail -2 $O/gpu_trained_checkpoint.log
B="python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-extra-configs --no-one-stream --no-forward-only"
bash tools/pmc_kernel.sh gi_chain $O/pmc_chain_kernels.txt -- $B > /dev/null 2>&1
GI_CHAIN_XCD=0 bash tools/pmc_kernel.sh gi_chain $O/pmc_chain_kernels_dispatch_order.txt -- $B > /dev/null 2>&1
cat $O/pmc_chain_kernels.txt; echo; cat $O/pmc_chain_kernels_dispatch_order.txt
