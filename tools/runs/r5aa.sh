cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5aa; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "row_independent or rows32" > $O/k.log 2>&1; tail -5 $O/k.log
GI_CHAIN_X2R_DUAL=0 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "row_independent or rows32" > $O/k0.log 2>&1; tail -2 $O/k0.log
cd /tmp
for d in 1 0; do
rm -rf /tmp/cs; GI_CHAIN_X2R_DUAL=$d CHAIN_SCALING_KINDS=x2r rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o t -- python /root/repo/tools/chain_scaling.py run > $O/run.log 2>&1
echo "GI_CHAIN_X2R_DUAL=$d"; python /root/repo/tools/chain_scaling.py report /tmp/cs/*kernel_trace.csv; done > $O/chain_scaling.txt 2>&1; cat $O/chain_scaling.txt
export CHAIN_SCALING_KINDS=x2r
for cfg in "1 512" "0 256"; do set -- $cfg
for m in 0 1 6 7 8 16 24 32 38 39; do
rm -rf /tmp/cs; GI_DBG_X2=$m GI_CHAIN_X2R_DUAL=$1 CHAIN_SCALING_BLOCKS=$2 rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o t -- python /root/repo/tools/chain_scaling.py run > $O/run.log 2>&1
echo "dual=$1 blocks=$2 mask=$m: $(python /root/repo/tools/chain_scaling.py report /tmp/cs/*kernel_trace.csv | tail -1)"; done; done > $O/x2r_breakdown.txt 2>&1; cat $O/x2r_breakdown.txt
