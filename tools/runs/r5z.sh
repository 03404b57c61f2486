cd /tmp; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5z; mkdir -p $O
export CHAIN_SCALING_KINDS=x2r
for cfg in "1 512" "0 256"; do set -- $cfg
for m in 0 1 2 4 6 7 8 16 24 32 33 38 39; do
rm -rf /tmp/cs; GI_DBG_X2=$m GI_CHAIN_X2R_DUAL=$1 CHAIN_SCALING_BLOCKS=$2 rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o t -- python /root/repo/tools/chain_scaling.py run > $O/run.log 2>&1
echo "dual=$1 blocks=$2 mask=$m: $(python /root/repo/tools/chain_scaling.py report /tmp/cs/*kernel_trace.csv | tail -1)"; done; done > $O/x2r_breakdown.txt 2>&1; cat $O/x2r_breakdown.txt
