cd /tmp; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5w; mkdir -p $O
rm -rf /tmp/cs; rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o t -- python /root/repo/tools/chain_scaling.py run > $O/run.log 2>&1; tail -3 $O/run.log
head -1 /tmp/cs/*kernel_trace.csv
python /root/repo/tools/chain_scaling.py report /tmp/cs/*kernel_trace.csv > $O/chain_scaling.txt 2>&1; cat $O/chain_scaling.txt
