# round 6, call 1: the new-dimension parity tests, the baseline bench line of HEAD, and how long the message stacks'
# weight gradients take as bf16x3 launches of gi_b3p_kernel (GI_B3W_MSG=1) in a traced step
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_dims_gpu.py -q -s -x > $O/dims.log 2>&1; echo "dims rc $?"; tail -25 $O/dims.log
timeout 300 python bench.py > $O/bench_head.json 2> $O/bench_head.err; tail -c 1500 $O/bench_head.json
tools/ab.sh -r 2 -o $O/ab_msg "default" "msg_wgrad_bf16x3 GI_B3W_MSG=1" > /dev/null 2>&1; cat $O/ab_msg/summary.txt
GI_B3W_MSG=1 tools/collect_traces.sh r6a_msgb3 > $O/traces.log 2>&1; tail -15 $O/traces.log
