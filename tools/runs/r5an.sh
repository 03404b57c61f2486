cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5an; mkdir -p $O
tools/ab.sh -r 3 -o $O/ab_default "default" "msg_wgrad_bf16x3 GI_B3W_MSG=1" > /dev/null 2>&1; cat $O/ab_default/summary.txt
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "default" "msg_wgrad_bf16x3 GI_B3W_MSG=1" > /dev/null 2>&1; cat $O/ab_zinc/summary.txt
