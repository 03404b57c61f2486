cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5p; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "weight_gradient or fp16x2" > $O/k.log 2>&1; tail -1 $O/k.log
tools/ab.sh -r 3 -o /root/repo/$O/ab "remap" "noremap GI_B3P_WGRAD_REMAP=0" > $O/ab.log 2>&1; cat $O/ab/summary.txt
tools/ab.sh -r 2 -o /root/repo/$O/abz -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "remap" "noremap GI_B3P_WGRAD_REMAP=0" > $O/abz.log 2>&1; cat $O/abz/summary.txt
cd /tmp; for v in 1 0; do GI_B3P_WGRAD_REMAP=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st$v -o b -- python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-forward-only --no-probe --no-one-stream --steps 10 --warmup 3 > /dev/null 2>&1; echo "REMAP=$v"; grep "gi_b3p_kernel" /tmp/st$v/*kernel_stats.csv | cut -c1-160; done
