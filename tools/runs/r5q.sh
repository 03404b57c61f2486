cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5q; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "wgrad or split_k" > $O/k.log 2>&1; tail -1 $O/k.log
tools/ab.sh -r 3 -o /root/repo/$O/ab "slab" "noslab GI_WGRAD_SLAB_ORDER=0" "r5base GI_WGRAD_SLAB_ORDER=0 GI_B3P_WGRAD_REMAP=0" > $O/ab.log 2>&1; cat $O/ab/summary.txt
tools/ab.sh -r 2 -o /root/repo/$O/abz -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "slab" "noslab GI_WGRAD_SLAB_ORDER=0" "r5base GI_WGRAD_SLAB_ORDER=0 GI_B3P_WGRAD_REMAP=0" > $O/abz.log 2>&1; cat $O/abz/summary.txt
