cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5j; mkdir -p $O
tools/ab.sh -r 2 -o /root/repo/$O/ab "new" "t12 GI_WGRAD_TN=12" "t12_128 GI_WGRAD_TN=12 GI_WGRAD_WGS=128" "t12_256 GI_WGRAD_TN=12 GI_WGRAD_WGS=256" > $O/ab.log 2>&1; cat $O/ab/summary.txt
tools/ab.sh -r 1 -o /root/repo/$O/abz -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "new" "t12 GI_WGRAD_TN=12" > $O/abz.log 2>&1; cat $O/abz/summary.txt
