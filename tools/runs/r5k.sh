cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5k; mkdir -p $O
tools/ab.sh -r 3 -o /root/repo/$O/ab "new" "nohold GI_HOLD_KICKS=0" > $O/ab.log 2>&1; cat $O/ab/summary.txt
tools/ab.sh -r 1 -o /root/repo/$O/abz -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "new" "nohold GI_HOLD_KICKS=0" > $O/abz.log 2>&1; cat $O/abz/summary.txt
