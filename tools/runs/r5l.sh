cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5l; mkdir -p $O
tools/ab.sh -r 2 -o /root/repo/$O/abz -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "hold" "nohold GI_HOLD_KICKS=0" > $O/abz.log 2>&1; cat $O/abz/summary.txt
tools/ab.sh -r 2 -o /root/repo/$O/abc -a "--shape chembl --model attggnn --batch 250 --steps 10 --warmup 3" "hold" "nohold GI_HOLD_KICKS=0" > $O/abc.log 2>&1; cat $O/abc/summary.txt
tools/ab.sh -r 2 -o /root/repo/$O/ab4 -a "--batch 4000 --steps 10 --warmup 3" "hold" "nohold GI_HOLD_KICKS=0" > $O/ab4.log 2>&1; cat $O/ab4/summary.txt
tools/ab.sh -r 2 -o /root/repo/$O/ab2 -a "--batch 2000 --steps 10 --warmup 3" "hold" "nohold GI_HOLD_KICKS=0" > $O/ab2.log 2>&1; cat $O/ab2/summary.txt
