cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5o; mkdir -p $O
tools/ab.sh -r 3 -o /root/repo/$O/ab "new" "ring3small GI_CHAIN_RING3_SMALL=32" > $O/ab.log 2>&1; cat $O/ab/summary.txt
