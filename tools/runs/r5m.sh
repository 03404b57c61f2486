cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5m; mkdir -p $O
tools/ab.sh -r 2 -o /root/repo/$O/ab "new" "kick4 GI_KICK_N=4" "kick6 GI_KICK_N=6" "prio GI_SIDE_PRIO=0" "prio_kick4 GI_SIDE_PRIO=0 GI_KICK_N=4" > $O/ab.log 2>&1; cat $O/ab/summary.txt
