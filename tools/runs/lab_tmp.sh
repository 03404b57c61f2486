tools/ab.sh -r 3 -o gpurun_out/ab_hold "hold" "nohold GI_HOLD_KICKS=0" "kick4 GI_KICK_N=4" "msgslab GI_MSG_SLAB_ROWS=920"
