#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
tools/ab.sh -r 2 -o $O/ab_head "default" "msgb3v GI_B3W_MSG=1 GI_B3V_GROUPED=1" "msgb3p GI_B3W_MSG=1" > $O/ab_head.txt 2>&1
cat $O/ab_head.txt
