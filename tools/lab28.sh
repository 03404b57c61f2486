#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r4z; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 > $O/pytest.txt
tools/ab.sh -r 2 -o $O/ab_head "chainx2" "chainfp32 GI_CHAIN_X2=0" > $O/ab_head.txt 2>&1
tools/ab.sh -r 1 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "chainx2" "chainfp32 GI_CHAIN_X2=0" > $O/ab_zinc.txt 2>&1
tools/ab.sh -r 1 -o $O/ab_att -a "--shape chembl --batch 250 --model attggnn --steps 10 --warmup 3" "chainx2" "chainfp32 GI_CHAIN_X2=0" > $O/ab_att.txt 2>&1
tools/ab.sh -r 1 -o $O/ab_4k -a "--batch 4000 --steps 10 --warmup 3" "chainx2" "chainfp32 GI_CHAIN_X2=0" > $O/ab_4k.txt 2>&1
cat $O/pytest.txt $O/ab_head.txt $O/ab_zinc.txt $O/ab_att.txt $O/ab_4k.txt
