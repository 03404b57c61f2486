cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5ap; mkdir -p $O
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "default" "bwd_x2r GI_CHAIN_BWD_X2R=1" > /dev/null 2>&1; cat $O/ab_zinc/summary.txt
tools/ab.sh -r 2 -o $O/ab_chembl -a "--model attggnn --shape chembl --batch 250 --steps 10 --warmup 3" "default" "bwd_x2r GI_CHAIN_BWD_X2R=1" > /dev/null 2>&1; cat $O/ab_chembl/summary.txt
tools/ab.sh -r 2 -o $O/ab_default "default" "bwd_x2r GI_CHAIN_BWD_X2R=1" > /dev/null 2>&1; cat $O/ab_default/summary.txt
