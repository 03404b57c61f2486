cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5al; mkdir -p $O
tools/ab.sh -r 3 -o $O/ab_chembl -a "--model attggnn --shape chembl --batch 250 --steps 10 --warmup 3" "default" "dual0 GI_CHAIN_X2R_DUAL=0" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_chembl/summary.txt
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "default" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_zinc/summary.txt
