cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5ae; mkdir -p $O
tools/ab.sh -r 3 -o $O/ab_default "default" "p0_bwd_x2 GI_CHAIN_BWD_X2R=0" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_default/summary.txt
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "default" "p0_bwd_x2 GI_CHAIN_BWD_X2R=0" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_zinc/summary.txt
tools/ab.sh -r 2 -o $O/ab_chembl -a "--model attggnn --shape chembl --batch 250 --steps 10 --warmup 3" "default" "p0_bwd_x2 GI_CHAIN_BWD_X2R=0" "fwd_fp32 GI_CHAIN_FWD_X2=0" > /dev/null 2>&1; cat $O/ab_chembl/summary.txt
timeout 500 python -m pytest tests/test_model_gpu.py -q -x > $O/model.log 2>&1; tail -4 $O/model.log
timeout 300 python -m pytest tests/test_x2_trial_gpu.py -q -x -k "trained or guard" > $O/trial.log 2>&1; tail -4 $O/trial.log
