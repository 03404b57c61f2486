cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6m; mkdir -p $O
GI_B3V_X2_FWD=1 timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "bench_batch or golden or guard" 2>&1 | tail -3
GI_B3V_X2_FWD=1 timeout 900 python -m pytest tests/test_x2_trial_gpu.py -q -x -k "guard" 2>&1 | tail -3
tools/ab.sh -r 3 -o $O/ab "default" "b3v_x2_fwd GI_B3V_X2_FWD=1" > /dev/null 2>&1; cat $O/ab/summary.txt
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "default" "b3v_x2_fwd GI_B3V_X2_FWD=1" > /dev/null 2>&1; cat $O/ab_zinc/summary.txt
