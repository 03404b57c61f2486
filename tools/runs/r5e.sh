cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py tests/test_golden_shapes_gpu.py tests/test_dp_gpu.py tests/test_dropout_gpu.py -q --maxfail=10 > $O/model.log 2>&1; echo "model rc $?"; tail -3 $O/model.log
tools/ab.sh -r 3 -o /root/repo/$O/ab "new" "nobias GI_WGRAD_BIAS=0" "r4like GI_PREPACK=0 GI_CHAIN_XCD=0 GI_WGRAD_BIAS=0" > $O/ab.log 2>&1; cat $O/ab/summary.txt
tools/ab.sh -r 2 -o /root/repo/$O/abz -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "new" "nobias GI_WGRAD_BIAS=0" "r4like GI_PREPACK=0 GI_CHAIN_XCD=0 GI_WGRAD_BIAS=0" > $O/abz.log 2>&1; cat $O/abz/summary.txt
tools/ab.sh -r 2 -o /root/repo/$O/abc -a "--shape chembl --model attggnn --batch 250 --steps 10 --warmup 3" "new" "nobias GI_WGRAD_BIAS=0" "r4like GI_PREPACK=0 GI_CHAIN_XCD=0 GI_WGRAD_BIAS=0" > $O/abc.log 2>&1; cat $O/abc/summary.txt
timeout 900 python -m pytest tests/test_x2_trial_gpu.py -q -s --maxfail=20 > $O/trial.log 2>&1; echo "trial rc $?"; grep -n "trained checkpoint\|^E  \|passed\|failed" $O/trial.log | cut -c1-400
