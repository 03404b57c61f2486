cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_x2_trial_gpu.py -q -s --maxfail=20 > gpurun_out/r5a/trial.log 2>&1; echo "trial rc $?" 
tail -5 gpurun_out/r5a/trial.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 --deselect tests/test_x2_trial_gpu.py > gpurun_out/r5a/suite.log 2>&1; echo "suite rc $?"
tail -5 gpurun_out/r5a/suite.log
tools/ab.sh -r 2 -o /root/repo/gpurun_out/r5a/ab "new" "noprepack GI_PREPACK=0" "noxcd GI_CHAIN_XCD=0" "nobias GI_WGRAD_BIAS=0" "r4like GI_PREPACK=0 GI_CHAIN_XCD=0 GI_WGRAD_BIAS=0" > gpurun_out/r5a/ab.log 2>&1
cat gpurun_out/r5a/ab/summary.txt
timeout 300 python tests/test_x2_trial_gpu.py > gpurun_out/r5a/x2_trial.txt 2>&1; tail -3 gpurun_out/r5a/x2_trial.txt
