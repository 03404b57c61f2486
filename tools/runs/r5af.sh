cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5af; mkdir -p $O
timeout 600 python -m pytest tests/test_x2_trial_gpu.py -q -s > $O/trial.log 2>&1; tail -4 $O/trial.log; grep "trained checkpoint" $O/trial.log | cut -c1-900
