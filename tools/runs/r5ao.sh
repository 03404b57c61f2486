cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5ao; mkdir -p $O
timeout 300 python -m pytest tests/test_x2_trial_gpu.py -q -x -k "guard_trips" > $O/g.log 2>&1; tail -15 $O/g.log
