cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -6 $O/gpu_suite.log
