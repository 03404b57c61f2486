cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6c; mkdir -p $O
timeout 1200 python -m pytest tests/test_dims_gpu.py -q -s > $O/dims.log 2>&1; echo "dims rc $?"; grep -v "^  \|^    \|^$" $O/dims.log | tail -40
