cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6d; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "mlp_chain or c_amax" > $O/k.log 2>&1; echo "kernels rc $?"; tail -15 $O/k.log
timeout 1200 python -m pytest tests/test_dims_gpu.py -q -s > $O/dims.log 2>&1; echo "dims rc $?"; grep -v "^  \|^    \|^$" $O/dims.log | tail -20
