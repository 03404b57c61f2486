cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "weight_gradient or absmax or c_amax or fp16x2" > $O/k.log 2>&1; echo "kernels rc $?"; tail -5 $O/k.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dims_gpu.py tests/test_attggnn_gpu.py -q -x > $O/model.log 2>&1; echo "model rc $?"; tail -5 $O/model.log
tools/ab.sh -r 2 -o $O/ab "default" "r4_set GI_WGRAD_X2_ALL=0" "t128_none GI_WGRAD_T128=0" "t128_all GI_WGRAD_T128=2" "slab920 GI_MSG_SLAB_ROWS=920" > /dev/null 2>&1; cat $O/ab/summary.txt
tools/collect_traces.sh r6e_tr > $O/traces.log 2>&1; tail -14 $O/traces.log
