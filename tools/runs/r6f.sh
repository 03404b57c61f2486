cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6f; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dims_gpu.py tests/test_attggnn_gpu.py -q -x > $O/model.log 2>&1; echo "model rc $?"; tail -3 $O/model.log
GI_WGRAD_X2_ALL=1 timeout 900 python -m pytest tests/test_dims_gpu.py tests/test_attggnn_gpu.py -q -x > $O/model_all.log 2>&1; echo "model(all) rc $?"; tail -3 $O/model_all.log
tools/ab.sh -r 3 -o $O/ab "default" "no_msg_cells GI_MSG_WGRAD_X2=0" > /dev/null 2>&1; cat $O/ab/summary.txt
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "default" "no_msg_cells GI_MSG_WGRAD_X2=0" "x2_all GI_WGRAD_X2_ALL=1" > /dev/null 2>&1; cat $O/ab_zinc/summary.txt
tools/ab.sh -r 2 -o $O/ab_chembl -a "--shape chembl --model attggnn --batch 250 --steps 10 --warmup 3" "default" "no_msg_cells GI_MSG_WGRAD_X2=0" "x2_all GI_WGRAD_X2_ALL=1" > /dev/null 2>&1; cat $O/ab_chembl/summary.txt
