cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5g; mkdir -p $O
for u in 1 2 4 2 4 1; do echo "U=$u: $(GI_SEGSUM_U=$u python bench.py --probe-only 2>/dev/null | tail -1)"; done > $O/segsum_probe.txt 2>&1; cat $O/segsum_probe.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "seg_sum or gru_gates or softmax" > $O/k.log 2>&1; tail -2 $O/k.log
GI_SEGSUM_U=4 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "seg_sum or gru_gates" > $O/k4.log 2>&1; tail -2 $O/k4.log
tools/ab.sh -r 2 -o /root/repo/$O/ab "u2" "u1 GI_SEGSUM_U=1" "u4 GI_SEGSUM_U=4" > $O/ab.log 2>&1; cat $O/ab/summary.txt
