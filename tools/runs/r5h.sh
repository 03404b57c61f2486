cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5h; mkdir -p $O
for v in "4 1" "4 0" "8 1" "8 0" "2 1" "4 1" "8 1"; do set -- $v; echo "U=$1 NT=$2: $(GI_SEGSUM_U=$1 GI_SEGSUM_NT=$2 python bench.py --probe-only 2>/dev/null | tail -1)"; done > $O/segsum_probe.txt 2>&1; cat $O/segsum_probe.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "seg_sum or gru_gates" > $O/k.log 2>&1; tail -2 $O/k.log
