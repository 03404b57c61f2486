#!/bin/bash
# round 4: fp16x2 in the model after the fast absmax kernel — A/B at the headline and ZINC shapes, ZINC traces of both
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
tools/ab.sh -r 2 -o $O/ab_head "x2" "bf16x3 GI_X2=0" "r3 GI_B3P=0 GI_X2=0" > $O/ab_head.txt 2>&1
tools/ab.sh -r 2 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "x2" "bf16x3 GI_X2=0" "r3 GI_B3P=0 GI_X2=0" > $O/ab_zinc.txt 2>&1
cd /tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-forward-only --no-probe --no-one-stream --steps 8 --warmup 3 --shape zinc --batch 1000"
for v in x2 b3; do
  rm -rf /tmp/tz_$v
  if [ $v = x2 ]; then E="GI_X2=1"; else E="GI_X2=0"; fi
  env $E timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tz_$v -o t -- $B > /tmp/tz_$v.log 2>&1
  head -40 /tmp/tz_$v/t_kernel_stats.csv > /root/repo/$O/zinc_stats_$v.csv
done
cd /root/repo
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_golden_shapes_gpu.py tests/test_attggnn_gpu.py tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | tail -8 > $O/pytest.txt
cat $O/ab_head.txt $O/ab_zinc.txt $O/pytest.txt
