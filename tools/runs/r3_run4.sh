#!/bin/bash
# round 3, GPU call 4: is the fp32-MFMA GEMM power (DVFS) limited?  constant vs random operands
OUT=/root/repo/gpurun_out/r3_run4; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
L=$OUT/lab.txt; : > $L
tools/mfma_power >> $L 2>&1
for fill in 1 0 0.001; do
  for cls in "fwd 1 1" "fwd 2 2"; do echo -n "fill=$fill " >> $L; GI_LAB_FILL=$fill timeout 60 tools/gemm_lab $cls 0 >> $L 2>&1; done
done
cat $L
