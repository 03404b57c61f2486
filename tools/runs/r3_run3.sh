#!/bin/bash
# round 3, GPU call 3: s_setprio outside the main loop / bounds-check-free interior epilogue (lab variants)
OUT=/root/repo/gpurun_out/r3_run3; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
L=$OUT/lab.txt; : > $L
for cls in "fwd 1 1" "fwd 2 2" "dgrad 1 1" "wgrad 1 1" "tier2 1 1"; do
  for ps in 0 11; do
    for v in "" _prio1 _prio _fast _fastprio; do echo -n "lab$v " >> $L; timeout 60 tools/gemm_lab$v $cls $ps >> $L 2>&1; done
  done
done
timeout 60 tools/gemm_lab_trace fwd 1 1 0 $OUT/trace_fwd_11_p0.csv >> $L 2>&1
timeout 60 tools/gemm_lab_trace fwd 1 1 11 $OUT/trace_fwd_11_p11.csv >> $L 2>&1
cat $L
