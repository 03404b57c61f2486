#!/bin/bash
# round 3, GPU call 1: GEMM lab baseline (hot batched launches, per-workgroup trace, SQ counters) + the
# experiments round 2 prepared (tools/ab/next_round.sh).
OUT=/root/repo/gpurun_out/r3_run1; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
L=$OUT/lab.txt; : > $L
for cfg in "fwd 1 1" "fwd 1 2" "fwd 2 2" "dgrad 1 1" "dgrad 2 2" "fwd1 1 1" "fwd1 2 2" "tier2 1 1"; do
  timeout 60 tools/gemm_lab $cfg >> $L 2>&1
done
timeout 60 tools/gemm_lab_trace fwd 1 1 $OUT/trace_fwd_11.csv >> $L 2>&1
timeout 60 tools/gemm_lab_trace fwd 2 2 $OUT/trace_fwd_22.csv >> $L 2>&1
timeout 60 tools/gemm_lab_trace tier2 1 1 $OUT/trace_tier2_11.csv >> $L 2>&1
cat $L
timeout 200 bash tools/pmc_kernel.sh gi_gemm $OUT/pmc_fwd_11.txt -- /root/repo/tools/gemm_lab fwd 1 1 > /dev/null 2>&1
timeout 200 bash tools/pmc_kernel.sh gi_gemm $OUT/pmc_fwd_22.txt -- /root/repo/tools/gemm_lab fwd 2 2 > /dev/null 2>&1
cat $OUT/pmc_fwd_11.txt $OUT/pmc_fwd_22.txt
timeout 600 bash tools/ab/next_round.sh > $OUT/next_round.log 2>&1
cp gpurun_out/next/summary.txt $OUT/next_summary.txt; cat $OUT/next_summary.txt
