#!/bin/bash
# final collection of the round on the final code (the suite of r3_run41 ran on this library)
cd /root/repo; export TMPDIR=/tmp
bash tools/collect_traces.sh r03 2>&1 | grep "bf16x3\|ms per step\|wgrad \|gi_gemm /\|step (from"
bash tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1; tail -1 gpurun_out/r03/collect.log
