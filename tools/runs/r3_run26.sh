#!/bin/bash
# traces + reports, profile collection (PMC passes on training steps only), quick check that the microbenchmark tools run
cd /root/repo; export TMPDIR=/tmp
bash tools/collect_traces.sh r03 2>&1 | tail -70
bash tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1; tail -3 gpurun_out/r03/collect.log
cat gpurun_out/r03/traffic.json
(timeout 120 python tools/bench_gemm.py 2>&1 | tail -8) > gpurun_out/r03/bench_gemm.txt; tail -8 gpurun_out/r03/bench_gemm.txt
(timeout 120 python tools/bench_chain.py 2>&1 | tail -8) > gpurun_out/r03/bench_chain.txt; tail -8 gpurun_out/r03/bench_chain.txt
