#!/bin/bash
OUT=/root/repo/gpurun_out/run6; mkdir -p $OUT; cd /root/repo
(timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "chain or gru" 2>&1 | tail -15) > $OUT/tests.log
python tools/bench_chain.py both 8400 >> $OUT/micro.txt 2>&1
python tools/bench_chain.py both 26000 >> $OUT/micro.txt 2>&1
python tools/trace_chain.py 8400 >> $OUT/micro.txt 2>&1
cat $OUT/tests.log $OUT/micro.txt
