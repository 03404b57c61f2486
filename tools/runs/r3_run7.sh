#!/bin/bash
# round 3, GPU call 7: full suite on the pruned library + new parity tests + bench multi-rank test
OUT=/root/repo/gpurun_out/r3_run7; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $OUT/suite.log; tail -25 $OUT/suite.log
python bench.py --no-cpu-baseline --no-extra-configs --no-probe --steps 20 --warmup 5 2>$OUT/err.txt | tail -1 > $OUT/bench.json; cut -c1-600 $OUT/bench.json
