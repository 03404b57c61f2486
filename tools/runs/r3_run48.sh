#!/bin/bash
# the GPU suite and smoke() on the final build of the round
OUT=/root/repo/gpurun_out/r3_run48; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension" | tail -40) > $OUT/gpu_suite_final.log; grep "passed\|failed" $OUT/gpu_suite_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
