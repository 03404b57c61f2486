#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_syncfree_gpu.py tests/test_p0cache_gpu.py tests/test_dp_gpu.py tests/test_bench_gpu.py -m gpu -q 2>&1 | grep "passed\|failed"
