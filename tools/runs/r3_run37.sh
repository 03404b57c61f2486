#!/bin/bash
# soak: the GPU suite twice more in fresh processes (flakiness check), in random test-file order the second time
cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep "passed\|failed\|Aborted\|rror" | head -5
timeout 1500 python -m pytest $(ls tests/test_*gpu*.py | sort -r) -m gpu -q 2>&1 | grep "passed\|failed\|Aborted\|rror" | head -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
