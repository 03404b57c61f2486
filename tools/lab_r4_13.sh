#!/bin/bash
# round 4: the pass-0 rows' weight gradients in one slab, everything else kicked before the pass-0 chain
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "fp16x2" 2>&1 | tail -4 > $O/pytest_k.txt
tools/ab.sh -r 2 -o $O/ab_head "lw2" "lw0 GI_P0_LAYERWISE=0" "lw3 GI_P0_LAYERWISE=3" > $O/ab_head.txt 2>&1
tools/ab.sh -r 1 -o $O/ab_zinc -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "lw2" "lw0 GI_P0_LAYERWISE=0" > $O/ab_zinc.txt 2>&1
tools/ab.sh -r 1 -o $O/ab_att -a "--shape chembl --batch 250 --model attggnn --steps 10 --warmup 3" "lw2" "lw0 GI_P0_LAYERWISE=0" > $O/ab_att.txt 2>&1
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_golden_shapes_gpu.py tests/test_attggnn_gpu.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest.txt
tools/collect_traces.sh r4m > $O/traces.txt 2>&1
cat $O/pytest_k.txt $O/ab_head.txt $O/ab_zinc.txt $O/ab_att.txt $O/pytest.txt
