#!/bin/bash
# the multi-rank branch of bench.py with every leg on (as the driver's N > 1 runs have it), two ranks sharing the GPU over gloo
cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 --no-extra-configs --no-cpu-baseline --no-probe 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.load(sys.stdin); print(d['n_gpus'], d['ms_per_step'], d['allreduce']['ranks_seen'], d['roofline']['one_stream']['ms_per_step'], d['fp32_mfma_only'], d['forward_only']['ms_per_step'])"
