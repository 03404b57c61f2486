#!/bin/bash
# Weak-scaling run of bench.py on ONE node with N = 1, 2, 4, 8 ranks (one rank per GPU, backend nccl = RCCL over
# xGMI), exactly as the driver launches it; one JSON line per N -> $OUT/scale_N.json, and a summary with the whole-job
# graphs/s, the step time and the exposed milliseconds of the gradient exchange (allreduce.exposed_ms_overlapped: product
# path, readout tail exchanged under the message passes' backward; exposed_ms_after_backward: one all-reduce behind
# the backward).  Prediction to judge the first real curve against: DESIGN.md section 7.
#   tools/scale.sh [OUT_DIR] [extra bench.py args...]        (needs the GPUs; N is capped at the visible device count)
# SCALE_BACKEND=gloo: the same four launches with the ranks SHARING the visible device(s) over gloo — a dry run of this
# script's control flow on a 1-GPU box (tests/test_bench_gpu.py), not a measurement; N is then not capped.
OUT=${1:-gpurun_out/scale}; shift
BACKEND=${SCALE_BACKEND:-nccl}
mkdir -p $OUT
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NDEV=$(python -c 'import torch; print(torch.cuda.device_count())')
for N in 1 2 4 8; do
  [ "$BACKEND" = nccl ] && [ "$N" -gt "$NDEV" ] && { echo "N=$N skipped: $NDEV device(s) visible"; continue; }
  ARGS="--gpus $N --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-extra-configs --no-forward-only --no-one-stream $*"
  if [ "$N" = 1 ]; then python bench.py $ARGS > $OUT/scale_$N.log 2>&1
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((${SCALE_PORT:-29500} + N)) bench.py --backend $BACKEND $ARGS > $OUT/scale_$N.log 2>&1
  fi
  grep '^{"metric"' $OUT/scale_$N.log > $OUT/scale_$N.json || { echo "N=$N FAILED"; tail -5 $OUT/scale_$N.log; continue; }
  python - $OUT/scale_$N.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
ar = d.get("allreduce")
print("N=%d  %.0f graphs/s  %.3f ms/step" % (d["n_gpus"], d["value"], d["ms_per_step"]) +
      ("" if not isinstance(ar, dict) else "  exchange exposed %.3f ms overlapped / %.3f ms after the backward (bucket %.1f MB, ranks %d)"
       % (ar["exposed_ms_overlapped"], ar["exposed_ms_after_backward"], ar["bucket_MB"], ar["ranks_seen"])))
PY
done
