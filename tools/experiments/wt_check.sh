export GI_DGRAD_WT=1
timeout 500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_attggnn_gpu.py tests/test_dp_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  " | head -5
for i in 1 2; do GI_DGRAD_WT=1 python bench.py --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c136-162; GI_DGRAD_WT=0 python bench.py --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c136-162; done
