#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3_run34
bash tools/pmc_kernel.sh gi_gemm_bf3_kernel /root/repo/gpurun_out/r3_run34/pmc_bf3_fwd.txt -- /root/repo/tools/gemm_lab fwd3 1 1 0 2>&1 | cut -c1-700
bash tools/pmc_kernel.sh gi_gemm_bf3_kernel /root/repo/gpurun_out/r3_run34/pmc_bf3_fwd1.txt -- /root/repo/tools/gemm_lab fwd13 1 1 0 2>&1 | cut -c1-700
bash tools/pmc_kernel.sh gi_gemm_tiles_kernel /root/repo/gpurun_out/r3_run34/pmc_fp32_fwd.txt -- /root/repo/tools/gemm_lab fwd 1 1 0 2>&1 | cut -c1-700
