cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5ag; mkdir -p $O
for cls in fwd1 dgrad1; do for M in 64 512 2048 4096 7258 16384 32768; do echo "== $cls fp32 M=$M"; GI_LAB_M=$M GI_LAB_N=1 timeout 60 tools/gemm_lab $cls 1 1 2>&1 | tail -2; done; done > $O/gemm_scaling.txt 2>&1
for M in 128 512 2048 4096 7258 16384 32768; do echo "== fwd13f fp16x2 M=$M"; GI_LAB_X2=1 GI_LAB_M=$M GI_LAB_N=1 timeout 60 tools/gemm_lab fwd13f 1 1 2>&1 | tail -2; done >> $O/gemm_scaling.txt 2>&1
for M in 128 512 2048 4096 7258 16384 32768; do echo "== fwd13f bf16x3 M=$M"; GI_LAB_M=$M GI_LAB_N=1 timeout 60 tools/gemm_lab fwd13f 1 1 2>&1 | tail -2; done >> $O/gemm_scaling.txt 2>&1
cat $O/gemm_scaling.txt
