# Round 6: one case of tests/test_dims_gpu.py stage by stage against the CPU dataflow model, per arithmetic mode (tools/diag_dims.py)
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r06_dims; mkdir -p $O
for c in wide_h implicit_h wide_enn ggnn_n88; do for m in fp16x2 bf16x3 fp32; do timeout 300 python tools/diag_dims.py $c $m 2>&1 | grep -v "amdgpu.ids\|Warning\|Consider\|print(" > $O/diag_${c}_$m.log; grep -v "^pass\|^h " $O/diag_${c}_$m.log | head -8; done; done
