cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6b; mkdir -p $O
for m in fp16x2 bf16x3 fp32; do timeout 300 python tools/diag_dims.py implicit_h $m 2>&1 | grep -v "amdgpu.ids\|Warning\|Consider\|print(" > $O/diag_implicit_h_$m.log; grep -v "^pass\|^h " $O/diag_implicit_h_$m.log; done
