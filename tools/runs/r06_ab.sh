# Round 6, the A/B legs behind profiles/r06/ab_*.txt (tools/ab.sh; each line of a summary = one bench.py run on the box):
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r06_ab; mkdir -p $O
Z="--shape zinc --batch 1000 --steps 10 --warmup 3"; C="--shape chembl --model attggnn --batch 250 --steps 10 --warmup 3"
# the message / energy stacks' weight gradients from the chain kernels' amax cells (default) against the fp32 MFMA, and EVERY weight gradient as fp16x2
tools/ab.sh -r 3 -o $O/head_msg "default" "no_msg_cells GI_MSG_WGRAD_X2=0" "x2_all GI_WGRAD_X2_ALL=1" "x2_all_t128 GI_WGRAD_X2_ALL=1 GI_WGRAD_T128=1" > /dev/null 2>&1; cat $O/head_msg/summary.txt
tools/ab.sh -r 2 -o $O/zinc_msg -a "$Z" "default" "no_msg_cells GI_MSG_WGRAD_X2=0" "x2_all GI_WGRAD_X2_ALL=1" > /dev/null 2>&1; cat $O/zinc_msg/summary.txt
tools/ab.sh -r 2 -o $O/chembl_msg -a "$C" "default" "no_msg_cells GI_MSG_WGRAD_X2=0" "x2_all GI_WGRAD_X2_ALL=1" > /dev/null 2>&1; cat $O/chembl_msg/summary.txt
# slab height of those launches; the fp16x2 forward / dgrad launches on the 32-deep-tile kernel
tools/ab.sh -r 2 -o $O/head_slab "default" "slab230 GI_MSG_SLAB_ROWS=230" "slab920 GI_MSG_SLAB_ROWS=920" "b3v_x2_fwd GI_B3V_X2_FWD=1" > /dev/null 2>&1; cat $O/head_slab/summary.txt
