cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
L=$GRAFT_REPO_ROOT/tools/gemm_lab
{
for r in 1 2; do
GI_B3P=1 $L fwd3f 1 1 0
GI_B3P=0 GI_B3V=0 $L fwd3f 1 1 0
done
GI_B3P=1 $L dgrad3f 1 1 0
GI_B3P=1 GI_LAB_FILL=0 $L fwd3f 1 1 0
GI_B3P=1 GI_LAB_M=26000 $L fwd3f 1 1 0
GI_B3P=0 GI_B3V=0 GI_LAB_M=26000 $L fwd3f 1 1 0
GI_B3P=1 GI_LAB_M=32768 $L fwd13f 1 1 0
GI_B3P=0 GI_B3V=0 GI_LAB_M=32768 $L fwd13f 1 1 0
} 2>&1 | tee $O/lab4.txt
GI_B3P=1 bash tools/pmc_kernel.sh gi_b3p $O/pmc_fwd3f_b3p2.txt -- $L fwd3f 1 1 0 > /dev/null
cat $O/pmc_fwd3f_b3p2.txt
