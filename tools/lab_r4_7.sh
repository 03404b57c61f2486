cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
L=tools/gemm_lab
{
for ns in 6 8 9 10 12 18; do GI_LAB_N=5 GI_LAB_NSPLIT_ALL=$ns GI_B3P=1 $L wgrad3 1 1 0; done
GI_LAB_N=5 GI_LAB_WMUL=3 GI_B3P=0 GI_B3V=1 $L wgrad3 1 1 0
GI_B3P=1 $L dgrad3m 1 1 0
GI_B3P=1 $L dgrad3f 1 1 0
GI_B3P=1 $L fwd3f 1 1 0
GI_B3P=0 GI_B3V=0 $L fwd3f 1 1 0
GI_LAB_M=26000 GI_LAB_N=5 GI_LAB_NSPLIT_ALL=9 GI_B3P=1 $L wgrad3 1 1 0
GI_LAB_M=26000 GI_LAB_N=5 GI_LAB_NSPLIT_ALL=18 GI_B3P=1 $L wgrad3 1 1 0
GI_LAB_M=26000 GI_B3P=1 $L fwd3f 1 1 0
} 2>&1 | tee $O/lab7.txt
