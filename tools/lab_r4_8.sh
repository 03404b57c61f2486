cd $GRAFT_REPO_ROOT
L=tools/gemm_lab_x2
echo "TIMING ONLY (wrong results expected): three MFMA products instead of six, same staging"
GI_B3P=0 GI_B3V=0 $L fwd3f 1 1 0
GI_B3P=0 GI_B3V=0 $L dgrad3f 1 1 0
GI_B3P=1 GI_B3P_ALL=1 $L fwd3f 1 1 0
GI_LAB_N=5 GI_LAB_NSPLIT_ALL=9 GI_B3P=1 $L wgrad3 1 1 0
GI_LAB_M=26000 GI_B3P=1 $L fwd3f 1 1 0
GI_LAB_M=26000 GI_B3P=0 GI_B3V=0 $L fwd3f 1 1 0
