set -x
cd $GRAFT_REPO_ROOT
L=tools/gemm_lab
for v in 0 1; do
  GI_B3V=$v $L fwd3f 1 1 0
  GI_B3V=$v $L dgrad3f 1 1 0
done
$L dgrad3m 1 1 0
GI_LAB_N=5 $L wgrad 1 1 0
GI_LAB_N=5 $L wgrad 2 2 0
for m in 2 3 4 6; do GI_LAB_N=5 GI_LAB_WMUL=$m $L wgrad3 1 1 0; done
GI_LAB_FILL=0 $L fwd3f 1 1 0
GI_LAB_M=26000 $L fwd3f 1 1 0
GI_B3V=0 GI_LAB_M=26000 $L fwd3f 1 1 0
GI_LAB_M=26000 GI_LAB_N=5 GI_LAB_WMUL=2 $L wgrad3 1 1 0
GI_LAB_M=26000 GI_LAB_N=5 $L wgrad 1 1 0
