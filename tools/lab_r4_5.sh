cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
{
tools/mfma_fill
for L in tools/gemm_lab tools/gemm_lab_noslp; do
echo "== $L"
GI_B3P=0 GI_B3V=1 $L fwd3f 1 1 0
GI_B3P=0 GI_B3V=0 $L fwd3f 1 1 0
GI_LAB_N=5 GI_LAB_WMUL=3 $L wgrad3 1 1 0
$L fwd 2 2 0
GI_LAB_N=5 $L wgrad 1 1 0
done
} 2>&1 | tee $O/lab5.txt
