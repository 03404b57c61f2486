cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
L=$GRAFT_REPO_ROOT/tools/gemm_lab
GI_B3V=1 bash tools/pmc_kernel.sh gi_b3v $O/pmc_fwd3f_b3v.txt -- $L fwd3f 1 1 0
GI_B3V=0 bash tools/pmc_kernel.sh gi_gemm_bf3 $O/pmc_fwd3f_old.txt -- $L fwd3f 1 1 0
GI_LAB_N=5 GI_LAB_WMUL=3 bash tools/pmc_kernel.sh gi_b3v $O/pmc_wgrad3.txt -- $L wgrad3 1 1 0
GI_LAB_M=26000 bash tools/pmc_kernel.sh gi_b3v $O/pmc_fwd3f_b3v_26k.txt -- $L fwd3f 1 1 0
