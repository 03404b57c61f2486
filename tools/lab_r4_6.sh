cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
tools/b3p_lab_trace 7258 $O/pipe_trace.txt; cat $O/pipe_trace.txt
bash tools/pmc_kernel.sh gi_b3p $O/pmc_b3p_pipelined.txt -- $GRAFT_REPO_ROOT/tools/b3p_lab 7258 > /dev/null; cat $O/pmc_b3p_pipelined.txt
bash tools/pmc_kernel.sh gi_b3p $O/pmc_b3p_pipelined_26k.txt -- $GRAFT_REPO_ROOT/tools/b3p_lab 26000 > /dev/null; cat $O/pmc_b3p_pipelined_26k.txt
