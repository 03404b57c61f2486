cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6k; mkdir -p $O
tools/ab.sh -r 3 -o $O/ab "default" "lab_readout_late GI_LAB_READOUT_LATE=1" > /dev/null 2>&1; cat $O/ab/summary.txt
GI_LAB_READOUT_LATE=1 tools/collect_traces.sh r6k_tr > /dev/null 2>&1; cat gpurun_out/r6k_tr/trace_summary.txt; cat gpurun_out/r6k_tr/critical_path_default.txt
