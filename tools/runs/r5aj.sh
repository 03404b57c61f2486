cd /tmp; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5aj; mkdir -p $O
: > $O/x2_chain_breakdown_final.txt
for m in 0 1 2 3 4 7 8 16 24; do
  rm -rf /tmp/cb$m
  GI_DBG_X2=$m BENCH_CHAIN_X2=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb$m -o b -- python /root/repo/tools/bench_chain.py bwd > /dev/null 2>&1
  echo "GI_DBG_X2=$m: $(grep 'gi_chain_x2_kernel' /tmp/cb$m/*kernel_stats.csv | cut -d, -f2-4)" >> $O/x2_chain_breakdown_final.txt
done
for m in 0 1 6 7 8 16 24 32 38 39; do
  rm -rf /tmp/cc$m
  GI_DBG_X2=$m BENCH_CHAIN_X2=1 BENCH_CHAIN_ROWS32=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cc$m -o b -- python /root/repo/tools/bench_chain.py fwd > /dev/null 2>&1
  echo "x2r forward, two per CU, GI_DBG_X2=$m: $(grep 'gi_chain_x2r_kernel' /tmp/cc$m/*kernel_stats.csv | cut -d, -f2-4)" >> $O/x2_chain_breakdown_final.txt
done
cat $O/x2_chain_breakdown_final.txt
