cd /tmp; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r5s; mkdir -p $O
: > $O/x2_chain_breakdown.txt
for m in 0 1 2 3 4 7 8 16 24; do
  rm -rf /tmp/cb$m
  GI_DBG_X2=$m BENCH_CHAIN_X2=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb$m -o b -- python /root/repo/tools/bench_chain.py bwd > /dev/null 2>&1
  echo "GI_DBG_X2=$m: $(grep 'gi_chain_x2_kernel' /tmp/cb$m/*kernel_stats.csv | cut -d, -f2-4)" >> $O/x2_chain_breakdown.txt
done
rm -rf /tmp/cbf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cbf -o b -- python /root/repo/tools/bench_chain.py both > /dev/null 2>&1
echo "fp32 chains: $(grep 'gi_chain_kernel' /tmp/cbf/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-140)" >> $O/x2_chain_breakdown.txt
cat $O/x2_chain_breakdown.txt
