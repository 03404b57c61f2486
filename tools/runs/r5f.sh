cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5f; mkdir -p $O
tools/ab.sh -r 2 -o /root/repo/$O/ab "new" "tn2_48 GI_WGRAD_TN=2 GI_WGRAD_WGS=48" "tn2_96 GI_WGRAD_TN=2 GI_WGRAD_WGS=96" "tn2_192 GI_WGRAD_TN=2" "w128 GI_WGRAD_WGS=128" "w256 GI_WGRAD_WGS=256" "w320 GI_WGRAD_WGS=320" > $O/ab.log 2>&1; cat $O/ab/summary.txt
