cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5n; mkdir -p $O
tools/ab.sh -r 3 -o /root/repo/$O/ab "new" "lw2 GI_P0_LAYERWISE=2" "lw1 GI_P0_LAYERWISE=1" "lw3 GI_P0_LAYERWISE=3" > $O/ab.log 2>&1; cat $O/ab/summary.txt
