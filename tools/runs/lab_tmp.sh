tools/ab.sh -r 3 -o gpurun_out/ab_first2 "both" "off GI_FIRST_X2=0" "fwd GI_FIRST_X2=2" "wgrad GI_FIRST_X2=3"
