#!/bin/bash
cd /root/repo; export LD_LIBRARY_PATH=/root/repo/graphinvent_amd:$LD_LIBRARY_PATH
L=tools/gemm_lab
for m in 2800 8700; do
for c in fwd dgrad; do GI_LAB_SQ=250 GI_LAB_M=$m $L $c 1 1 0; done
for c in fwd3f dgrad3f; do
  GI_LAB_SQ=250 GI_LAB_M=$m GI_LAB_X2=1 GI_B3P=0 GI_B3V=0 $L $c 1 1 0
  GI_LAB_SQ=250 GI_LAB_M=$m GI_LAB_X2=0 GI_B3P=0 GI_B3V=0 $L $c 1 1 0
done
done
