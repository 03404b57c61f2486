cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5c; mkdir -p $O
{
echo "== first layers forward: 4 problems 7258 x {250x136, 500x128, 500x128, 250x128}"
GI_LAB_DIMS="500x128,500x128,250x136,250x128" tools/gemm_lab fwd 1 1 0
GI_LAB_DIMS="500x128,500x128,250x136,250x128" GI_B3V=0 tools/gemm_lab fwd3f 1 1 0
GI_LAB_DIMS="500x128,500x128,250x136,250x128" GI_B3V=0 GI_LAB_X2=1 tools/gemm_lab fwd3f 1 1 0
echo "== GRU projections forward: 2 problems 7258 x 384x128"
GI_LAB_DIMS="384x128,384x128" tools/gemm_lab fwd 1 1 0
GI_LAB_DIMS="384x128,384x128" GI_B3V=0 GI_LAB_X2=1 tools/gemm_lab fwd3f 1 1 0
echo "== first layers dgrad: 4 problems 7258 x {128x500, 128x500, 128x250, 128x250}"
GI_LAB_DIMS="128x500,128x500,128x250,128x250" tools/gemm_lab dgrad 1 1 0
GI_LAB_DIMS="128x500,128x500,128x250,128x250" GI_B3V=0 GI_LAB_X2=1 tools/gemm_lab dgrad3f 1 1 0
echo "== GRU dgrad: 2 problems 7258 x 128x384"
GI_LAB_DIMS="128x384,128x384" tools/gemm_lab dgrad 1 1 0
GI_LAB_DIMS="128x384,128x384" GI_B3V=0 GI_LAB_X2=1 tools/gemm_lab dgrad3f 1 1 0
echo "== last layers forward: 7258 x {45x500, 3x500, 100x250, 100x250}"
GI_LAB_DIMS="45x500,3x500,100x250,100x250" tools/gemm_lab fwd 1 1 0
GI_LAB_DIMS="45x500,3x500,100x250,100x250" GI_B3V=0 GI_LAB_X2=1 tools/gemm_lab fwd3f 1 1 0
echo "== graph level hidden: 3 x 1000 x 500x500"
tools/gemm_lab tier2 1 1 0
GI_B3V=0 GI_LAB_X2=1 tools/gemm_lab tier23f 1 1 0
} > $O/lab.txt 2>&1
cat $O/lab.txt
timeout 900 python -m pytest tests/test_x2_trial_gpu.py -q -s --maxfail=20 > $O/trial.log 2>&1; echo "trial rc $?"; grep -n "trained checkpoint\|^E  \|passed\|failed" $O/trial.log | cut -c1-700
timeout 900 python -m pytest tests/test_model_gpu.py::test_reinforcement_learning_call_pattern_several_forwards_one_backward tests/test_bench_gpu.py::test_scale_script_dry_run_over_gloo -q -s > $O/new_tests.log 2>&1; echo "new tests rc $?"; tail -4 $O/new_tests.log; grep "RL pattern" $O/new_tests.log
tools/ab.sh -r 3 -o /root/repo/$O/ab "new" "r4like GI_PREPACK=0 GI_CHAIN_XCD=0" "noxcd GI_CHAIN_XCD=0" > $O/ab.log 2>&1; cat $O/ab/summary.txt
tools/ab.sh -r 2 -o /root/repo/$O/abz -a "--shape zinc --batch 1000 --steps 10 --warmup 3" "new" "r4like GI_PREPACK=0 GI_CHAIN_XCD=0" "noxcd GI_CHAIN_XCD=0" > $O/abz.log 2>&1; cat $O/abz/summary.txt
tools/ab.sh -r 2 -o /root/repo/$O/abc -a "--shape chembl --model attggnn --batch 250 --steps 10 --warmup 3" "new" "r4like GI_PREPACK=0 GI_CHAIN_XCD=0" "noxcd GI_CHAIN_XCD=0" > $O/abc.log 2>&1; cat $O/abc/summary.txt
