# A/B of two builds of the library on one box: in-tree .so against graphinvent_amd/libgi_split4.so (-DGI_X2_SPLIT4 -fno-slp-vectorize)
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r6i; mkdir -p $O
cp graphinvent_amd/libgraphinvent_amd.so /tmp/base.so; cp graphinvent_amd/libgi_split4.so /tmp/split4.so
B="python bench.py --no-cpu-baseline --no-extra-configs --no-probe --no-forward-only --no-one-stream --steps 20 --warmup 5"
run() { cp /tmp/$1.so graphinvent_amd/libgraphinvent_amd.so; $B $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$1 $3', d['ms_per_step'], d['config']['loss'], d.get('bf16x3_only',{}).get('ms_per_step'))"; }
for rep in 1 2 3; do run base "" head; run split4 "" head; done | tee $O/ab_head.txt
for rep in 1 2; do run base "--shape zinc --batch 1000 --steps 10 --warmup 3" zinc; run split4 "--shape zinc --batch 1000 --steps 10 --warmup 3" zinc; done | tee $O/ab_zinc.txt
cp /tmp/split4.so graphinvent_amd/libgraphinvent_amd.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "fp16x2 or x2 or chain" 2>&1 | tail -3
cp /tmp/base.so graphinvent_amd/libgraphinvent_amd.so
