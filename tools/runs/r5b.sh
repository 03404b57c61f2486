cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests/test_x2_trial_gpu.py -q -s --maxfail=20 > $O/trial.log 2>&1; echo "trial rc $?"; tail -3 $O/trial.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_attggnn_gpu.py tests/test_golden_shapes_gpu.py tests/test_dp_gpu.py -q --maxfail=10 > $O/model.log 2>&1; echo "model rc $?"; tail -3 $O/model.log
tools/ab.sh -r 4 -o /root/repo/$O/ab "new" "nobias GI_WGRAD_BIAS=0" "noprepack GI_PREPACK=0" "noxcd GI_CHAIN_XCD=0" "r4like GI_PREPACK=0 GI_CHAIN_XCD=0 GI_WGRAD_BIAS=0" > $O/ab.log 2>&1
cat $O/ab/summary.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --no-forward-only --no-probe --no-one-stream --steps 20 --warmup 5 > /root/repo/$O/rocprof.log 2>&1
cd /root/repo; python tools/trace_summary.py $O/stats/*kernel_trace.csv > $O/trace_summary.txt 2>&1 || true
head -45 $O/stats/*kernel_stats.csv | cut -c1-150
python - <<'P'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/r5b/stats/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# keep the last ~2 steps
ad=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo=ad[-4]; hi=ad[-2]
keep=rows[lo:hi+1]
w=csv.DictWriter(open('/root/repo/gpurun_out/r5b/trace_default.csv','w'),fieldnames=['Kernel_Name','Queue_Id','Stream_Id','Start_Timestamp','End_Timestamp','Grid_Size_X','Workgroup_Size_X','LDS_Block_Size','VGPR_Count'])
w.writeheader(); t0=int(keep[0]['Start_Timestamp'])
for r in keep:
    w.writerow({'Kernel_Name':r['Kernel_Name'],'Queue_Id':r['Queue_Id'],'Stream_Id':r.get('Stream_Id',0),'Start_Timestamp':int(r['Start_Timestamp'])-t0,'End_Timestamp':int(r['End_Timestamp'])-t0,'Grid_Size_X':r['Grid_Size_X'],'Workgroup_Size_X':r['Workgroup_Size_X'],'LDS_Block_Size':r.get('LDS_Block_Size',0),'VGPR_Count':r.get('VGPR_Count',0)})
P
rm -f $O/stats/*kernel_trace.csv
