"""Per-launch GEMM efficiency: joins the host-side launch log (GI_GEMM_LOG) with a rocprofv3
kernel-trace CSV by launch order.

    GI_GEMM_LOG=/tmp/gemm.log rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- \
        python bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/gemm_launch_report.py /tmp/gemm.log /tmp/kt/*kernel_trace.csv <launches_per_step>
Prints the last full step: one line per launch with duration, useful TFLOP/s and the problems."""
import csv
import sys

log = [l.split() for l in open(sys.argv[1]) if l.strip()]
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "gi_gemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per_step = int(sys.argv[3])
n = min(len(log), len(rows))
log, rows = log[:n], rows[:n]
assert n >= per_step, (len(log), len(rows))
# the profiled extra steps come after the timed ones: take the last complete step
lo = n - per_step
tot_us = tot_fl = 0.0
by_class = {}
print("%3s %-3s %2s %6s %8s %7s  %s" % ("#", "lay", "np", "blocks", "us", "TF/s", "problems MxNxK:groups:splits"))
for i in range(lo, n):
    lay, nprob, blocks, flops = log[i][0], int(log[i][1]), int(log[i][2]), float(log[i][3])
    us = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    tot_us += us; tot_fl += flops
    c = by_class.setdefault({"00": "fwd", "01": "dgrad", "11": "wgrad"}[lay], [0.0, 0.0, 0])
    c[0] += us; c[1] += flops; c[2] += 1
    print("%3d %-3s %2d %6d %8.1f %7.1f  %s" % (i - lo, lay, nprob, blocks, us, flops / us / 1e6,
                                                " ".join(log[i][4:])))
print("step total: %.1f us, %.1f GFLOP, %.1f TF/s" % (tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6))
for k, (us, fl, cnt) in by_class.items():
    print("  %-6s %3d launches %8.1f us  %6.1f TF/s" % (k, cnt, us, fl / us / 1e6))
