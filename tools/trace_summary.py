"""Summarise a rocprofv3 kernel-trace CSV: GEMM launches grouped by (variant, grid), other kernels by name."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
g = collections.defaultdict(list); o = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', ''); d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    wg = int(r['Workgroup_Size_X'])
    if 'gi_gemm' in n:
        key = (n.replace('void gi_gemm_kernel', '').replace('(gi_gemm_params)', ''), int(r['Grid_Size_X']) // wg, int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
        g[key].append(d)
    else:
        o[n.split('(')[0][-60:]].append(d)
tot = sum(sum(v) for v in g.values())
print("%-22s %5s %5s %3s %6s %6s %8s %9s" % ("gemm", "gx", "gy", "gz", "waves", "n/st", "avg_us", "us/step"))
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%-22s %5d %5d %3d %6d %6.1f %8.1f %9.1f" % (k[0], k[1], k[2], k[3], k[1] * k[2] * k[3], len(v) / steps, sum(v) / len(v), sum(v) / steps))
print("GEMM total us/step", round(tot / steps, 1), "launches/step", sum(len(v) for v in g.values()) / steps)
print("--- other kernels (us/step)")
for k, v in sorted(o.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-62s n/st %5.1f avg %7.1f  us/step %8.1f" % (k, len(v) / steps, sum(v) / len(v), sum(v) / steps))
print("other total us/step", round(sum(sum(v) for v in o.values()) / steps, 1))
