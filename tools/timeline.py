"""Timeline of the last training step from a rocprofv3 kernel-trace CSV: start offset, duration,
queue and short kernel name for every launch (to see what overlaps what).
    python tools/timeline.py trace.csv <kernels_per_step_guess>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last compact_count_kernel launch onwards minus trailing stuff
idx = [i for i, r in enumerate(rows) if "compact_count_kernel" in r["Kernel_Name"]]
lo = idx[-1]
hi = len(rows) - 1
t0 = int(rows[lo]["Start_Timestamp"])
busy = {}
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:46]
    q = r.get("Queue_Id", "?")
    busy.setdefault(q, 0)
    busy[q] += e - s
    print("%9.1f %8.1f  q%-3s %s  grid %s" % (s / 1e3, (e - s) / 1e3, q, name, r["Grid_Size_X"]))
print("span us", (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3, "busy per queue us", {k: v / 1e3 for k, v in busy.items()})
