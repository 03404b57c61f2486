"""Per-CU balance and per-workgroup phase report of a tools/gemm_lab_trace CSV (fwd / dgrad class).
usage: python tools/gemm_trace_report.py trace.csv tm tn"""
import csv, collections, sys
import numpy as np

def main():
    name, tm, tn = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rows = [{k: int(v) for k, v in r.items()} for r in csv.DictReader(open(name))]
    M = 7258
    nt500 = ((500 + 64 * tn - 1) // (64 * tn)) * ((M + 64 * tm - 1) // (64 * tm))
    for r in rows:
        r['units'] = (16 if r['wg'] < 2 * nt500 else 8) * tm * tn
        r['xcc'] = r['xcc_id'] & 15
    x0 = {}
    for r in rows: x0[r['xcc']] = min(x0.get(r['xcc'], 1 << 62), r['t_start'])
    cu = collections.defaultdict(list)
    for r in rows:
        hw = r['hw_id']; key = (r['xcc'], (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        r['s'] = r['t_start'] - x0[r['xcc']]; r['e'] = r['t_end'] - x0[r['xcc']]
        cu[key].append(r)
    units = np.array([sum(r['units'] for r in v) for v in cu.values()])
    ends = np.array([max(r['e'] for r in v) for v in cu.values()])
    print(name, "units/CU mean %.1f min %d max %d ; CU end mean %.0f min %d max %d" % (units.mean(), units.min(), units.max(), ends.mean(), ends.min(), ends.max()))
    print("  ideal cycles/CU = units*1024: mean %.0f max %.0f; kernel end max %d -> pipe util(mean) %.3f, balance (mean/max units) %.3f" % (units.mean() * 1024, units.max() * 1024, ends.max(), units.mean() * 1024 / ends.max(), units.mean() / units.max()))
    for x in sorted(x0):
        e = [max(r['e'] for r in v) for k, v in cu.items() if k[0] == x]; u = [sum(r['units'] for r in v) for k, v in cu.items() if k[0] == x]
        print("   xcd", x, "CUs", len(e), "end max", max(e), "min", min(e), "units sum", sum(u), "max", max(u), "min", min(u))
    k0 = sorted(cu)[5]; v = sorted(cu[k0], key=lambda r: r['s'])
    print("  sample CU", k0)
    for r in v:
        print("     wg %5d start %7d pro %6d loop %6d epi %6d end %7d units %d" % (r['wg'], r['s'], r['t_prologue'] - r['t_start'], r['t_loop'] - r['t_prologue'], r['t_end'] - r['t_loop'], r['e'], r['units']))

main()
