"""Per-launch-class efficiency of the GEMM family with every launch ALONE on the device:

    python tools/gemm_class_report.py profiles/r03/trace_onestream.csv profiles/r03/gemm_launch_log.txt

Joins the `GI_GEMM_LOG` launch log (layout class, problems, useful FLOP, M x N x K per problem; one line per
gi_gemm / gi_gemm_batch launch in launch order) with the kernel durations of a trace taken with
`bench.py --one-stream` (tools/collect_traces.sh).  Forward (class 00) and dgrad (class 01) launches are matched one
to one by their order inside a step — their sequence does not depend on the stream schedule; weight-gradient
launches (class 11) are batched differently with and without the side stream, so they are reported as a total.
TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak (`frac`) and against the fp32-equivalent peak of the pipe the launch
ran on (`own`: 157.3 fp32 MFMA, 2382 / 6 = 397 bf16x3, 2382 / 3 = 794 fp16x2)."""
import csv
import sys

PEAK = 157.3
OWN = {"": 157.3, "bf16x3": 2382.0 / 6, "fp16x2": 2382.0 / 3}


def log_steps(path):
    """launch lines grouped into steps (a step starts with the pass-0 edge-count GEMM, class 01, 1 problem)."""
    steps, cur = [], None
    for line in open(path):
        f = line.split()
        if len(f) < 5:
            continue
        rec = dict(cls=f[0], nprob=int(f[1]), blocks=int(f[2]), flop=float(f[3]), probs=f[4:])
        # bf16x3 launches: "b0" / "b1" / "b2" (gi_gemm_bf3.hip, gi_gemm_b3v.hip), "p0" / "p1" / "p2" (gi_gemm_b3p.hip):
        # forward / dgrad / weight-gradient layout
        # fp16x2 launches: "x0" / "x1" (gi_gemm_bf3.hip), "y0" / "y1" / "y2" (gi_gemm_b3p.hip)
        rec["pipe"] = "bf16x3" if rec["cls"][0] in "bp" else ("fp16x2" if rec["cls"][0] in "xy" else "")
        rec["bf3"] = bool(rec["pipe"])
        if rec["bf3"]:
            rec["cls"] = {"0": "00", "1": "01", "2": "11"}[rec["cls"][1]]
        first = rec["cls"] == "01" and rec["nprob"] == 1 and rec["probs"][0].split(":")[0].endswith("x45") \
            and cur is not None and any(r["cls"] == "11" for r in cur)
        if cur is None or first:
            cur = []
            steps.append(cur)
        cur.append(rec)
    return steps


def trace_step(path, step=0):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    st = rows[adam[step] + 1:adam[step + 1]]
    return [r for r in st if r["Kernel_Name"].startswith(("gi_gemm", "gi_b3p", "gi_b3v"))]


def label(rec):
    dims = [tuple(int(x) for x in p.split(":")[0].split("x")) for p in rec["probs"]]
    m = max(d[0] for d in dims)
    k = max(d[2] for d in dims)
    n = max(d[1] for d in dims)
    what = {"00": "forward", "01": "dgrad  ", "11": "wgrad  "}[rec["cls"]]
    if rec.get("bf3"):
        what = what.rstrip() + " " + rec["pipe"]
    if rec["cls"] == "01" and k <= 64:
        what = "agg p0 "                               # pass-0 aggregation cmat . m0 (same operand layouts)
    return "%s  %d problem(s), rows %d, N <= %d, K <= %d" % (what, rec["nprob"], m, n, k)


def main():
    trace, log = sys.argv[1], sys.argv[2]
    steps = [s for s in log_steps(log) if any(r["cls"] == "11" for r in s)]
    recs = steps[-1]                                   # a complete step of the log
    launches = trace_step(trace)
    def kind(name):
        if "gi_gemm_bf3_kernel" in name:                        # <1, ..>: bias + SELU epilogue = forward; <2 / 0, ..>: dgrad
            return "00" if "gi_gemm_bf3_kernel<1" in name else "01"
        if "gi_b3p_kernel" in name or "gi_b3v_kernel" in name:  # <A_MAJOR, B_MAJOR, EPI>
            if "<true, true" in name:
                return "11"
            return "00" if ", 1>" in name else "01"
        return "11" if "true, true" in name else ("01" if "false, true" in name else "00")
    by_cls = {c: [r for r in launches if kind(r["Kernel_Name"]) == c] for c in ("00", "01", "11")}
    out = []
    for c in ("00", "01"):
        lg = [r for r in recs if r["cls"] == c]
        tr = by_cls[c]
        assert len(lg) == len(tr), (c, len(lg), len(tr))
        for a, b in zip(lg, tr):
            us = b["e"] - b["s"]
            out.append((label(a), a["flop"], us, OWN[a["pipe"]]))
    print("%-58s %9s %8s %8s %6s %6s" % ("launch", "GFLOP", "us", "TFLOP/s", "frac", "own"))
    for lab, flop, us, own in out:
        tf = flop / us / 1e6
        print("%-58s %9.3f %8.1f %8.1f %6.2f %6.2f" % (lab, flop / 1e9, us, tf, tf / PEAK, tf / own))
    wf = sum(r["flop"] for r in recs if r["cls"] == "11")
    wus = sum(r["e"] - r["s"] for r in by_cls["11"])
    print("%-58s %9.3f %8.1f %8.1f %6.2f" % ("wgrad    all batches of the step", wf / 1e9, wus, wf / wus / 1e6,
                                           wf / wus / 1e6 / PEAK))
    w3f = sum(r["flop"] for r in recs if r["cls"] == "11" and r.get("bf3"))
    w3us = sum(r["e"] - r["s"] for r in by_cls["11"] if "gi_b3" in r["Kernel_Name"])
    if w3us > 0:
        pipes = {r["pipe"] for r in recs if r["cls"] == "11" and r.get("bf3")}
        own = OWN[pipes.pop()] if len(pipes) == 1 else OWN["bf16x3"]
        print("%-58s %9.3f %8.1f %8.1f %6.2f %6.2f" % ("  of which on the 16-bit pipe (%s)" % ("fp16x2" if own > 500 else "bf16x3"),
                                                     w3f / 1e9, w3us, w3f / w3us / 1e6, w3f / w3us / 1e6 / PEAK, w3f / w3us / 1e6 / own))
        print("%-58s %9.3f %8.1f %8.1f %6.2f" % ("  of which on the fp32 MFMA", (wf - w3f) / 1e9, wus - w3us,
                                               (wf - w3f) / (wus - w3us) / 1e6, (wf - w3f) / (wus - w3us) / 1e6 / PEAK))
    tot_f = sum(f for _, f, _, _ in out) + wf
    tot_us = sum(u for _, _, u, _ in out) + wus
    print("%-58s %9.3f %8.1f %8.1f %6.2f" % ("gi_gemm / gi_gemm_batch launches (chains not logged)", tot_f / 1e9,
                                           tot_us, tot_f / tot_us / 1e6, tot_f / tot_us / 1e6 / PEAK))


if __name__ == "__main__":
    main()
