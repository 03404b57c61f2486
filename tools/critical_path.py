"""Where one training step spends its time, from a (trimmed) rocprofv3 kernel trace as written by
tools/collect_traces.sh (columns Kernel_Name, Queue_Id, Start_Timestamp, End_Timestamp in ns):

    python tools/critical_path.py profiles/r03/trace_default.csv [step_index] [--list]

The step between two consecutive `adam_kernel` launches is split into the phases of the fused forward /
backward by structural markers (kernel names, not grid sizes): per phase the wall-clock span on the main
queue, the kernel time on the main queue inside it, and the kernel time of the weight-gradient side queue
that overlaps it.  `--list` also prints every launch of the step (start, duration, queue, name, grid)."""
import csv
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"] = int(r["Start_Timestamp"]) / 1e3
        r["e"] = int(r["End_Timestamp"]) / 1e3
    rows.sort(key=lambda r: r["s"])
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = load(args[0])
    step = int(args[1]) if len(args) > 1 else 0
    adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    lo, hi = adam[step] + 1, adam[step + 1] + 1
    main_q = rows[adam[step]]["Queue_Id"]
    st = rows[lo:hi]
    fill = next(r for r in st if "compact_fill" in r["Kernel_Name"] and r["Queue_Id"] == main_q)
    mq = [r for r in st if r["Queue_Id"] == main_q and r["s"] >= fill["s"]]
    # the second queue: weight-gradient GEMMs, slab reductions and (round 5) what the forward packs ahead for the backward
    # plus the fp16x2 layers' amax / dynamic-range passes — everything but the next batch's compaction prefetch
    prefetch = ("compact_", "copyBuffer", "fillBufferAligned", "FillFunctor")
    side = [r for r in st if r["Queue_Id"] != main_q and
            (not any(n in r["Kernel_Name"] for n in prefetch) or
             ("fillBufferAligned" in r["Kernel_Name"] and r["s"] >= fill["s"] and int(r["Grid_Size_X"]) > 1024))]
    t0 = fill["s"]

    def first(pred, after=0.0):
        return next(r for r in mq if pred(r["Kernel_Name"]) and r["s"] >= after)

    gates_f = [r for r in mq if "gru_gates_fwd" in r["Kernel_Name"] or "gru_fused_fwd" in r["Kernel_Name"]]   # (round 6: the fused update)
    gather_f = first(lambda n: "gather_fwd" in n)
    kl = first(lambda n: "kl_loss" in n)
    mean = first(lambda n: "mean_rows" in n)
    gather_b = first(lambda n: "gather_bwd" in n)
    colsum = first(lambda n: "colsum" in n)
    # the message passes' backward starts with the GRU gate backward of the last pass (round 4: with the dZ chains'
    # weight pack, which the forward now does ahead on the second queue)
    gates_b = [r for r in mq if "gru_gates_bwd" in r["Kernel_Name"]]
    packs = [r for r in mq if "chain_pack" in r["Kernel_Name"] and r["s"] > colsum["e"]]
    pack_b = packs[-1] if packs else (gates_b[0] if gates_b else colsum)
    adam_k = mq[-1]
    last_bwd = mq[-2]
    bounds = [
        ("fwd: message passes", t0, gates_f[-1]["e"] if gates_f else gather_f["s"]),
        ("fwd: node-level readout", gates_f[-1]["e"] if gates_f else t0, gather_f["s"]),
        ("fwd: gather + tier 2", gather_f["s"], kl["s"]),
        ("loss", kl["s"], mean["e"]),
        ("bwd: tier 2", mean["e"], gather_b["s"]),
        ("bwd: gather / slot glue", gather_b["s"], colsum["e"]),
        ("bwd: node-level readout", colsum["e"], pack_b["s"]),
        ("bwd: message passes", pack_b["s"], last_bwd["e"]),
        ("tail (side queue only)", last_bwd["e"], adam_k["s"]),
        ("adam", adam_k["s"], adam_k["e"]),
    ]
    print("%-28s %9s %9s %9s" % ("phase", "span us", "main us", "side us"))
    tot = [0.0, 0.0, 0.0]
    for name, a, b in bounds:
        km = sum(min(r["e"], b) - max(r["s"], a) for r in mq if r["s"] < b and r["e"] > a)
        ks = sum(min(r["e"], b) - max(r["s"], a) for r in side if r["s"] < b and r["e"] > a)
        print("%-28s %9.1f %9.1f %9.1f" % (name, b - a, km, ks))
        tot = [tot[0] + b - a, tot[1] + km, tot[2] + ks]
    print("%-28s %9.1f %9.1f %9.1f" % ("step (from compact_fill)", *tot))
    if "--list" in sys.argv:
        for r in st:
            if r["s"] < t0:
                continue
            print("%9.1f %8.1f  q%-2s %-44s grid %s" % (r["s"] - t0, r["e"] - r["s"], r["Queue_Id"],
                                                      r["Kernel_Name"][:44], r["Grid_Size_X"]))


if __name__ == "__main__":
    main()
