"""How does the duration of ONE chain launch grow with its number of workgroups?  Forward chain of the headline stack
(128 -> 250^4 -> 128, one bond type so that every workgroup streams the same 1.15 MB image), row blocks = 8 ... 512, for
the three chain kernels (fp32: 32-row blocks forced; fp16x2: 64-row blocks; fp16x2 row-independent: 32-row blocks
forced).  Run under `rocprofv3 --kernel-trace` and feed the trace to this script's `report` mode:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o t -- python tools/chain_scaling.py run
    python tools/chain_scaling.py report /tmp/cs/*kernel_trace.csv
A kernel whose workgroups do not compete for anything takes the same time at 8 and at 256 workgroups (one per CU)."""
import csv
import os
import statistics
import sys

BLOCKS = tuple(int(b) for b in os.environ.get("CHAIN_SCALING_BLOCKS", "8,32,64,128,192,256,264,384,512,528,768,1024").split(","))
KINDS = tuple(os.environ.get("CHAIN_SCALING_KINDS", "fp32,x2,x2r").split(","))


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from graphinvent_amd import lib as L, ops
    lib = L.load()
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    sizes = (128, 250, 250, 250, 250, 128)
    Ws = [[(torch.randn(o, i, generator=g) / i ** 0.5).to(dev)] for i, o in zip(sizes, sizes[1:])]
    bs = [[torch.randn(o, generator=g).to(dev)] for o in sizes[1:]]
    for kind in KINDS:
        L.check(lib.gi_mlp_chain_config(32, 0, 2, None), "cfg")       # 32-row blocks (fp32, x2r), never the 64-row fp32 variant
        for nb in BLOCKS:
            U = nb * (64 if kind == "x2" else 32)
            X = torch.randn(U, 128, generator=g).to(dev)
            outs = [torch.empty(U, ops.r4(o), device=dev) for o in sizes[1:]]
            spec = dict(X=X, x_idx=None, grp_off=None, group_rows=[U], rows=U,
                        layers=[dict(W=Ws[l], bias=bs[l], out=outs[l], act=None, K=sizes[l], N=sizes[l + 1])
                                for l in range(5)])
            for _ in range(6):
                ops.mlp_chain([spec], backward=False, x2=kind != "fp32", rows32=kind == "x2r")
            torch.cuda.synchronize()
    lib.gi_mlp_chain_config(0, -1, 2, None)


def report(path):
    by = {}
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        if "gi_chain" not in name or "pack" in name:
            continue
        short = "x2r" if "x2r" in name else "x2" if "x2" in name else "fp32"
        wgs = int(row["Grid_Size_X"]) // int(row["Workgroup_Size_X"]) if "Grid_Size_X" in row else int(row["Grid_Size"]) // 512
        by.setdefault((short, wgs), []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    print("workgroups   " + "".join(f"{k:>10s}" for k in ("fp32", "x2", "x2r")) + "    (us per launch, median of the last 4 of 6)")
    for nb in sorted({k[1] for k in by}):
        print(f"{nb:10d}   " + "".join(f"{statistics.median(by[(k, nb)][2:]):10.1f}" if (k, nb) in by else f"{'-':>10s}"
                                        for k in ("fp32", "x2", "x2r")))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
