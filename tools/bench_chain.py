"""Micro-benchmark of gi_mlp_chain on the headline workload's message-row shapes (U ~ 8.4 k rows in
three bond-type groups 85 / 14 / 1 %, stack H -> 250^4 -> M) against the same stack run layer by layer
through gi_gemm; prints microseconds per chain (HIP events, stream-ordered)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphinvent_amd import lib as L, ops  # noqa: E402

DEV = "cuda"


def main(U=8400, sizes=(128, 250, 250, 250, 250, 128), reps=30, backward=False):
    g = torch.Generator().manual_seed(0)
    cut = [0, int(U * 0.85), int(U * 0.99), U]
    off = torch.tensor(cut, dtype=torch.int32, device=DEV)
    rows_g = [cut[i + 1] - cut[i] for i in range(3)]
    R = 7300
    ldx = ops.r4(sizes[0] + 8)
    h = torch.randn(R, ldx, generator=g).to(DEV)
    idx = torch.randint(0, R, (U,), generator=g, dtype=torch.int32).to(DEV)
    if backward:
        sizes = tuple(reversed(sizes))
    Ws = [[(torch.randn(o, i, generator=g) / i ** 0.5).to(DEV) for _ in range(3)]
          for i, o in zip(sizes, sizes[1:])]
    if backward:      # W_l stored [K][N]
        Ws = [[w.t().contiguous() for w in ws] for ws in Ws]
    bs = [[torch.randn(o, generator=g).to(DEV) for _ in range(3)] for o in sizes[1:]]
    outs = [torch.empty(U, ops.r4(o), device=DEV) for o in sizes[1:]]
    acts = [torch.randn(U, ops.r4(o), device=DEV) for o in sizes[1:]]
    X = torch.randn(U, ops.r4(sizes[0]), device=DEV) if backward else h
    spec = dict(X=X, x_idx=None if backward else idx, grp_off=off, group_rows=rows_g, rows=U,
                layers=[dict(W=Ws[l], bias=bs[l], out=outs[l], act=acts[l] if backward else None,
                             K=sizes[l], N=sizes[l + 1]) for l in range(len(sizes) - 1)])

    def layered():
        x, ld, a_idx = X, X.stride(0), (None if backward else idx)
        for l in range(len(sizes) - 1):
            K, N = sizes[l], sizes[l + 1]
            if backward:
                ops.gemm(x, None, outs[l], U, N, K, ld, N, outs[l].stride(0), flags=L.EPI_DSELU,
                         act=acts[l], ldact=acts[l].stride(0), b_major=True, grp_off=off, ngroups=3,
                         max_group_rows=max(rows_g), Bg=Ws[l])
            else:
                ops.gemm(x, None, outs[l], U, N, K, ld, K, outs[l].stride(0),
                         flags=L.EPI_BIAS | L.EPI_SELU, a_idx=a_idx, grp_off=off, ngroups=3,
                         max_group_rows=max(rows_g), Bg=Ws[l], biasg=bs[l])
            x, ld, a_idx = outs[l], outs[l].stride(0), None

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    flops = 2.0 * U * sum(a * b for a, b in zip(sizes, sizes[1:]))
    x2 = bool(int(os.environ.get("BENCH_CHAIN_X2", "0")))          # the fp16x2 kernel (its pack launches are timed along:
    r32 = bool(int(os.environ.get("BENCH_CHAIN_ROWS32", "0")))     # ... its row-independent 32-row variant
    t_chain = timed(lambda: ops.mlp_chain([spec], backward=backward, x2=x2 or r32, rows32=r32))   # use rocprofv3 --stats for the kernel alone)
    t_layer = timed(layered)
    print(f"{'backward' if backward else 'forward '} U={U}: chain {t_chain:7.1f} us ({flops / t_chain / 1e6:5.1f} TF/s)"
          f"   layer-by-layer {t_layer:7.1f} us ({flops / t_layer / 1e6:5.1f} TF/s)", flush=True)


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else "both"
    U = int(sys.argv[2]) if len(sys.argv) > 2 else 8400
    if only in ("both", "fwd"):
        main(U=U)
    if only in ("both", "bwd"):
        main(U=U, backward=True)
