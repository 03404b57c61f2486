"""Why is dgrad ~20 % slower than the forward GEMM of the same shape?  B-major operand layout vs the
DSELU epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_gemm as G
from graphinvent_amd import lib as L, ops

def dgrad(M, n_out, n_in, dselu):
    dZ = torch.randn(M, ops.r4(n_out), device="cuda"); W = torch.randn(n_out, n_in, device="cuda")
    act = torch.randn(M, ops.r4(n_in), device="cuda"); out = torch.empty(M, ops.r4(n_in), device="cuda")
    return lambda: ops.gemm(dZ, W, out, M, n_in, n_out, dZ.shape[1], n_in, out.shape[1],
                            flags=L.EPI_DSELU if dselu else 0, act=act if dselu else None,
                            ldact=act.shape[1] if dselu else 0, b_major=True, tm=1, tn=1)

for (M, N, K) in [(7355, 500, 500), (7355, 250, 250), (8692, 250, 250), (1000, 500, 500)]:
    fl = 2.0 * M * N * K
    G.report(f"fwd   {M}x{N}x{K} bias+selu", G.timeit(G.fwd(M, N, K, 1, 1)), fl)
    G.report(f"fwd   {M}x{N}x{K} no epilogue", G.timeit(G.fwd(M, N, K, 1, 1, flags=0)), fl)
    G.report(f"dgrad {M}x{N}x{K} B-major, no epilogue", G.timeit(dgrad(M, K, N, False)), fl)
    G.report(f"dgrad {M}x{N}x{K} B-major + dselu", G.timeit(dgrad(M, K, N, True)), fl)
