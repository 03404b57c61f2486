import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from graphinvent_amd import lib as L
import bench_gemm as G
for (M, N, K) in [(7300, 250, 250), (7300, 500, 500), (1000, 500, 500), (13900, 250, 250), (29200, 512, 512)]:
    for extra in (0, 64):
        us = G.timeit(G.fwd(M, N, K, 1, 1, flags=L.EPI_BIAS | L.EPI_SELU | extra))
        G.report(f"fwd {M}x{N}x{K} fences={'on' if extra else 'off'}", us, 2.0 * M * N * K)
