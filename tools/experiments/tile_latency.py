"""Per-k-tile time of the GEMM kernel at low occupancy: slope of time vs K for a lone block and for
the tier-2 shape (128 blocks on 256 CUs).  MFMA-bound floor for one wave per SIMD: 16 MFMA x 64
cycles = 1024 cycles = 0.43 us per 32-wide k-tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_gemm as G

for (M, N) in [(64, 64), (1000, 500), (64 * 256, 64), (64 * 512, 64), (64 * 1024, 64)]:
    prev = None
    for K in (128, 256, 512, 1024, 2048):
        us = G.timeit(G.fwd(M, N, K, 1, 1), reps=50)
        slope = "" if prev is None else f"  d/dtile {(us - prev[1]) / ((K - prev[0]) / 32):6.3f} us"
        print(f"M={M:6d} N={N:4d} K={K:5d}: {us:8.1f} us{slope}")
        prev = (K, us)
