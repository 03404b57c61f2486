"""GPU microbenchmark of the gi_gemm family: back-to-back launches of one shape, torch events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphinvent_amd import lib as L, ops

def timeit(fn, reps=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us

def fwd(M, N, K, tm, tn, flags=L.EPI_BIAS | L.EPI_SELU):
    X = torch.randn(M, ops.r4(K), device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda"); Y = torch.empty(M, ops.r4(N), device="cuda")
    return lambda: ops.gemm(X, W, Y, M, N, K, X.shape[1], K, Y.shape[1], flags=flags, bias=b, tm=tm, tn=tn)

def dgrad(M, n_out, n_in, tm, tn):
    dZ = torch.randn(M, ops.r4(n_out), device="cuda"); W = torch.randn(n_out, n_in, device="cuda")
    act = torch.randn(M, ops.r4(n_in), device="cuda")
    return lambda: ops.gemm(dZ, W, act, M, n_in, n_out, dZ.shape[1], n_in, act.shape[1], flags=L.EPI_DSELU, act=act, ldact=act.shape[1], b_major=True, tm=tm, tn=tn)

def wgrad(R, n_out, n_in, nsplit, tn):
    dZ = torch.randn(R, ops.r4(n_out), device="cuda"); X = torch.randn(R, ops.r4(n_in), device="cuda")
    ld = ops.r4(n_in + 1); stride = ops.r4(n_out * ld)
    slabs = torch.empty(nsplit * stride, device="cuda")
    return lambda: ops.gemm(dZ, X, slabs, n_out, n_in + 1, R, dZ.shape[1], X.shape[1], ld, flags=L.GEMM_SPLITK, a_major=True, b_major=True, tm=1, tn=tn, nsplit=nsplit, c_split_stride=stride, ones_col=n_in)

def report(name, us, flops):
    print(f"{name:55s} {us:9.1f} us  {flops / us / 1e6:7.1f} TF")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":      # single shape, for rocprofv3 --pmc passes
        M, N, K, tm, tn = (int(x) for x in sys.argv[2:7])
        f = fwd(M, N, K, tm, tn)
        for _ in range(10): f()
        torch.cuda.synchronize()
        sys.exit(0)
    for (M, N, K) in [(7300, 500, 500), (7300, 250, 250), (7300, 128, 250), (7300, 384, 128), (1000, 500, 685), (1000, 500, 500), (13000, 250, 250), (7300, 512, 4096), (29200, 512, 512)]:
        for tm, tn in [(1, 1), (1, 2), (2, 2)]:
            report(f"fwd M={M} N={N} K={K} tile=({tm},{tn})", timeit(fwd(M, N, K, tm, tn)), 2.0 * M * N * K)
        report(f"fwd M={M} N={N} K={K} tile=(1,2) no-epilogue", timeit(fwd(M, N, K, 1, 2, flags=0)), 2.0 * M * N * K)
    for (M, no, ni) in [(7300, 500, 500), (7300, 250, 250), (1000, 500, 500)]:
        for tm, tn in [(1, 2), (2, 2)]:
            report(f"dgrad M={M} n_out={no} n_in={ni} tile=({tm},{tn})", timeit(dgrad(M, no, ni, tm, tn)), 2.0 * M * no * ni)
    for (R, no, ni, ns) in [(7300, 500, 500, 8), (7300, 500, 500, 16), (7300, 250, 250, 32), (1000, 500, 500, 8), (12600, 250, 250, 11)]:
        report(f"wgrad R={R} n_out={no} n_in={ni} nsplit={ns}", timeit(wgrad(R, no, ni, ns, 2)), 2.0 * R * no * (ni + 1))
