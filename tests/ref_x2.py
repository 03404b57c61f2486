"""numpy model of the fp16x2 operand split of csrc/gi_x2.h (test infrastructure): what `gx_scale` / `gx_split2` and the
three-product MFMA sum compute, so that the accuracy statements of that header and of DESIGN.md section 2 are checked on
CPU as arithmetic, independently of the kernels that use them."""
import numpy as np


def scale(amax: float):
    """gx_scale: s = 2^(13 - floor(log2 amax)) so that amax * s is in [2^13, 2^14); amax == 0 / denormal / NaN -> 1."""
    amax = np.float32(amax)
    e = (amax.view(np.uint32) >> 23) & 0xFF
    if e == 0 or not (amax == amax):
        return np.float32(1.0), np.float32(1.0)
    se = int(min(max(267 - int(e), 2), 252))
    s = np.uint32(se << 23).view(np.float32)
    inv = np.uint32((254 - se) << 23).view(np.float32)
    return s, inv


def split(x: np.ndarray, s: np.float32):
    """gx_split2: y = x * s (fp32), h1 = fp16(y) round-to-nearest-even, h2 = fp16(y - h1) (the residual is exact in fp32)."""
    y = (x.astype(np.float32) * s).astype(np.float32)
    h1 = y.astype(np.float16)
    h2 = (y - h1.astype(np.float32)).astype(np.float16)
    return h1, h2


def matmul(a: np.ndarray, b: np.ndarray):
    """C = A @ B^T as the fp16x2 kernels form it: per-tensor scales from the largest magnitudes, three products of fp16
    planes (each exact in fp32), accumulated here in float64 (the MFMA accumulates in fp32 over 16-deep blocks; the
    difference is the fp32 accumulation error every fp32 GEMM has), descaled by the two inverse scales."""
    sa, ia = scale(np.abs(a).max())
    sb, ib = scale(np.abs(b).max())
    a1, a2 = split(a, sa)
    b1, b2 = split(b, sb)
    f = lambda h: h.astype(np.float64)
    acc = f(a2) @ f(b1).T + f(a1) @ f(b2).T + f(a1) @ f(b1).T
    return acc * float(ia) * float(ib)
