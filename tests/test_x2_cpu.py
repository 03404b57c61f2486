"""CPU checks of the fp16x2 arithmetic (tests/ref_x2.py mirrors csrc/gi_x2.h): the scale keeps every tensor inside fp16's
range, two planes carry 22 bits of an element near the tensor's maximum, and the three-product sum is as close to the
fp64 product as the header and DESIGN.md section 2 say — including what is given up: dynamic range inside one tensor."""
import numpy as np
import pytest

from tests import ref_x2 as X


@pytest.mark.parametrize("amax", [1e-30, 3e-7, 0.24, 1.0, 7.9, 8.0, 65504.0, 3e4, 1e12, 3e37])
def test_scale_puts_the_largest_magnitude_into_2_13_2_14(amax):
    s, inv = X.scale(amax)
    y = np.float32(amax) * s
    assert 2.0 ** 13 <= float(y) < 2.0 ** 14 and float(s) * float(inv) == 1.0
    assert float(np.float16(y)) < 65504.0                      # never overflows fp16
    m, e = np.frexp(float(s))
    assert m == 0.5                                            # a power of two: scaling is exact


def test_scale_of_zero_nan_and_denormal_is_one():
    for v in (0.0, float("nan"), 1e-45):
        assert X.scale(v) == (np.float32(1.0), np.float32(1.0))


def test_two_planes_carry_22_bits_near_the_maximum_and_degrade_gracefully_below_it():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(200_000).astype(np.float32)
    s, inv = X.scale(np.abs(x).max())
    h1, h2 = X.split(x, s)
    back = (h1.astype(np.float64) + h2.astype(np.float64)) * float(inv)
    err = np.abs(back - x.astype(np.float64))
    big = np.abs(x) > np.abs(x).max() * 2.0 ** -10
    assert (err[big] / np.abs(x[big])).max() < 2.0 ** -21        # 22 bits on everything within 2^10 of the maximum
    assert err.max() < np.abs(x).max() * 2.0 ** -22               # overall: 2^-23 of the maximum ...
    small = np.abs(x) < np.abs(x).max() * 2.0 ** -16                # ... and for what lies 2^16 below it an absolute
    assert small.any() and err[small].max() < np.abs(x).max() * 2.0 ** -37      # error ~2^-38 of the maximum
    # an element 2^-24 below the tensor's maximum keeps ~14 bits (the statement of DESIGN.md section 2)
    tiny = np.array([1.0, 2.0 ** -24 * 1.2345], dtype=np.float32)
    s, inv = X.scale(1.0)
    h1, h2 = X.split(tiny, s)
    rel = abs((float(h1[1]) + float(h2[1])) * float(inv) - float(tiny[1])) / float(tiny[1])
    assert 1e-6 < rel < 2e-4


@pytest.mark.parametrize("scale_a,scale_b", [(1.0, 1.0), (3e4, 1e-3), (1e-6, 0.06)])
def test_three_products_match_the_fp64_product_like_an_fp32_gemm_does(scale_a, scale_b):
    rng = np.random.default_rng(1)
    a = (rng.standard_normal((96, 500)) * scale_a).astype(np.float32)
    b = (rng.standard_normal((64, 500)) * scale_b).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    got = X.matmul(a, b)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    fp32 = np.abs((a @ b.T).astype(np.float64) - ref).max() / np.abs(ref).max()     # numpy's own fp32 GEMM
    assert err < 3e-7 and err < 4 * fp32 + 1e-7, (err, fp32)
    # the dropped a2 b2 term is what bounds it: 2^-22 relative per product
    assert err > 1e-10


def test_the_ones_column_of_the_weight_gradient_layout_is_exact_for_any_scale():
    """gi_gemm_b3p stages the bias-gradient column as 1 / s_b, so that it is exactly 1.0 after scaling whatever the
    activations' amax (a plain 1.0 times s_b overflows fp16 when max |X| < 0.25)."""
    for amax in (1e-6, 0.2, 1.0, 5e3):
        s, inv = X.scale(amax)
        h1, h2 = X.split(np.array([inv], dtype=np.float32), s)
        assert float(h1[0]) == 1.0 and float(h2[0]) == 0.0
        plain = np.float32(1.0) * s
        assert (float(plain) > 65504.0) == (amax < 0.25)


def guard_counts(a: np.ndarray) -> int:
    """numpy model of gi_gemm_params.x2_guard (gi_gemm_bf3.hip / gi_gemm_b3p.hip): rows of A whose largest SCALED
    magnitude is non-zero and below 2^-11."""
    s, _ = X.scale(np.abs(a).max())
    rowmax = np.abs(a).max(axis=1).astype(np.float32) * s
    return int(((rowmax > 0) & (rowmax < np.float32(2.0 ** -11))).sum())


def test_guard_threshold_marks_the_rows_that_lose_their_bits():
    """The threshold of the dynamic-range guard in the terms of this arithmetic model: a row whose largest element scales
    below 2^-11 (= lies more than 2^24 below the tensor's maximum, which scales into [2^13, 2^14)) is represented with a
    per-row product error above the path's 1e-4 bar (here 6e-4 at 2^-25.5 below the maximum); a row just above the
    threshold (2^-20.5 below) stays at 1e-5; exactly-zero rows are exact and not counted: the guard counts the rows
    that can miss the bar and no others."""
    rng = np.random.default_rng(7)
    a = rng.standard_normal((64, 256)).astype(np.float32)
    b = rng.standard_normal((32, 256)).astype(np.float32)
    a[0, 0] = 8.0                                                   # the tensor's maximum: scale 2^10
    a[5] *= np.float32(2.0 ** -22)                                 # largest element ~2^-20.5 below the maximum: not counted
    a[6] *= np.float32(2.0 ** -27)                                 # ~2^-25.5 below: counted
    a[7] = 0.0                                                     # exact
    assert guard_counts(a) == 1
    s, inv = X.scale(np.abs(a).max())
    h1, h2 = X.split(a, s)
    back = (h1.astype(np.float64) + h2.astype(np.float64)) * float(inv)
    rel_row = np.abs(back - a).max(axis=1) / np.maximum(np.abs(a).astype(np.float64).max(axis=1), 1e-300)
    assert rel_row[7] == 0.0
    assert rel_row[5] < 2.0 ** -13 < rel_row[6]                       # what is left of the row's elements: ~14 bits / ~9 bits
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    got = X.matmul(a, b)
    S = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T
    err_row = np.abs(got - ref).max(axis=1) / S.max(axis=1).clip(1e-300)
    assert np.delete(err_row, [5, 6, 7]).max() < 1e-6                # rows near the maximum
    assert err_row[5] < 2e-5 and err_row[6] > 1e-4                    # the uncounted row meets the 1e-4 bar, the counted one may not
    # ... and per TENSOR nothing of it shows (the bar of the parity suite)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 3e-7
