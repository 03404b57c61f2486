// fp32 GEMM operands as TWO scaled fp16 values each ("fp16x2"), shared by the bf16-pipe GEMM kernels.
//
// x s = h1 + h2 + O(2^-22 |x s|),  h1 = fp16(x s) (round to nearest even), h2 = fp16(x s - h1) (the residual is exact in
// fp32), s = a power of two chosen PER TENSOR from its largest magnitude so that max |x| s lies in [2^13, 2^14): h1 never
// overflows fp16 (65 504) and an element keeps the full 22 bits as long as it is within 2^16 of the tensor's largest
// (below that its ABSOLUTE error is 2^-25 / s, i.e. 2^-38 of the largest element: nothing a dot product can see).
//   a b = (a1 b1 + a1 b2 + a2 b1) / (sa sb) + O(2^-22 |a b|)
// THREE f16 MFMA products per fp32 product, exact in the MFMA's fp32 accumulate, instead of the six bf16 products of
// the three-way bf16 split (whose planes need no scale because bf16 has fp32's exponent range, at 8 bits a plane):
// half the matrix-pipe work, two thirds of the LDS traffic, 6 instead of 11 VALU instructions per pair split
// (v_mul x 2, v_cvt_pk_f16_f32, v_fma_mix_f32 x 2 — the residual straight from the packed halves —, v_cvt_pk_f16_f32).
// Measured against the fp64 product (tests/test_kernels_gpu.py): the same ~3e-7 of max |ref| as the bf16 split and as
// the fp32 MFMA chain itself.  On a chip that clocks to its power budget under these kernels (1.5 GHz under the bf16x3
// kernels) halving the MFMA count is what buys throughput.
//
// The scale needs the tensor's largest magnitude BEFORE the GEMM that reads it: every kernel that produces a GEMM operand
// of such a launch (the GEMM epilogues of this library, gi_absmax for the weights) leaves max |value| in a caller-owned
// float (`gi_gemm_params.c_amax`, atomic max on the bit pattern: values are >= 0), and the consumer reads it through
// `a_amax` / `b_amax`.
#pragma once
#include "gi_common.h"

typedef _Float16 gx_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 gx_f16x8 __attribute__((ext_vector_type(8)));
typedef float gx_f32x2 __attribute__((ext_vector_type(2)));

// s = 2^(13 - floor(log2 amax)) as (scale, 1 / scale); amax == 0, denormal or NaN -> 1
__device__ __forceinline__ void gx_scale(float amax, float& s, float& inv) {
    const unsigned bits = __builtin_bit_cast(unsigned, amax);
    const int e = (int)((bits >> 23) & 0xffu);
    int se = 267 - e;                                  // biased exponent of 2^(140 - e)
    se = (e == 0 || !(amax == amax)) ? 127 : min(max(se, 2), 252);
    s = __builtin_bit_cast(float, (unsigned)se << 23);
    inv = __builtin_bit_cast(float, (unsigned)(254 - se) << 23);
}
// two fp32 values (already multiplied by nothing: the scale is applied here) -> their two fp16 planes, packed pairwise
__device__ __forceinline__ void gx_split2(float x0, float x1, float s, unsigned& p0, unsigned& p1) {
#ifdef GI_X2_SPLIT4
    // FOUR instructions per pair on v_fma_mixlo / mixhi_f16 (round 6; tools/split_lab.hip: both planes bit-identical to the
    // six-instruction form below over fp16 normals, subnormals and zeros): the product x s rounded straight to fp16 (one
    // rounding of the exact product: s is a power of two), the residual x s - h1 formed by ONE fma from the fp16 half and
    // rounded straight to fp16 (x s - h1 is exact in fp32, so fp16(fma) == fp16(fl32(x s) - h1)).
    // (plain C: with -fno-slp-vectorize hipcc selects v_fma_mixlo / mixhi_f16 for these — real VALU instructions the
    // scheduler and gi_gemm_b3p.hip's sched_group_barrier pattern can place; the SLP vectoriser turns them back into
    // v_pk_fma_f32 + v_cvt_pk_f16_f32)
    gx_f16x2 h, r;
    h.x = (_Float16)__builtin_fmaf(x0, s, 0.f);
    h.y = (_Float16)__builtin_fmaf(x1, s, 0.f);
    r.x = (_Float16)__builtin_fmaf(x0, s, -(float)h.x);
    r.y = (_Float16)__builtin_fmaf(x1, s, -(float)h.y);
    p0 = __builtin_bit_cast(unsigned, h);
    p1 = __builtin_bit_cast(unsigned, r);
#else
    const float y0 = x0 * s, y1 = x1 * s;
    gx_f32x2 v = {y0, y1};
    const gx_f16x2 h = __builtin_convertvector(v, gx_f16x2);            // v_cvt_pk_f16_f32 (RNE)
    p0 = __builtin_bit_cast(unsigned, h);
    gx_f32x2 r = {y0 - (float)h.x, y1 - (float)h.y};                    // v_fma_mix_f32 x 2 (exact)
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, gx_f16x2));
#endif
}
// A tensor's largest magnitude lives in an "amax cell": GI_AMAX_WORDS floats = 64 slots one 128-byte line apart.
// Thousands of waves finish their epilogues within microseconds of one another; into ONE word their atomics queue on
// the same address in L2 (measured: +65 us on a 42-us launch of 11 000 waves, profiles/r04); spread over 64 lines by
// wave number they do not.  Readers take the maximum of the 64 slots (one load per lane).
#define GX_AMAX_SLOTS 64
#define GX_AMAX_STRIDE (GI_AMAX_WORDS / GX_AMAX_SLOTS)
// whole wave, uniform result
__device__ __forceinline__ float gx_amax_read(const float* cell) {
    float m = cell[(threadIdx.x & 63) * GX_AMAX_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
// largest |v| over the lanes of a wave -> one atomic max on the bit pattern (all values >= 0) of the wave's slot
__device__ __forceinline__ void gx_amax_publish(float m, float* cell) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float* slot = cell + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (GX_AMAX_SLOTS - 1)) * GX_AMAX_STRIDE;
    // look first: a stale value only costs a redundant atomic (the maximum is monotonic)
    if ((threadIdx.x & 63) == 0 && m > *reinterpret_cast<volatile float*>(slot))
        atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, m));
}
