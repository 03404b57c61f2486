// MFMA-kernel helpers shared by the GEMM family (gi_gemm.hip) and the resident-activation MLP chain
// (gi_chain.hip): vector types and the branch-free guarded 16-byte staging load.
#pragma once
#include "gi_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4f_u __attribute__((aligned(4)));     // global rows are only guaranteed 4-byte aligned


// Branch-free guarded 16-byte staging load, split in two so the data is not touched until it is
// written to LDS (the loads of a tile issue back to back and stay in flight under the MFMAs):
//   gi_load4_raw : address clamped so the load always ends inside the row ([ncols-4, ncols), or
//                  [0,4) for rows narrower than 4 floats, whose storage is padded to 4)
//   gi_fix4      : v[j] = (valid && col + j < ncols) ? rowp[col + j] : 0 by shifting lanes back
// cmax = largest column a 16-byte load may start at: ncols-4 for exactly-sized rows, r4(ncols)-4
// when the row storage is padded to 4 floats (then no lane shift is ever needed).
__device__ __forceinline__ v4f gi_load4_raw(const float* rowp, int col, int cmax) {
    return *(const v4f_u*)(rowp + max(min(col, cmax), 0));
}
__device__ __forceinline__ v4f gi_fix4(v4f w, int col, int cmax, int ncols, bool valid) {
    const int s = col - max(min(col, cmax), 0);          // 0..3 = lanes to shift back, >= 4 = nothing valid
    const bool s1 = (s & 1) != 0, s2 = (s & 2) != 0;
    float x = w.x, y = w.y, z = w.z, t = w.w;            // two select stages (v_cndmask), no branches
    x = s1 ? y : x; y = s1 ? z : y; z = s1 ? t : z; t = s1 ? 0.f : t;
    x = s2 ? z : x; y = s2 ? t : y; z = s2 ? 0.f : z; t = s2 ? 0.f : t;
    const bool v = valid & (s < 4);
    v4f r;
    r.x = (v & (col < ncols)) ? x : 0.f;
    r.y = (v & (col + 1 < ncols)) ? y : 0.f;
    r.z = (v & (col + 2 < ncols)) ? z : 0.f;
    r.w = (v & (col + 3 < ncols)) ? t : 0.f;
    return r;
}

