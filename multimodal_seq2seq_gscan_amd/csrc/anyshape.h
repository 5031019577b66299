// Building blocks of the any-shape kernels (decoder_any.hip, the any-size command encoder in lstm_encoder.hip): products
// of a weight matrix in global memory (the reference's row-major [out, in] layout, L2-resident: every workgroup streams
// the same weights) with a vector in LDS, by a 256-thread workgroup.
#pragma once
#include "step.h"

namespace gscan {

constexpr int kAnyThreads = 256, kAnyWaves = kAnyThreads / 64;

// sum over the sixteen lanes of a DPP row; every lane of the row gets it (all 64 lanes active)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xb1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_move<0x4e, 0xf>(v);        // quad_perm [2,3,0,1]
    v += dpp_move<0x124, 0xf>(v);       // row_ror:4
    v += dpp_move<0x128, 0xf>(v);       // row_ror:8
    return v;
}

// y[r] = W[r, 0:C] . x  for r < R, delivered through `store(r, value)` by one lane per row.  W row-major with row
// stride ldw (global); x in LDS.  Sixteen lanes share a row; V4: 16-byte loads (ldw, C multiples of 4, aligned bases).
template <bool V4, typename Store>
__device__ __forceinline__ void matvec_rows(const float *__restrict__ W, int ldw, int R, int C, const float *x, Store store) {
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4, ngrp = kAnyThreads / 16;
    for (int r0 = 0; r0 < R; r0 += ngrp) {                    // uniform trip count: the DPP sums need every lane
        const int r = r0 + grp, rc = min(r, R - 1);
        const float *wrow = W + (int64_t)rc * ldw;
        float acc = 0.f;
        if (V4) {
            for (int c = 4 * l16; c < C; c += 64) {
                const float4 w = *reinterpret_cast<const float4 *>(wrow + c);
                const float4 v = *reinterpret_cast<const float4 *>(x + c);
                acc = fmaf(w.x, v.x, fmaf(w.y, v.y, fmaf(w.z, v.z, fmaf(w.w, v.w, acc))));
            }
        } else {
            for (int c = l16; c < C; c += 16) acc = fmaf(wrow[c], x[c], acc);
        }
        acc = row16_sum(acc);
        if (l16 == 0 && r < R) store(r, acc);
    }
}

// y[c] = sum_{r < R} W[r, c0 + c] * x[r]  for c < C (the transposed product), through `store(c, value)`: a lane per
// column, coalesced across lanes; x in LDS (broadcast reads).
template <typename Store>
__device__ __forceinline__ void matvec_cols(const float *__restrict__ W, int ldw, int c0, int R, int C, const float *x, Store store) {
    for (int c = threadIdx.x; c < C; c += kAnyThreads) {
        const float *wcol = W + c0 + c;
        float a0 = 0.f, a1 = 0.f;
        int r = 0;
        for (; r + 1 < R; r += 2) {
            a0 = fmaf(wcol[(int64_t)r * ldw], x[r], a0);
            a1 = fmaf(wcol[(int64_t)(r + 1) * ldw], x[r + 1], a1);
        }
        if (r < R) a0 = fmaf(wcol[(int64_t)r * ldw], x[r], a0);
        store(c, a0 + a1);
    }
}

}  // namespace gscan
