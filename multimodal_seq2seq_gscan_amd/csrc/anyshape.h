// Building blocks of the any-shape kernels (decoder_any.hip, the any-size command encoder in lstm_encoder.hip): products
// of a weight matrix in global memory (the reference's row-major [out, in] layout, L2-resident: every workgroup streams
// the same weights) with a vector in LDS, by a 1024-thread workgroup.
#pragma once
#include "step.h"

namespace gscan {

constexpr int kAnyThreads = 1024, kAnyWaves = kAnyThreads / 64;     // sixteen waves: the loads in flight hide the L2 latency

// sum over the sixteen lanes of a DPP row; every lane of the row gets it (all 64 lanes active)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xb1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_move<0x4e, 0xf>(v);        // quad_perm [2,3,0,1]
    v += dpp_move<0x124, 0xf>(v);       // row_ror:4
    v += dpp_move<0x128, 0xf>(v);       // row_ror:8
    return v;
}

// y[r] = W[r, 0:C] . x  for r < R, delivered through `store(r, value)` by one lane per row.  W row-major with row
// stride ldw (global); x in LDS.  Sixteen lanes share a row and a thread works on four rows at once (four independent
// 16-byte loads in flight per 64 columns, one read of x for the four); V4: 16-byte loads (ldw, C multiples of 4,
// aligned bases).
template <bool V4, typename Store>
__device__ __forceinline__ void matvec_rows(const float *__restrict__ W, int ldw, int R, int C, const float *x, Store store) {
    constexpr int U = 4, ngrp = kAnyThreads / 16;
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    for (int r0 = 0; r0 < R; r0 += ngrp * U) {                // uniform trip count: the DPP sums need every lane
        const float *wrow[U];
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wrow[u] = W + (int64_t)min(r0 + grp + u * ngrp, R - 1) * ldw;
            acc[u] = 0.f;
        }
        if (V4) {
#pragma unroll 2
            for (int c = 4 * l16; c < C; c += 64) {
                float4 w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const float4 *>(wrow[u] + c);
                const float4 v = *reinterpret_cast<const float4 *>(x + c);
#pragma unroll
                for (int u = 0; u < U; ++u)
                    acc[u] = fmaf(w[u].x, v.x, fmaf(w[u].y, v.y, fmaf(w[u].z, v.z, fmaf(w[u].w, v.w, acc[u]))));
            }
        } else {
#pragma unroll 2
            for (int c = l16; c < C; c += 16) {
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = wrow[u][c];
                const float v = x[c];
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u] = fmaf(w[u], v, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float sum = row16_sum(acc[u]);
            const int r = r0 + grp + u * ngrp;
            if (l16 == 0 && r < R) store(r, sum);
        }
    }
}

// y[c] = sum_{r < R} W[r, c0 + c] * x[r]  for c < C (the transposed product), through `store(c, value)`.  A lane per
// column (coalesced across lanes), the rows dealt round-robin to the kAnyThreads / CB thread groups that share a
// column block (CB = C rounded up to a wave, at most the workgroup), four loads in flight per thread; the groups'
// partial sums meet in `scratch` (kAnyThreads floats of LDS).  x in LDS (broadcast reads).  Every thread of the
// workgroup must call it; it ends with a barrier (what `store` wrote is visible to all on return).
template <typename Store>
__device__ __forceinline__ void matvec_cols(const float *__restrict__ W, int ldw, int c0, int R, int C, const float *x,
                                            float *scratch, Store store) {
    const int tid = threadIdx.x;
    const int CB = min((C + 63) & ~63, kAnyThreads), P = kAnyThreads / CB;
    const int cc = tid % CB, p = tid / CB;
    for (int cbase = 0; cbase < C; cbase += CB) {
        const int c = cbase + cc;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (p < P && c < C) {
            const float *wcol = W + c0 + c;
            int r = p;
#pragma unroll 2
            for (; r + 3 * P < R; r += 4 * P) {
                const float w0 = wcol[(int64_t)r * ldw], w1 = wcol[(int64_t)(r + P) * ldw],
                            w2 = wcol[(int64_t)(r + 2 * P) * ldw], w3 = wcol[(int64_t)(r + 3 * P) * ldw];
                a0 = fmaf(w0, x[r], a0);
                a1 = fmaf(w1, x[r + P], a1);
                a2 = fmaf(w2, x[r + 2 * P], a2);
                a3 = fmaf(w3, x[r + 3 * P], a3);
            }
            for (; r < R; r += P) a0 = fmaf(wcol[(int64_t)r * ldw], x[r], a0);
        }
        scratch[tid] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (tid < CB && c < C) {
            float sum = 0.f;
            for (int q = 0; q < P; ++q) sum += scratch[q * CB + tid];
            store(c, sum);
        }
        __syncthreads();
    }
}

}  // namespace gscan
