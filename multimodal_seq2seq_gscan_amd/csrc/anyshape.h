// Building blocks of the any-shape kernels (decoder_any.hip, the any-size command encoder in lstm_encoder.hip): products
// of a weight matrix in global memory (the reference's row-major [out, in] layout, L2-resident: every workgroup streams
// the same weights) with a vector in LDS, by a 1024-thread workgroup.
#pragma once
#include "step.h"

namespace gscan {

constexpr int kAnyThreads = 1024, kAnyWaves = kAnyThreads / 64;     // sixteen waves: the loads in flight hide the L2 latency

// threadIdx.x behind a compiler barrier.  The products below are inlined at a dozen call sites inside the kernels' time loops, and
// every per-thread address they derive from the thread index is loop-invariant: hoisted out of the time loop those were dozens of
// 64-bit pointers held in VGPRs across it, and the loops' loads were issued TWO at a time with a full wait in between (the ISA
// of round 4's kernels) because no registers were left to keep more in flight.  From an opaque index they are recomputed per call.
__device__ __forceinline__ int any_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// acc += w . v as two packed FMAs on register pairs that come out of ONE 16-byte load each ((x, y) and (z, w)).  Written as four
// scalar FMAs per row, the compiler packed ACROSS rows instead ({w[u].x, w[u+1].x} ...), shuffled every loaded float4 into
// that layout with v_mov right behind the loads, and so waited for the loads two at a time (the ISA of round 4's kernels):
// a thread never had more than 32 bytes in flight, whatever the source said.
using any_f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void any_fma4(any_f32x2 &a01, any_f32x2 &a23, const float4 &w, const float4 &v) {
    a01 += any_f32x2{w.x, w.y} * any_f32x2{v.x, v.y};
    a23 += any_f32x2{w.z, w.w} * any_f32x2{v.z, v.w};
}

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
    const int tid = any_tid(), l16 = tid & 15, grp = tid >> 4;
    for (int r0 = 0; r0 < R; r0 += ngrp * U) {                // uniform trip count: the DPP sums need every lane
        const float *wrow[U];
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wrow[u] = W + (int64_t)min(r0 + grp + u * ngrp, R - 1) * ldw;
            acc[u] = 0.f;
        }
        if (V4) {
            any_f32x2 a01[U], a23[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { a01[u] = any_f32x2{0.f, 0.f}; a23[u] = any_f32x2{0.f, 0.f}; }
#pragma unroll 2
            for (int c = 4 * l16; c < C; c += 64) {
                float4 w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const float4 *>(wrow[u] + c);
                const float4 v = *reinterpret_cast<const float4 *>(x + c);
#pragma unroll
                for (int u = 0; u < U; ++u) any_fma4(a01[u], a23[u], w[u], v);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = (a01[u].x + a01[u].y) + (a23[u].x + a23[u].y);
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

// matvec_rows fed from a TRIP-MAJOR image of the matrix (any_stream_image_*) and software-pipelined over the trips (a trip = this
// thread's four rows x one 64-column piece = four 16-byte loads; trip t + 1 is requested before trip t is consumed, across row
// blocks: two buffers, the registers of the plain form's loads).  Element (trip t, row slot u, thread) of the image is one
// float4 at ((t * 4 + u) * kAnyThreads + thread): every wave load is 1 KB of consecutive addresses, pieces outside the matrix
// are stored zeros (no clamping).  Used for the one long product of a forward step, [W_hh ; W_q2k_h ; W_qt] . h (any_wcat).
__host__ __device__ inline int any_stream_trips(int R, int C) { return ((R + 255) / 256) * ((C + 63) >> 6); }
__host__ __device__ inline size_t any_stream_image_floats(int R, int C) { return (size_t)any_stream_trips(R, C) * 4 * kAnyThreads * 4; }
template <typename Store>
__device__ __forceinline__ void matvec_rows_image(const float *__restrict__ img, int R, int C, const float *x, Store store) {
    constexpr int U = 4, ngrp = kAnyThreads / 16, RB = ngrp * U;
    const int tid = any_tid(), l16 = tid & 15, grp = tid >> 4;
    const int ncc = (C + 63) >> 6, ntrips = ((R + RB - 1) / RB) * ncc;
    const float4 *img4 = reinterpret_cast<const float4 *>(img) + tid;
    auto issue = [&](int t, float4 (&w)[U]) {
        if (t >= ntrips) return;
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = img4[(size_t)(t * U + u) * kAnyThreads];
    };
    any_f32x2 a01[U], a23[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { a01[u] = any_f32x2{0.f, 0.f}; a23[u] = any_f32x2{0.f, 0.f}; }
    auto consume = [&](int t, const float4 (&w)[U]) {
        if (t >= ntrips) return;
        const int rb = t / ncc, cc = t - rb * ncc;
        const int col = cc * 64 + 4 * l16;
        const float4 v = col < C ? *reinterpret_cast<const float4 *>(x + col) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u) any_fma4(a01[u], a23[u], w[u], v);
        if (cc == ncc - 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float sum = row16_sum((a01[u].x + a01[u].y) + (a23[u].x + a23[u].y));
                const int r = rb * RB + grp + u * ngrp;
                if (l16 == 0 && r < R) store(r, sum);
                a01[u] = any_f32x2{0.f, 0.f}; a23[u] = any_f32x2{0.f, 0.f};
            }
        }
    };
    float4 w0[U], w1[U];
    issue(0, w0);
    for (int t = 0; t < ntrips; t += 2) {
        issue(t + 1, w1);
        consume(t, w0);
        issue(t + 2, w0);
        consume(t + 1, w1);
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
    const int tid = any_tid();
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
        lds_barrier();
        if (tid < CB && c < C) {
            float sum = 0.f;
            for (int q = 0; q < P; ++q) sum += scratch[q * CB + tid];
            store(c, sum);
        }
        lds_barrier();
    }
}

// The transposed products of up to three matrices ADDED UP, y[c] = sum_s sum_{r < R_s} W_s[r, c] * x_s[r] for c < C, as one pass
// over the concatenated rows with 16-byte loads (matvec_cols4's form): dh_{t-1} = W_hh^T delta + W_qt^T dqt + W_q2k[:, :H]^T dzq
// of the reverse streaming kernel was three calls — three exchanges, six barriers, three ramps — for one vector (round 5).
// C, every ld multiples of 4; an unused set: R = 0.
struct ColsRows { const float *W; int ld, R; const float *x; };
template <typename Store>
__device__ __forceinline__ void matvec_cols4_sets(const ColsRows s0, const ColsRows s1, const ColsRows s2, int C, float *scratch,
                                                  Store store) {
    const int tid = any_tid();
    const int Q = C >> 2, QB = min((Q + 15) & ~15, kAnyThreads), P = kAnyThreads / QB;
    const int qq = tid % QB, p = tid / QB;
    const int R01 = s0.R + s1.R, Rt = R01 + s2.R;
    for (int qbase = 0; qbase < Q; qbase += QB) {
        const int q = qbase + qq;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if (p < P && q < Q) {
            auto at = [&](int r, float &xv) -> const float * {         // virtual row r -> its matrix row and vector element
                const int set = r < s0.R ? 0 : (r < R01 ? 1 : 2);
                const int rl = set == 0 ? r : (set == 1 ? r - s0.R : r - R01);
                const float *W = set == 0 ? s0.W : (set == 1 ? s1.W : s2.W);
                const int ld = set == 0 ? s0.ld : (set == 1 ? s1.ld : s2.ld);
                xv = (set == 0 ? s0.x : (set == 1 ? s1.x : s2.x))[rl];
                return W + (int64_t)rl * ld + 4 * q;
            };
            int r = p;
#pragma unroll 1
            for (; r + 3 * P < Rt; r += 4 * P) {
                float x0, x1, x2, x3;
                const float *p0 = at(r, x0), *p1 = at(r + P, x1), *p2 = at(r + 2 * P, x2), *p3 = at(r + 3 * P, x3);
                const float4 w0 = *reinterpret_cast<const float4 *>(p0), w1 = *reinterpret_cast<const float4 *>(p1),
                             w2 = *reinterpret_cast<const float4 *>(p2), w3 = *reinterpret_cast<const float4 *>(p3);
                acc.x = fmaf(w0.x, x0, fmaf(w1.x, x1, fmaf(w2.x, x2, fmaf(w3.x, x3, acc.x))));
                acc.y = fmaf(w0.y, x0, fmaf(w1.y, x1, fmaf(w2.y, x2, fmaf(w3.y, x3, acc.y))));
                acc.z = fmaf(w0.z, x0, fmaf(w1.z, x1, fmaf(w2.z, x2, fmaf(w3.z, x3, acc.z))));
                acc.w = fmaf(w0.w, x0, fmaf(w1.w, x1, fmaf(w2.w, x2, fmaf(w3.w, x3, acc.w))));
            }
            for (; r < Rt; r += P) {
                float x0;
                const float4 w0 = *reinterpret_cast<const float4 *>(at(r, x0));
                acc.x = fmaf(w0.x, x0, acc.x); acc.y = fmaf(w0.y, x0, acc.y); acc.z = fmaf(w0.z, x0, acc.z); acc.w = fmaf(w0.w, x0, acc.w);
            }
        }
#pragma unroll
        for (int comp = 0; comp < 4; ++comp) {
            scratch[tid] = comp == 0 ? acc.x : comp == 1 ? acc.y : comp == 2 ? acc.z : acc.w;
            lds_barrier();
            if (tid < QB && q < Q) {
                float sum = 0.f;
                for (int g = 0; g < P; ++g) sum += scratch[g * QB + tid];
                store(4 * q + comp, sum);
            }
            lds_barrier();
        }
    }
}

// y[m] = A0[m, :] . x0 + A1[m, :] . x1 + A2[m, :] . x2 for m < n: FEW rows (an attention's memories) against long vectors — a
// wave per row, a lane a 16-byte piece of every array's row per 256 columns, one wave sum per row (round 5).  matvec_rows
// gives sixteen lanes to a row and four rows to a thread: with 10 or 36 rows nine tenths of its loads were clamped
// duplicates, and the three arrays were three calls with a barrier each — 12-15 k cycles per attention for 23 k multiply-adds.
// C0..C2 multiples of 4 (an unused array: C = 0), rows and vectors 16-byte aligned.
struct FewRows { const float *A; int ld, C; const float *x; };
template <typename Epi>
__device__ __forceinline__ void rows_few(const FewRows a0, const FewRows a1, const FewRows a2, int n, Epi epi) {
    const int tid_ = any_tid(), lane = tid_ & 63, wave = tid_ >> 6;
    for (int m0 = 0; m0 < n; m0 += kAnyWaves) {
        const int m = m0 + wave, mc = min(m, n - 1);
        float p = 0.f;
        auto part = [&](const FewRows &a) {
            const float *row = a.A + (int64_t)mc * a.ld;
            for (int c = 4 * lane; c < a.C; c += 256) {
                const float4 w = *reinterpret_cast<const float4 *>(row + c), v = *reinterpret_cast<const float4 *>(a.x + c);
                p = fmaf(w.x, v.x, fmaf(w.y, v.y, fmaf(w.z, v.z, fmaf(w.w, v.w, p))));
            }
        };
        part(a0); part(a1); part(a2);
        p = wave_sum(p);
        if (lane == 0 && m < n) epi(m, p);
    }
}

// The transposed product with 16-byte loads (round 5): C, ldw and c0 multiples of 4, W 16-byte aligned.  A thread owns a
// column QUAD and every P-th row (one float4 per row, four in flight): a lane per column with 4-byte loads had every thread
// walk R / P rows four at a time — W_hh^T (4H x H) was sixteen dependent L2 round trips of 16 KB per workgroup; this form
// makes it four of 64 KB.  The groups' partial sums meet in `scratch` (kAnyThreads floats), one component at a time.
template <typename Store>
__device__ __forceinline__ void matvec_cols4(const float *__restrict__ W, int ldw, int c0, int R, int C, const float *x,
                                             float *scratch, Store store) {
    const int tid = any_tid();
    const int Q = C >> 2, QB = min((Q + 15) & ~15, kAnyThreads), P = kAnyThreads / QB;
    const int qq = tid % QB, p = tid / QB;
    for (int qbase = 0; qbase < Q; qbase += QB) {
        const int q = qbase + qq;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if (p < P && q < Q) {
            const float *wq = W + c0 + 4 * q;
            int r = p;
#pragma unroll 1
            for (; r + 3 * P < R; r += 4 * P) {
                const float4 w0 = *reinterpret_cast<const float4 *>(wq + (int64_t)r * ldw),
                             w1 = *reinterpret_cast<const float4 *>(wq + (int64_t)(r + P) * ldw),
                             w2 = *reinterpret_cast<const float4 *>(wq + (int64_t)(r + 2 * P) * ldw),
                             w3 = *reinterpret_cast<const float4 *>(wq + (int64_t)(r + 3 * P) * ldw);
                const float x0 = x[r], x1 = x[r + P], x2 = x[r + 2 * P], x3 = x[r + 3 * P];
                acc.x = fmaf(w0.x, x0, fmaf(w1.x, x1, fmaf(w2.x, x2, fmaf(w3.x, x3, acc.x))));
                acc.y = fmaf(w0.y, x0, fmaf(w1.y, x1, fmaf(w2.y, x2, fmaf(w3.y, x3, acc.y))));
                acc.z = fmaf(w0.z, x0, fmaf(w1.z, x1, fmaf(w2.z, x2, fmaf(w3.z, x3, acc.z))));
                acc.w = fmaf(w0.w, x0, fmaf(w1.w, x1, fmaf(w2.w, x2, fmaf(w3.w, x3, acc.w))));
            }
            for (; r < R; r += P) {
                const float4 w0 = *reinterpret_cast<const float4 *>(wq + (int64_t)r * ldw);
                const float x0 = x[r];
                acc.x = fmaf(w0.x, x0, acc.x); acc.y = fmaf(w0.y, x0, acc.y); acc.z = fmaf(w0.z, x0, acc.z); acc.w = fmaf(w0.w, x0, acc.w);
            }
        }
#pragma unroll
        for (int comp = 0; comp < 4; ++comp) {
            scratch[tid] = comp == 0 ? acc.x : comp == 1 ? acc.y : comp == 2 ? acc.z : acc.w;
            lds_barrier();
            if (tid < QB && q < Q) {
                float sum = 0.f;
                for (int g = 0; g < P; ++g) sum += scratch[g * QB + tid];
                store(4 * q + comp, sum);
            }
            lds_barrier();
        }
    }
}

// The transposed product over up to NA arrays that share the rows and the vector x (an attention's memories: projected
// keys [n, H], gate images [n, 4H], U2 [n, H] against the same attention distribution): y_j[c] = sum_{r < R} A_j[r, c] * x[r],
// through `store(j, c, value)`, as ONE pass over the concatenated columns — one exchange through `scratch` and two barriers
// for the lot instead of per array (round 5: a call costs a load round trip and two barriers whatever it moves).
struct ColsArray { const float *A; int ld, C; };
template <typename Store>
__device__ __forceinline__ void matvec_cols_arrays(const ColsArray a0, const ColsArray a1, const ColsArray a2, int R, const float *x,
                                                   float *scratch, Store store) {           // an unused array: C = 0
    const int tid = any_tid();
    const int Ctot = a0.C + a1.C + a2.C;
    const int CB = min((Ctot + 63) & ~63, kAnyThreads), P = kAnyThreads / CB;
    const int cc = tid % CB, p = tid / CB;
    for (int cbase = 0; cbase < Ctot; cbase += CB) {
        const int c = cbase + cc;
        // this thread's array and its column in it (plain scalars: an array of descriptors indexed per lane would live in scratch)
        const int j = c < a0.C ? 0 : (c < a0.C + a1.C ? 1 : 2);
        const int cl = j == 0 ? c : (j == 1 ? c - a0.C : c - a0.C - a1.C);
        const float *wcol = (j == 0 ? a0.A : (j == 1 ? a1.A : a2.A)) + cl;
        const int ldw = j == 0 ? a0.ld : (j == 1 ? a1.ld : a2.ld);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (p < P && c < Ctot) {
            int r = p;
#pragma unroll 2
            for (; r + 3 * P < R; r += 4 * P) {
                const float w0 = wcol[(int64_t)r * ldw], w1 = wcol[(int64_t)(r + P) * ldw],
                            w2 = wcol[(int64_t)(r + 2 * P) * ldw], w3 = wcol[(int64_t)(r + 3 * P) * ldw];
                s0 = fmaf(w0, x[r], s0);
                s1 = fmaf(w1, x[r + P], s1);
                s2 = fmaf(w2, x[r + 2 * P], s2);
                s3 = fmaf(w3, x[r + 3 * P], s3);
            }
            for (; r < R; r += P) s0 = fmaf(wcol[(int64_t)r * ldw], x[r], s0);
        }
        scratch[tid] = (s0 + s1) + (s2 + s3);
        lds_barrier();
        if (tid < CB && c < Ctot) {
            float sum = 0.f;
            for (int q = 0; q < P; ++q) sum += scratch[q * CB + tid];
            store(j, cl, sum);
        }
        lds_barrier();
    }
}

}  // namespace gscan
