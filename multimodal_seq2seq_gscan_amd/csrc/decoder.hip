// Joint-attention LSTM decoder over all T steps (seq2seq/seq2seq_model.py:359-492), forward
// and backward, as two persistent kernels: ONE WORKGROUP PER BATCH ROW, time loop inside.
//
// Why this shape on MI355X.  The decoder is a chain of T dependent steps per row, and rows
// never interact, so the latency of one step — not FLOPs — bounds the training step
// (SURVEY.md §7 hard part 1).  With a row per workgroup, 256 rows fill the 256 CUs and a
// step needs no inter-workgroup traffic at all:
//   * every matrix that multiplies the recurrent state ( W_hh | W_query_text | W_q2k[:, :H]
//     and W_query_vis: 7*H*H floats = 280 KB at H=100) stays in VGPRs for all T steps, one
//     row (forward) or one column segment (backward) per thread — 2 MB of registers per CU
//     is the largest on-chip store, LDS (160 KB) could not hold them;
//   * the row's memories stay in LDS for all T steps: projected keys PK_text [L,H] and
//     PK_vis [G*G,H], plus U = PK . W_ih[:, ctx]^T ([L,4H] and [G*G,4H]).  Because an attention
//     context is a convex combination of projected keys, W_ih[:,ctx] . ctx = sum_m alpha_m U[m]:
//     the context part of the LSTM input GEMM becomes an M-term LDS reduction with no weights
//     at all (U comes from one dense MFMA GEMM before the loop).  The embedding part of the
//     LSTM input is known for all t (teacher forcing) and is also a GEMM before the loop;
//   * tanh(q + PK) score tiles are recomputed in backward instead of being saved
//     (46*H floats per step per row would make the path HBM-bound).
// The output head does not feed back, so it is hoisted out of the loop (host side).
//
// Thread roles (forward), H = hidden size, tid in [0, 7H):
//   [0,4H)  gate row j: W_hh[j,:] in registers, also owns column j of U_text / U_vis
//   [4H,5H) text query row k: W_query_text[k,:]; owns ctx_text[k], ctx_vis[k]
//   [5H,6H) conditional: W_q2k[k, 0:H] (and U2_text column k);  else W_query_vis[k,:]
//   [6H,7H) conditional only: W_query_vis[k,:] applied to the conditional query
// Backward uses the transposed ownership (thread (s,k): rows s*H..s*H+H-1, column k).
#include "step.h"

namespace gscan {

template <int H>
__device__ __forceinline__ float dot_reg_lds(const float (&w)[H], const float *v) {
    static_assert(H % 4 == 0, "hidden size must be a multiple of 4");
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
#pragma unroll
    for (int i = 0; i < H / 4; ++i) {
        const float4 x = v4[i];
        a0 = fmaf(w[4 * i + 0], x.x, a0);
        a1 = fmaf(w[4 * i + 1], x.y, a1);
        a2 = fmaf(w[4 * i + 2], x.z, a2);
        a3 = fmaf(w[4 * i + 3], x.w, a3);
    }
    return (a0 + a1) + (a2 + a3);
}


// Diagnostic phase stamps (off unless DecoderArgs::stamps is set): thread 0 of workgroup 0 adds the cycles since
// the previous stamp to slot i.  Shares, not absolute times, are what to read from them.
#define GSCAN_STAMP(i)                                                         \
    if (a.stamps && blockIdx.x == 0 && tid == 0) {                             \
        const long long now_ = clock64();                                      \
        stamp_acc[i] += (float)(now_ - stamp_prev);                            \
        stamp_prev = now_;                                                     \
    }

// sum_{m<n} al[m] * mat[m*stride]: four independent chains so the LDS reads issue back to back
__device__ __forceinline__ float weighted_sum(const float *al, const float *mat, int stride, int n) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int m = 0;
    for (; m + 4 <= n; m += 4) {
        a0 = fmaf(al[m + 0], mat[(m + 0) * stride], a0);
        a1 = fmaf(al[m + 1], mat[(m + 1) * stride], a1);
        a2 = fmaf(al[m + 2], mat[(m + 2) * stride], a2);
        a3 = fmaf(al[m + 3], mat[(m + 3) * stride], a3);
    }
    for (; m < n; ++m) a0 = fmaf(al[m], mat[m * stride], a0);
    return (a0 + a1) + (a2 + a3);
}

// Additive-attention scores s_m = v . tanh(q + PK_m) for m < n: each wave takes m = wave, wave+nwave, ...
// four at a time (partials first, then four independent DPP reductions).
template <int H>
__device__ __forceinline__ void attention_scores(const float *v_s, const float *q_s, const float *pk, int n,
                                                 float *sc_s, int wave, int nwave, int lane) {
    for (int m0 = wave; m0 < n; m0 += 4 * nwave) {
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + i * nwave;
            p[i] = 0.f;
            if (m < n)
                for (int kk = lane; kk < H; kk += 64) p[i] += v_s[kk] * tanhf_(q_s[kk] + pk[m * H + kk]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + i * nwave;
            if (m < n) {                      // wave-uniform
                const float t = wave_sum(p[i]);
                if (lane == 0) sc_s[m] = t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS carve shared by both kernels (floats).  Everything is in the dynamic region so the base
// stays 16-byte aligned (all offsets are multiples of 4 floats).
// ------------------------------------------------------------------------------------------
struct DecoderLds {
    int uv, pkv, ut, pkt, u2t, dpkv, dpkt, vec, total;
};
__host__ __device__ inline DecoderLds decoder_lds(int H, int L, int M, bool cond, bool backward) {
    DecoderLds o;
    int p = 0;
    o.uv = p;  p += M * 4 * H;
    o.pkv = p; p += M * H;
    o.ut = p;  p += L * 4 * H;
    o.pkt = p; p += L * H;
    o.u2t = p; p += cond ? L * H : 0;
    o.dpkv = p; p += backward ? M * H : 0;
    o.dpkt = p; p += backward ? L * H : 0;
    o.vec = p;
    p += 26 * H + 256;               // small vectors (V_* below) + three 64-float slots
    o.total = p;
    return o;
}
// offsets inside the small-vector region, in units of H floats (then two 64-float slots)
enum { V_H = 0, V_QT = 1, V_ZQ = 2, V_Q2 = 3, V_QV = 4, V_VT = 5, V_VV = 6, V_GATE = 7 /*4H*/, V_D = 11 /*6H*/,
       V_EXC = 17, V_EXS = 18, V_PART = 19 /* 6H: partial sums */, V_END = 26 };

template <int H, bool COND>
constexpr int decoder_threads() { return (((COND ? 7 : 6) * H + 63) / 64) * 64; }

template <int H, bool COND>
__global__ __launch_bounds__((decoder_threads<H, COND>())) void decoder_fwd_kernel(DecoderArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    const int T = a.T, L = a.L, M = a.M;
    const DecoderLds o = decoder_lds(H, L, M, COND, false);
    float *Uv = smem + o.uv, *PKv = smem + o.pkv, *Ut = smem + o.ut, *PKt = smem + o.pkt, *U2t = smem + o.u2t;
    float *vec = smem + o.vec;
    float *h_s = vec + V_H * H, *qt_s = vec + V_QT * H, *zq_s = vec + V_ZQ * H, *q2_s = vec + V_Q2 * H,
          *qv_s = vec + V_QV * H, *vt_s = vec + V_VT * H, *vv_s = vec + V_VV * H, *gate_s = vec + V_GATE * H;
    float *sc_s = vec + V_END * H, *al_s = vec + V_END * H + 64, *stamp_acc = vec + V_END * H + 192;
    long long stamp_prev = a.stamps ? clock64() : 0;
    if (tid < 16) stamp_acc[tid] = 0.f;
    int len = a.cmd_lengths[b];
    len = max(1, min(len, L));

    // ---- one-time loads: memories -> LDS, weights -> registers ------------------------------
    for (int i = tid; i < M * 4 * H; i += blockDim.x) Uv[i] = a.u_v[(int64_t)b * M * 4 * H + i];
    for (int i = tid; i < M * H; i += blockDim.x) PKv[i] = a.pk_v[(int64_t)b * M * H + i];
    for (int i = tid; i < L * 4 * H; i += blockDim.x) Ut[i] = a.u_t[(int64_t)b * L * 4 * H + i];
    for (int i = tid; i < L * H; i += blockDim.x) PKt[i] = a.pk_t[(int64_t)b * L * H + i];
    if (COND)
        for (int i = tid; i < L * H; i += blockDim.x) U2t[i] = a.u2_t[(int64_t)b * L * H + i];
    if (tid < H) {
        h_s[tid] = a.hprev[(int64_t)b * T * H + tid];
        vt_s[tid] = a.v_t[tid];
        vv_s[tid] = a.v_v[tid];
    }
    const int role = tid / H, k = tid % H;     // role 0..3 gates, 4 text query, 5 q2k|vis, 6 vis (cond)
    float w[H];
    float bq = 0.f;
    {
        const float *src = nullptr;
        if (role < 4) src = a.w_hh + (int64_t)tid * H;
        else if (role == 4) src = a.w_qt + (int64_t)k * H;
        else if (role == 5) src = COND ? a.w_q2k + (int64_t)k * 2 * H : a.w_qv + (int64_t)k * H;
        else if (role == 6 && COND) src = a.w_qv + (int64_t)k * H;
        if (src) {
#pragma unroll
            for (int i = 0; i < H; ++i) w[i] = src[i];
        } else {
#pragma unroll
            for (int i = 0; i < H; ++i) w[i] = 0.f;
        }
        if (COND && role == 5) bq = a.b_q2k[k];
    }
    float c = (tid < H) ? a.hprev[(int64_t)b * T * H + tid] : 0.f;   // c0 = h0 (seq2seq_model.py:494-504)
    float att_acc = 0.f;                                              // wave 0, lane m
    lds_barrier();

    for (int t = 0; t < T; ++t) {
        GSCAN_STAMP(0)
        const unsigned bt = (unsigned)b * T + t;       // 32-bit offsets: B*T*4H < 2^31 is checked on the host
        // gate input from the embedding (global, issued early; consumed in phase G)
        const float ge = (role < 4) ? a.ge[bt * 4 * H + tid] : 0.f;

        // ---- A: everything that multiplies h_{t-1} -------------------------------------------
        float gh = 0.f;
        if (role < 6) {
            const float acc = dot_reg_lds<H>(w, h_s);
            if (role < 4) gh = acc;
            else if (role == 4) { qt_s[k] = acc; a.qt[bt * H + k] = acc; }
            else if (COND) zq_s[k] = acc;
            else { qv_s[k] = acc; a.qv[bt * H + k] = acc; }
        }
        lds_barrier();
        GSCAN_STAMP(1)

        // ---- B: textual scores s_m = v . tanh(q + PK_m), m < len (seq2seq_model.py:129-135) --
        attention_scores<H>(vt_s, qt_s, PKt, len, sc_s, wave, nwave, lane);
        lds_barrier();
        GSCAN_STAMP(2)
        if (wave == 0) {
            const float x = (lane < len) ? sc_s[lane] : -INFINITY;
            const float mx = wave_max(x);
            const float e = (lane < len) ? expf(x - mx) : 0.f;
            const float al = e / wave_sum(e);
            al_s[lane] = al;
            if (lane < L) a.alpha_c[bt * L + lane] = al;
        }
        lds_barrier();
        GSCAN_STAMP(3)

        // ---- C: textual context and its images under W_ih / W_q2k ----------------------------
        float uc = 0.f;
        if (role < 4) {
            uc = weighted_sum(al_s, Ut + tid, 4 * H, len);
        } else if (role == 4) {
            a.s[bt * 4 * H + H + k] = weighted_sum(al_s, PKt + k, H, len);
        } else if (COND && role == 5) {
            const float u2 = weighted_sum(al_s, U2t + k, H, len);
            const float q = tanhf_(zq_s[k] + u2 + bq);        // seq2seq_model.py:394-396
            q2_s[k] = q;
            a.q2[bt * H + k] = q;
        }
        if (COND) {
            lds_barrier();
            GSCAN_STAMP(4)
            // ---- D: visual query from the conditional query ---------------------------------
            if (role == 6) {
                const float acc = dot_reg_lds<H>(w, q2_s);
                qv_s[k] = acc;
                a.qv[bt * H + k] = acc;
            }
        }
        lds_barrier();
        GSCAN_STAMP(5)

        // ---- E: visual scores over all M cells (no mask: every row has M memories) -----------
        attention_scores<H>(vv_s, qv_s, PKv, M, sc_s, wave, nwave, lane);
        lds_barrier();
        GSCAN_STAMP(6)
        if (wave == 0) {
            const float x = (lane < M) ? sc_s[lane] : -INFINITY;
            const float mx = wave_max(x);
            const float e = (lane < M) ? expf(x - mx) : 0.f;
            const float al = e / wave_sum(e);
            al_s[lane] = al;
            if (lane < M) a.alpha_s[bt * M + lane] = al;
            att_acc += al;                                     // seq2seq_model.py:479,490
        }
        lds_barrier();
        GSCAN_STAMP(7)

        // ---- F+G: visual context, gate pre-activations, activations --------------------------
        if (role < 4) {
            const float us = weighted_sum(al_s, Uv + tid, 4 * H, M);
            const float pre = ge + gh + uc + us;
            const float g = (role == 2) ? tanhf_(pre) : sigmoidf_(pre);
            gate_s[tid] = g;
            a.gates[bt * 4 * H + tid] = g;
        } else if (role == 4) {
            a.s[bt * 4 * H + 2 * H + k] = weighted_sum(al_s, PKv + k, H, M);
        }
        lds_barrier();
        GSCAN_STAMP(8)

        // ---- H: cell update (seq2seq_model.py:414) ---------------------------------------------
        if (tid < H) {
            const float ig = gate_s[tid], fg = gate_s[H + tid], gg = gate_s[2 * H + tid], og = gate_s[3 * H + tid];
            c = fg * c + ig * gg;
            const float h = og * tanhf_(c);
            h_s[tid] = h;
            a.cells[bt * H + tid] = c;
            a.s[bt * 4 * H + 3 * H + tid] = h;
            if (t + 1 < T) a.hprev[(bt + 1) * H + tid] = h;
        }
        lds_barrier();
        GSCAN_STAMP(9)
    }
    if (wave == 0 && lane < M) a.att_sum[(int64_t)b * M + lane] = att_acc;
    if (a.stamps && blockIdx.x == 0 && tid < 16) a.stamps[tid] = stamp_acc[tid];
}

// ------------------------------------------------------------------------------------------
// Backward through time.  Emits per-step pre-activation gradients (delta for the LSTM gates,
// dzq for the conditional query, dqt / dqv for the projected queries); every parameter
// gradient is a dense GEMM over those afterwards.  Key gradients along the score path are
// accumulated in LDS across the T steps and written once; the value path
// (dPK += alpha^T . dctx) is a separate batched product outside (it needs W_ih^T . delta,
// which is again a dense GEMM).
// ------------------------------------------------------------------------------------------
template <int H, bool COND>
__global__ __launch_bounds__((decoder_threads<H, COND>())) void decoder_bwd_kernel(DecoderArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    const int T = a.T, L = a.L, M = a.M;
    const DecoderLds o = decoder_lds(H, L, M, COND, true);
    float *Uv = smem + o.uv, *PKv = smem + o.pkv, *Ut = smem + o.ut, *PKt = smem + o.pkt, *U2t = smem + o.u2t;
    float *dPKv = smem + o.dpkv, *dPKt = smem + o.dpkt;
    float *vec = smem + o.vec;
    float *dh_s = vec + V_H * H, *qt_s = vec + V_QT * H, *q2_s = vec + V_Q2 * H, *qv_s = vec + V_QV * H,
          *vt_s = vec + V_VT * H, *vv_s = vec + V_VV * H, *dqv_s = vec + V_ZQ * H, *d_s = vec + V_D * H,
          *exc_s = vec + V_EXC * H, *exs_s = vec + V_EXS * H, *part_s = vec + V_PART * H;
    float *sc_s = vec + V_END * H, *al_s = vec + V_END * H + 64, *datt_s = vec + V_END * H + 128,
          *stamp_acc = vec + V_END * H + 192;
    long long stamp_prev = a.stamps ? clock64() : 0;
    if (tid < 16) stamp_acc[tid] = 0.f;
    int len = a.cmd_lengths[b];
    len = max(1, min(len, L));
    const int nchunk = min(6, (int)blockDim.x / H);        // (chunk, k) ownership of key pairs; part_s holds 6H

    for (int i = tid; i < M * 4 * H; i += blockDim.x) Uv[i] = a.u_v[(int64_t)b * M * 4 * H + i];
    for (int i = tid; i < M * H; i += blockDim.x) { PKv[i] = a.pk_v[(int64_t)b * M * H + i]; dPKv[i] = 0.f; }
    for (int i = tid; i < L * 4 * H; i += blockDim.x) Ut[i] = a.u_t[(int64_t)b * L * 4 * H + i];
    for (int i = tid; i < L * H; i += blockDim.x) { PKt[i] = a.pk_t[(int64_t)b * L * H + i]; dPKt[i] = 0.f; }
    if (COND)
        for (int i = tid; i < L * H; i += blockDim.x) U2t[i] = a.u2_t[(int64_t)b * L * H + i];
    if (tid < H) { dh_s[tid] = 0.f; vt_s[tid] = a.v_t[tid]; vv_s[tid] = a.v_v[tid]; }
    if (tid < 64) datt_s[tid] = (a.datt && tid < M) ? a.datt[(int64_t)b * M + tid] : 0.f;

    // transposed weights: thread (seg, k) holds rows seg*H.. of [W_hh(4H) | W_qt | W_q2k_h or W_qv], column k
    const int seg = tid / H, k = tid % H;
    float wt[H];
    {
        const float *src = nullptr;
        int64_t stride = H;
        if (seg < 4) src = a.w_hh + (int64_t)seg * H * H + k;
        else if (seg == 4) src = a.w_qt + k;
        else if (seg == 5) { src = COND ? a.w_q2k + k : a.w_qv + k; stride = COND ? 2 * H : H; }
        else if (seg == 6 && COND) src = a.w_qv + k;
        if (src) {
#pragma unroll
            for (int j = 0; j < H; ++j) wt[j] = src[(int64_t)j * stride];
        } else {
#pragma unroll
            for (int j = 0; j < H; ++j) wt[j] = 0.f;
        }
    }
    float dc = 0.f, dvv_acc = 0.f, dvt_acc = 0.f;
    lds_barrier();

    for (int t = T - 1; t >= 0; --t) {
        GSCAN_STAMP(0)
        const unsigned bt = (unsigned)b * T + t;
        // ---- 1: LSTM cell backward -----------------------------------------------------------
        if (tid < H) {
            const float dh = dh_s[tid] + a.ds[bt * 4 * H + 3 * H + tid];
            const float *g = a.gates + bt * 4 * H;
            const float ig = g[tid], fg = g[H + tid], gg = g[2 * H + tid], og = g[3 * H + tid];
            const float c = a.cells[bt * H + tid];
            const float c_prev = (t > 0) ? a.cells[(bt - 1) * H + tid] : a.hprev[(int64_t)b * T * H + tid];
            const float tc = tanhf_(c);
            const float dct = dc + dh * og * (1.f - tc * tc);
            const float di = dct * gg * ig * (1.f - ig);
            const float df = dct * c_prev * fg * (1.f - fg);
            const float dg = dct * ig * (1.f - gg * gg);
            const float d_o = dh * tc * og * (1.f - og);
            dc = dct * fg;
            d_s[tid] = di; d_s[H + tid] = df; d_s[2 * H + tid] = dg; d_s[3 * H + tid] = d_o;
            float *dp = a.delta + bt * 4 * H;
            dp[tid] = di; dp[H + tid] = df; dp[2 * H + tid] = dg; dp[3 * H + tid] = d_o;
            // external gradients wrt the two contexts (output head) and the saved queries
            exc_s[tid] = a.ds[bt * 4 * H + H + tid];
            exs_s[tid] = a.ds[bt * 4 * H + 2 * H + tid];
            qt_s[tid] = a.qt[bt * H + tid];
            qv_s[tid] = a.qv[bt * H + tid];
            if (COND) q2_s[tid] = a.q2[bt * H + tid];
        }
        lds_barrier();
        GSCAN_STAMP(1)

        // ---- 2: d alpha_vis[m] = delta . U_vis[m] + dctx_vis(ext) . PK_vis[m] + d att_sum[m] ---
        for (int m0 = wave; m0 < M; m0 += 4 * nwave) {
            float p[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + i * nwave;
                p[i] = 0.f;
                if (m < M) {
                    for (int j = lane; j < 4 * H; j += 64) p[i] = fmaf(d_s[j], Uv[m * 4 * H + j], p[i]);
                    for (int kk = lane; kk < H; kk += 64) p[i] = fmaf(exs_s[kk], PKv[m * H + kk], p[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + i * nwave;
                if (m < M) {
                    const float tsum = wave_sum(p[i]);
                    if (lane == 0) sc_s[m] = tsum + datt_s[m];
                }
            }
        }
        lds_barrier();
        GSCAN_STAMP(2)
        if (wave == 0) {   // softmax backward: ds = alpha * (dalpha - sum alpha dalpha)
            const float al = (lane < M) ? a.alpha_s[bt * M + lane] : 0.f;
            const float da = (lane < M) ? sc_s[lane] : 0.f;
            const float dot = wave_sum(al * da);
            al_s[lane] = al * (da - dot);
        }
        lds_barrier();
        GSCAN_STAMP(3)

        // ---- 3: through tanh(q + PK) of the visual scores; (chunk,k) owns pairs (m,k) ----------
        if (seg < nchunk) {
            float pdq = 0.f;
            for (int m = seg; m < M; m += nchunk) {
                const float th = tanhf_(qv_s[k] + PKv[m * H + k]);
                const float term = al_s[m] * vv_s[k] * (1.f - th * th);
                dPKv[m * H + k] += term;
                pdq += term;
                dvv_acc = fmaf(al_s[m], th, dvv_acc);
            }
            part_s[seg * H + k] = pdq;
        }
        lds_barrier();
        GSCAN_STAMP(4)
        if (tid < H) {
            float dq = 0.f;
            for (int cch = 0; cch < nchunk; ++cch) dq += part_s[cch * H + tid];
            dqv_s[tid] = dq;
            a.dqv[bt * H + tid] = dq;
            if (!COND) d_s[5 * H + tid] = dq;               // visual query came straight from h
        }
        lds_barrier();
        GSCAN_STAMP(5)

        // ---- 4: conditional query: dq2 = W_qv^T dqv, through tanh ------------------------------
        if (COND) {
            if (seg == 6) {
                const float dq2 = dot_reg_lds<H>(wt, dqv_s);
                const float q = q2_s[k];
                const float dz = dq2 * (1.f - q * q);
                d_s[5 * H + k] = dz;
                a.dzq[bt * H + k] = dz;
            }
            lds_barrier();
            GSCAN_STAMP(6)
        }

        // ---- 5: d alpha_text[m] = delta . U_text[m] + dzq . U2_text[m] + dctx_text(ext) . PK_text[m]
        for (int m0 = wave; m0 < len; m0 += 4 * nwave) {
            float p[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + i * nwave;
                p[i] = 0.f;
                if (m < len) {
                    for (int j = lane; j < 4 * H; j += 64) p[i] = fmaf(d_s[j], Ut[m * 4 * H + j], p[i]);
                    for (int kk = lane; kk < H; kk += 64) {
                        p[i] = fmaf(exc_s[kk], PKt[m * H + kk], p[i]);
                        if (COND) p[i] = fmaf(d_s[5 * H + kk], U2t[m * H + kk], p[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + i * nwave;
                if (m < len) {
                    const float tsum = wave_sum(p[i]);
                    if (lane == 0) sc_s[m] = tsum;
                }
            }
        }
        lds_barrier();
        GSCAN_STAMP(7)
        if (wave == 0) {
            const float al = (lane < len) ? a.alpha_c[bt * L + lane] : 0.f;
            const float da = (lane < len) ? sc_s[lane] : 0.f;
            const float dot = wave_sum(al * da);
            al_s[lane] = al * (da - dot);
        }
        lds_barrier();
        GSCAN_STAMP(8)

        // ---- 6: through tanh(q + PK) of the textual scores -------------------------------------
        if (seg < nchunk) {
            float pdq = 0.f;
            for (int m = seg; m < len; m += nchunk) {
                const float th = tanhf_(qt_s[k] + PKt[m * H + k]);
                const float term = al_s[m] * vt_s[k] * (1.f - th * th);
                dPKt[m * H + k] += term;
                pdq += term;
                dvt_acc = fmaf(al_s[m], th, dvt_acc);
            }
            part_s[seg * H + k] = pdq;
        }
        lds_barrier();
        GSCAN_STAMP(9)
        if (tid < H) {
            float dq = 0.f;
            for (int cch = 0; cch < nchunk; ++cch) dq += part_s[cch * H + tid];
            d_s[4 * H + tid] = dq;
            a.dqt[bt * H + tid] = dq;
        }
        lds_barrier();
        GSCAN_STAMP(10)

        // ---- 7: dh_{t-1} = [W_hh | W_qt | W_q2k_h or W_qv]^T . [delta | dqt | dzq or dqv] ------
        float part = 0.f;
        if (seg < 6) part = dot_reg_lds<H>(wt, d_s + seg * H);
        if (seg < 6) part_s[seg * H + k] = part;            // phase 6's partials were consumed before the last barrier
        lds_barrier();
        GSCAN_STAMP(11)
        if (tid < H) {
            float dh = 0.f;
#pragma unroll
            for (int sgi = 0; sgi < 6; ++sgi) dh += part_s[sgi * H + tid];
            dh_s[tid] = dh;
        }
        lds_barrier();
        GSCAN_STAMP(12)
    }

    // ---- epilogue: initial-state gradient through the bridge tanh, key and energy gradients ----
    if (tid < H) {
        const float h0 = a.hprev[(int64_t)b * T * H + tid];
        a.dh0[(int64_t)b * H + tid] = (dh_s[tid] + dc) * (1.f - h0 * h0);   // h0 = c0 = tanh(.) (model.py:195)
    }
    for (int i = tid; i < M * H; i += blockDim.x) a.dpk_v[(int64_t)b * M * H + i] = dPKv[i];
    for (int i = tid; i < L * H; i += blockDim.x) a.dpk_t[(int64_t)b * L * H + i] = (i / H < len) ? dPKt[i] : 0.f;
    lds_barrier();
    if (seg < nchunk) part_s[seg * H + k] = dvv_acc;
    lds_barrier();
    if (tid < H) {
        float x = 0.f;
        for (int cch = 0; cch < nchunk; ++cch) x += part_s[cch * H + tid];
        atomicAdd(&a.dv_v[tid], x);
    }
    lds_barrier();
    if (seg < nchunk) part_s[seg * H + k] = dvt_acc;
    lds_barrier();
    if (tid < H) {
        float x = 0.f;
        for (int cch = 0; cch < nchunk; ++cch) x += part_s[cch * H + tid];
        atomicAdd(&a.dv_t[tid], x);
    }
    if (a.stamps && blockIdx.x == 0 && tid < 16) a.stamps[tid] = stamp_acc[tid];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
constexpr size_t kLdsLimit = 160 * 1024;

template <int H, bool COND>
static int launch_decoder(bool backward, int B, const DecoderArgs &a, hipStream_t stream) {
    constexpr int nt = decoder_threads<H, COND>();
    GSCAN_CHECK(nt <= 1024, "decoder: hidden size %d needs %d threads per row (> 1024)", H, nt);
    const DecoderLds o = decoder_lds(H, a.L, a.M, COND, backward);
    const size_t bytes = (size_t)o.total * sizeof(float);
    GSCAN_CHECK(bytes <= kLdsLimit,
                "decoder: a row's memories need %zu bytes of LDS (> 160 KiB): grid cells=%d command length=%d hidden=%d",
                bytes, a.M, a.L, H);
    // Algorithmic MACs of one decoder step that this kernel owns (SURVEY.md §8d MAC_step minus the
    // embedding part of the LSTM input and the output head, which run as GEMMs outside the loop):
    // query projections, both score/context reductions, and the [ctx_text|ctx_vis|h] part of the LSTM.
    const double macs = (double)H * H + 2.0 * a.L * H + (COND ? 2.0 * H * H : 0.0) + (double)H * H +
                        2.0 * a.M * H + 4.0 * H * 3.0 * H;
    ProbeScope probe(backward ? P_DECODER_BWD : P_DECODER_FWD, stream, 2.0 * macs * B * a.T);
    if (backward) {
        static bool attr_set = false;
        if (!attr_set) {
            GSCAN_HIP(hipFuncSetAttribute((const void *)decoder_bwd_kernel<H, COND>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
            attr_set = true;
        }
        hipLaunchKernelGGL((decoder_bwd_kernel<H, COND>), dim3(B), dim3(nt), bytes, stream, a);
        GSCAN_LAUNCHED("decoder_bwd_kernel");
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            GSCAN_HIP(hipFuncSetAttribute((const void *)decoder_fwd_kernel<H, COND>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
            attr_set = true;
        }
        hipLaunchKernelGGL((decoder_fwd_kernel<H, COND>), dim3(B), dim3(nt), bytes, stream, a);
        GSCAN_LAUNCHED("decoder_fwd_kernel");
    }
    return 0;
}

#define GSCAN_DEC_HIDDEN_SIZES(X) X(20) X(32) X(64) X(100)

bool decoder_hidden_supported(int h) {
#define X(n) if (h == n) return true;
    GSCAN_DEC_HIDDEN_SIZES(X)
#undef X
    return false;
}

size_t decoder_lds_bytes(int H, int L, int M, bool cond, bool backward) {
    return (size_t)decoder_lds(H, L, M, cond, backward).total * sizeof(float);
}

int decoder_run(bool backward, int B, int H, bool cond, const DecoderArgs &a, hipStream_t stream) {
    GSCAN_CHECK(B > 0 && a.T > 0 && a.L > 0 && a.M > 0, "decoder: bad dims B=%d T=%d L=%d M=%d", B, a.T, a.L, a.M);
    GSCAN_CHECK(a.L <= 64 && a.M <= 64, "decoder: at most 64 command tokens / grid cells per row (L=%d, cells=%d)",
                a.L, a.M);
    switch (H) {
#define X(n) case n: return cond ? launch_decoder<n, true>(backward, B, a, stream) : launch_decoder<n, false>(backward, B, a, stream);
        GSCAN_DEC_HIDDEN_SIZES(X)
#undef X
        default: break;
    }
    GSCAN_CHECK(false, "decoder_hidden_size %d has no compiled kernel (supported: 20 32 64 100)", H);
}

}  // namespace gscan
