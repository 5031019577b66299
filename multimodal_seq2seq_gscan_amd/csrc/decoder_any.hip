// The decoder recurrence for ANY shape (seq2seq/seq2seq_model.py:359-492): every --decoder_hidden_size, command
// length and grid size the reference accepts, where decoder.hip's kernels (weights in registers, memories in LDS, the
// attention distribution in the 64 lanes of a wave) are compiled for hidden sizes up to 100 and hold at most 64
// memories per attention.  One 1024-thread workgroup per batch row, and
//   * the weights are STREAMED from L2 every step, in the reference's own [out, in] layouts (no register images):
//     a product with a vector is either "rows" (sixteen lanes per output row, 256-byte pieces of the row, a DPP sum)
//     or "columns" (a lane per output column of the transposed product, coalesced across lanes);
//   * the projected keys are read from global memory (no LDS residency: any number of memories), the attention
//     distribution lives in LDS;
//   * nothing depends on the hidden size at compile time except a 16-byte-load variant for multiples of 4.
// Same arguments (DecoderArgs), same saved activations and the same outputs as the fast kernels — the two are
// interchangeable per launch, and every launch behind them (the dS += product, keys backward, the weight-gradient
// GEMMs) is unchanged.  At the benchmark shape this path is ~10 x slower than the fast one; it is what runs when the
// fast one has no kernel for the shape (decoder_run picks).
#include "anyshape.h"

// The step loops below are long and nearly everything in them is an address derived from a handful of loop-invariant
// scalars; hoisted out of the loop by the compiler those were ~290 scalar registers, spilled to vector registers and from
// there to scratch (bwd: 44 VGPR spills, 164 B of scratch per lane).  LAUNDER makes a scalar opaque at the top of an
// iteration, so what derives from it is recomputed per step (a few dozen SALU operations against 30+ us of streaming).
#define LAUNDER(x) asm volatile("" : "+s"(x))
// Barriers inside the two time loops are lds_barrier() (s_waitcnt lgkmcnt(0) + s_barrier, common.h), not __syncthreads():
// everything that crosses threads inside a step goes through LDS (the resident memories included: FLAT accesses count on
// lgkmcnt), while __syncthreads() also drains vmcnt — i.e. every one of the ~12 (forward) / ~20 (backward) barriers of a step
// waited for the global STORES of saved activations issued in its phase, a store round trip each.  (Round 5; the resident
// kernels have done the same since round 1.)

namespace gscan {

// softmax of sc[0..n) in place (LDS) by wave 0; returns nothing, the caller synchronises
__device__ __forceinline__ void softmax_lds(float *sc, int n) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        float mx = -INFINITY;
        for (int m = tid; m < n; m += 64) mx = fmaxf(mx, sc[m]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int m = tid; m < n; m += 64) { const float e = __expf(sc[m] - mx); sc[m] = e; sum += e; }
        const float inv = 1.f / wave_sum(sum);
        for (int m = tid; m < n; m += 64) sc[m] *= inv;
    }
}

// additive-attention scores sc[m] = v . tanh(q + PK[m]) for m < n: a wave per memory, lanes over the features
__device__ __forceinline__ void scores_any(const float *v_s, const float *q_s, const float *__restrict__ pk, int n, int H,
                                           float *sc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m0 = 0; m0 < n; m0 += kAnyWaves) {
        const int m = m0 + wave, mc = min(m, n - 1);
        float p = 0.f;
        for (int k = lane; k < H; k += 64) p = fmaf(v_s[k], tanhf_(q_s[k] + pk[(int64_t)mc * H + k]), p);
        p = wave_sum(p);
        if (lane == 0 && m < n) sc[m] = p;
    }
}

// Round 5: the row's memories RESIDENT IN LDS where they fit (the streaming kernels took them from global memory every
// step so that any number of memories works: ~10 dependent L2 round trips per forward step, ~20 per backward step, 1.5-2 us
// each with 256 workgroups loading at once — the step was a chain of load latencies, not of streamed bytes).  What a kernel
// holds is a PREFIX of: projected keys (both attentions) | backward: their score-path gradients | textual gate images U_t,
// U2_t | visual gate images U_v; the rest stays in global memory.  The arrays are reached through generic pointers (an LDS
// address or a global one: FLAT loads), so one body serves every residency.
enum { kAnyMemPK = 1, kAnyMemDPK = 2, kAnyMemUT = 4, kAnyMemUV = 8 };
struct AnyResidency { int flags, pkt, pkv, dpkt, dpkv, ut, u2t, uv, floats; };
__host__ __device__ inline AnyResidency any_residency(int H, int L, int M, bool cond, bool use_u, bool backward, int base_floats,
                                                     int limit_floats) {
    AnyResidency r{};
    int p = (base_floats + 3) & ~3;
    auto take = [&](int n) { const int at = p; p += (n + 3) & ~3; return at; };
    const int start = p;
    if (p + (L + M) * H + 8 <= limit_floats) {
        r.flags |= kAnyMemPK; r.pkt = take(L * H); r.pkv = take(M * H);
        if (!backward || p + (L + M) * H + 8 <= limit_floats) {
            if (backward) { r.flags |= kAnyMemDPK; r.dpkt = take(L * H); r.dpkv = take(M * H); }
            if (use_u && p + L * (cond ? 5 : 4) * H + 8 <= limit_floats) {
                r.flags |= kAnyMemUT; r.ut = take(L * 4 * H); r.u2t = take(cond ? L * H : 0);
                if (p + M * 4 * H + 4 <= limit_floats) { r.flags |= kAnyMemUV; r.uv = take(M * 4 * H); }
            }
        }
    }
    r.floats = p - start;
    return r;
}
constexpr int kAnyLdsFloats = 160 * 1024 / 4;
// cooperative global -> LDS copy (n floats, both 16-byte aligned when n % 4 == 0)
__device__ __forceinline__ void any_stage(float *dst, const float *__restrict__ src, int n) {
    if ((n & 3) == 0) {
        for (int i = threadIdx.x; i < n / 4; i += kAnyThreads)
            reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i];
    } else {
        for (int i = threadIdx.x; i < n; i += kAnyThreads) dst[i] = src[i];
    }
}

struct AnyLds { int hc, qt, q2, qv, vt, vv, bq, pre, cell, sc, misc, scr, total; };
__host__ __device__ inline AnyLds any_lds_fwd(int H, int L, int M) {
    const int HP = (H + 3) / 4 * 4, NM = ((L > M ? L : M) + 3) / 4 * 4;
    AnyLds o;
    int p = 0;
    o.hc = p; p += 3 * HP + 4;         // [h | ctx_text | ctx_vis], contiguous: the inputs of W_q2k and of W_ih[:, H:3H]
    o.qt = p; p += HP; o.q2 = p; p += HP; o.qv = p; p += HP; o.vt = p; p += HP; o.vv = p; p += HP; o.bq = p; p += HP;
    o.pre = p; p += 4 * HP;            // gate pre-activations
    o.cell = p; p += HP;
    o.sc = p; p += NM;
    o.misc = p; p += 64;
    o.scr = p; p += kAnyThreads;       // partial sums of the column products
    o.total = p;
    return o;
}

// ------------------------------------------------------------------------------------------
// forward (teacher forcing, or GREEDY: the row feeds its own argmax back and stops at <EOS>; predict.py:101-112)
// ------------------------------------------------------------------------------------------
// diagnostic phase stamps (DecoderArgs::stamps set: tools/decoder_stamps.py): thread 0 of workgroup 0 adds the cycles since
// the previous stamp to slot i of an LDS table, written out at the end of the kernel
#define ANY_STAMP(i) if (a.stamps && blockIdx.x == 0 && tid == 0) { const long long n_ = clock64(); stamp_s[i] += (float)(n_ - stamp_prev); stamp_prev = n_; }
template <bool V4, bool GREEDY>
__global__ __launch_bounds__(kAnyThreads) void decoder_fwd_any_kernel(DecoderArgs a, int H, int cond) {
    // (no TraceScope: with -DGSCAN_TRACE its destructor's branch on blockIdx makes this compiler's back end fail on the
    // kernel — 'illegal VGPR to SGPR copy' — and the in-kernel timeline is a tool for the resident kernels' schedule)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = a.T, L = a.L, M = a.M, V = a.V;
    const AnyLds o = any_lds_fwd(H, L, M);
    float *hc = smem + o.hc, *qt_s = smem + o.qt, *q2_s = smem + o.q2, *qv_s = smem + o.qv, *vt_s = smem + o.vt,
          *vv_s = smem + o.vv, *bq_s = smem + o.bq, *pre_s = smem + o.pre, *c_s = smem + o.cell, *sc = smem + o.sc,
          *scr = smem + o.scr;
    int *tok_s = reinterpret_cast<int *>(smem + o.misc);
    float *stamp_s = smem + o.misc + 16;
    long long stamp_prev = 0;
    if (tid < 16) stamp_s[tid] = 0.f;
    float *h_s = hc, *ctxt_s = hc + H, *ctxv_s = hc + 2 * H;
    int len = a.cmd_lengths[b];
    len = max(1, min(len, L));
    const float *pk_t = a.pk_t + (int64_t)b * L * H, *pk_v = a.pk_v + (int64_t)b * M * H;
    // gate images of the row's memories, U = PK . W_ih[:, ctx]^T (columns unit-major: 4 unit + gate) and, conditional,
    // U2 = PK_text . W_q2k[:, H:]^T: W_ih[:, ctx] . ctx = sum_m alpha_m U[m] (a context is a convex combination of
    // projected keys), so the 8 H^2 floats of W_ih's context columns are never streamed
    const float *u_t = a.u_t + (int64_t)b * L * 4 * H, *u2_t = a.u2_t + (int64_t)b * L * H, *u_v = a.u_v + (int64_t)b * M * 4 * H;
    // the reference's own parameter layouts (row-major [out, in]), carried in the argument struct by decoder_run
    const float *W_hh = a.any_w_hh, *W_ih = a.any_w_ih, *W_qt = a.any_w_qt, *W_qv = a.any_w_qv, *W_q2k = a.any_w_q2k;
    const bool use_u = a.any_use_u != 0;       // decoder_any_uses_gate_images: wide hidden sizes with few memories
    // the memories this launch keeps in LDS (any_residency; the host computed the same plan for the LDS size)
    const AnyResidency res = any_residency(H, L, M, cond != 0, use_u, false, o.total, a.any_lds_floats);
    if (res.flags & kAnyMemPK) { any_stage(smem + res.pkt, pk_t, L * H); any_stage(smem + res.pkv, pk_v, M * H); }
    if (res.flags & kAnyMemUT) { any_stage(smem + res.ut, u_t, L * 4 * H); if (cond) any_stage(smem + res.u2t, u2_t, L * H); }
    if (res.flags & kAnyMemUV) any_stage(smem + res.uv, u_v, M * 4 * H);

    for (int k = tid; k < H; k += kAnyThreads) {
        const float h0 = a.hprev[(int64_t)b * (GREEDY ? 1 : T) * H + k];
        h_s[k] = h0;
        c_s[k] = a.c0 ? a.c0[(int64_t)b * H + k] : h0;                 // c0 = h0 unless given (seq2seq_model.py:494-504)
        vt_s[k] = a.v_t[k];
        vv_s[k] = a.v_v[k];
        bq_s[k] = cond ? a.b_q2k[k] : 0.f;
    }
    if (tid == 0) tok_s[0] = GREEDY ? a.sos : 0;
    for (int m = tid; m < M; m += kAnyThreads) a.att_sum[(int64_t)b * M + m] = 0.f;
    __syncthreads();
    int steps_done = 0;

    int Hl = H, Ll = L, Ml = M, lenl = len;
    for (int t = 0; t < T; ++t) {
        LAUNDER(Hl); LAUNDER(Ll); LAUNDER(Ml); LAUNDER(lenl);
        const int H = Hl, L = Ll, M = Ml, len = lenl;
        const float *pk_t = (res.flags & kAnyMemPK) ? smem + res.pkt : a.pk_t + (int64_t)b * L * H;
        const float *pk_v = (res.flags & kAnyMemPK) ? smem + res.pkv : a.pk_v + (int64_t)b * M * H;
        const float *u_t = (res.flags & kAnyMemUT) ? smem + res.ut : a.u_t + (int64_t)b * L * 4 * H;
        const float *u2_t = (res.flags & kAnyMemUT) ? smem + res.u2t : a.u2_t + (int64_t)b * L * H;
        const float *u_v = (res.flags & kAnyMemUV) ? smem + res.uv : a.u_v + (int64_t)b * M * 4 * H;
        const int64_t bt = (int64_t)b * T + t;
        if (t == 0 && a.stamps && blockIdx.x == 0 && tid == 0) stamp_prev = clock64();
        // ---- everything that multiplies h_{t-1}: W_query_text h, the gates' recurrent part
        {
            const int64_t ge_row = GREEDY ? (int64_t)tok_s[0] : bt;    // greedy: row of the [V, 4H] table of the token fed in
            const float *ge = a.ge + ge_row * 4 * H;
            if (V4 && use_u) {      // one call over the concatenated [W_hh ; W_q2k[:, :H] or W_query_vis ; W_query_text] (any_wcat)
                matvec_rows_image(a.any_wcat_stream, 6 * H, H, h_s, [&](int r, float v) {
                    if (r < 4 * H) pre_s[r] = v + ge[r];
                    else if (r >= 5 * H) qt_s[r - 5 * H] = v;
                    else if (cond) q2_s[r - 4 * H] = v + bq_s[r - 4 * H];          // W_q2k[:, :H] h + b
                    else { qv_s[r - 4 * H] = v; if (!GREEDY) a.qv[bt * H + r - 4 * H] = v; }   // the visual query comes straight from h
                });
            } else {
                matvec_rows<V4>(W_qt, H, H, H, h_s, [&](int r, float v) { qt_s[r] = v; });
                matvec_rows<V4>(W_hh, H, 4 * H, H, h_s, [&](int r, float v) { pre_s[r] = v + ge[r]; });
                if (cond && use_u) matvec_rows<V4>(W_q2k, 2 * H, H, H, h_s, [&](int r, float v) { q2_s[r] = v + bq_s[r]; });   // W_q2k[:, :H] h + b
            }
        }
        lds_barrier();
        ANY_STAMP(0)
        // ---- textual attention (seq2seq_model.py:129-139)
        scores_any(vt_s, qt_s, pk_t, len, H, sc);
        lds_barrier();
        ANY_STAMP(1)
        softmax_lds(sc, len);
        lds_barrier();
        ANY_STAMP(2)
        for (int m = tid; m < L; m += kAnyThreads) a.alpha_c[bt * L + m] = m < len ? sc[m] : 0.f;
        // context = alpha . PK (:138-139) and, through the gate images, the textual part of the gates' input product and of the
        // conditional query (tanh(W_q2k [h; ctx_text] + b), :394-396): one pass over [PK_t | U_t | U2_t]
        {
            matvec_cols_arrays(ColsArray{pk_t, H, H}, ColsArray{u_t, 4 * H, use_u ? 4 * H : 0},
                               ColsArray{u2_t, H, (use_u && cond) ? H : 0}, len, sc, scr, [&](int j, int k, float v) {
                if (j == 0) {
                    ctxt_s[k] = v;
                    if (!GREEDY) { a.s[bt * 4 * H + H + k] = v; a.qt[bt * H + k] = qt_s[k]; }
                } else if (j == 1) {
                    pre_s[(k & 3) * H + (k >> 2)] += v;
                } else {
                    const float q = tanhf_(q2_s[k] + v);
                    q2_s[k] = q;
                    if (!GREEDY) a.q2[bt * H + k] = q;
                }
            });
        }
        if (use_u) {
        } else if (cond) {
            matvec_rows<V4>(W_q2k, 2 * H, H, 2 * H, hc, [&](int r, float v) {
                const float q = tanhf_(v + bq_s[r]);
                q2_s[r] = q;
                if (!GREEDY) a.q2[bt * H + r] = q;
            });
            lds_barrier();
        }
        ANY_STAMP(3)
        if (cond || !(V4 && use_u)) {                                  // (unconditional, one-call form: computed with the others above)
            matvec_rows<V4>(W_qv, H, H, H, cond ? q2_s : h_s, [&](int r, float v) {
                qv_s[r] = v;
                if (!GREEDY) a.qv[bt * H + r] = v;
            });
            lds_barrier();
        }
        ANY_STAMP(4)
        // ---- visual attention over all M cells
        scores_any(vv_s, qv_s, pk_v, M, H, sc);
        lds_barrier();
        ANY_STAMP(5)
        softmax_lds(sc, M);
        lds_barrier();
        ANY_STAMP(6)
        for (int m = tid; m < M; m += kAnyThreads) {
            a.alpha_s[bt * M + m] = sc[m];
            a.att_sum[(int64_t)b * M + m] += sc[m];                    // seq2seq_model.py:479,490 (this thread's element)
        }
        // visual context and (gate images) the visual part of the gates' input product: one pass over [PK_v | U_v]
        {
            matvec_cols_arrays(ColsArray{pk_v, H, H}, ColsArray{u_v, 4 * H, use_u ? 4 * H : 0}, ColsArray{pk_v, H, 0}, M, sc, scr,
                               [&](int j, int k, float v) {
                if (j == 0) {
                    ctxv_s[k] = v;
                    if (!GREEDY) a.s[bt * 4 * H + 2 * H + k] = v;
                } else {
                    pre_s[(k & 3) * H + (k >> 2)] += v;
                }
            });
        }
        ANY_STAMP(7)
        // ---- LSTM cell (seq2seq_model.py:414): the context part of the input product, then the gates
        if (use_u) {
        } else {
            matvec_rows<V4>(W_ih + H, 3 * H, 4 * H, 2 * H, ctxt_s, [&](int r, float v) { pre_s[r] += v; });
            lds_barrier();
        }
        ANY_STAMP(8)
        for (int u = tid; u < H; u += kAnyThreads) {
            const float ig = sigmoidf_(pre_s[u]), fg = sigmoidf_(pre_s[H + u]), gg = tanhf_(pre_s[2 * H + u]),
                        og = sigmoidf_(pre_s[3 * H + u]);
            const float c = fg * c_s[u] + ig * gg;
            const float h = og * tanhf_(c);
            c_s[u] = c;
            h_s[u] = h;
            if (!GREEDY) {
                a.gates[bt * 4 * H + u] = ig; a.gates[bt * 4 * H + H + u] = fg;
                a.gates[bt * 4 * H + 2 * H + u] = gg; a.gates[bt * 4 * H + 3 * H + u] = og;
                a.cells[bt * H + u] = c;
                a.s[bt * 4 * H + 3 * H + u] = h;
                if (t + 1 < T) a.hprev[(bt + 1) * H + u] = h;
            }
        }
        lds_barrier();
        if (GREEDY) {
            // output head on [e | ctx_text | ctx_vis | h] as the one matrix Wc = W_h2o . W_o2h ([V, 4H], S order), argmax
            // (the first of equal maxima), feed back, stop at <EOS>
            const int tok = tok_s[0];
            float *logit_s = pre_s;                                    // the gates are done with it
            for (int v = wave; v < V; v += kAnyWaves) {
                const float *wrow = a.head_wc + (int64_t)v * 4 * H;
                float p = 0.f;
                for (int k = lane; k < H; k += 64)
                    p += wrow[k] * a.dec_emb[(int64_t)tok * H + k] + wrow[H + k] * ctxt_s[k] + wrow[2 * H + k] * ctxv_s[k] +
                         wrow[3 * H + k] * h_s[k];
                p = wave_sum(p);
                if (lane == 0) logit_s[v] = p;
            }
            lds_barrier();
            if (tid == 0) {
                int best = 0;
                float top = logit_s[0];
                for (int v = 1; v < V; ++v)
                    if (logit_s[v] > top) { top = logit_s[v]; best = v; }
                tok_s[0] = best;
                a.tokens_out[bt] = best;
            }
            steps_done = t + 1;
            lds_barrier();
            if (tok_s[0] == a.eos) break;                              // uniform: every thread reads the same LDS word
        }
    }
    if (GREEDY) {
        if (tid == 0) a.steps_out[b] = steps_done;
        return;
    }
    if (a.h_last) for (int k = tid; k < H; k += kAnyThreads) a.h_last[(int64_t)b * H + k] = h_s[k];
    ANY_STAMP(9)
    if (a.stamps && blockIdx.x == 0 && tid < 16) a.stamps[tid] = stamp_s[tid];
    __syncthreads();                                                   // the row's S and att_sum are complete
    // ---- auxiliary head: log_softmax over the cells of the summed visual attention (model.py:205) and this row's
    //      get_auxiliary_loss term (model.py:162-164)
    if (a.aux_saved) {
        const float *att = a.att_sum + (int64_t)b * M;
        if (tid < 64) {
            float mx = -INFINITY;
            for (int m = tid; m < M; m += 64) mx = fmaxf(mx, att[m]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int m = tid; m < M; m += 64) sum += expf(att[m] - mx);
            const float lse = mx + logf(wave_sum(sum));
            for (int m = tid; m < M; m += 64) {
                a.aux_saved[(int64_t)b * M + m] = att[m] - lse;
                a.aux_out[(int64_t)b * M + m] = att[m] - lse;
            }
            if (a.row_stats && tid == 0) {
                const int64_t pos = a.positions ? a.positions[b] : (int64_t)-1;
                a.row_stats[4 * b + 2] = (pos >= 0 && pos < M) ? lse - att[pos] : 0.f;
            }
        }
    } else if (a.row_stats && tid == 0) {
        a.row_stats[4 * b + 2] = 0.f;
    }
    // ---- output head of the row's T steps: logits_t = Wc . S_t (a wave per logit), log_softmax (model.py:203), and the
    //      row's get_loss partial sums (model.py:147-160)
    for (int q = wave; q < T * V; q += kAnyWaves) {
        const int tt = q / V, v = q - tt * V;
        const float *srow = a.s + ((int64_t)b * T + tt) * 4 * H, *wrow = a.head_wc + (int64_t)v * 4 * H;
        float p = 0.f;
        for (int k = lane; k < 4 * H; k += 64) p = fmaf(wrow[k], srow[k], p);
        p = wave_sum(p);
        if (lane == 0) a.logits[((int64_t)b * T + tt) * V + v] = p;
    }
    __syncthreads();
    float nll_acc = 0.f, cnt_acc = 0.f;
    for (int tt = tid; tt < T; tt += kAnyThreads) {
        const float *row = a.logits + ((int64_t)b * T + tt) * V;
        float mx = -INFINITY;
        for (int j = 0; j < V; ++j) mx = fmaxf(mx, row[j]);
        float sum = 0.f;
        for (int j = 0; j < V; ++j) sum += expf(row[j] - mx);
        const float lse = mx + logf(sum);
        for (int j = 0; j < V; ++j) {
            const float y = row[j] - lse;
            a.logp_saved[((int64_t)b * T + tt) * V + j] = y;
            a.logp_out[((int64_t)b * T + tt) * V + j] = y;
        }
        if (a.row_stats) {
            const int64_t tgt = (tt + 1 < T) ? a.targets[(int64_t)b * T + tt + 1] : (int64_t)0;
            if (tgt != a.pad_tgt && tgt >= 0 && tgt < V) { nll_acc += lse - row[tgt]; cnt_acc += 1.f; }
        }
    }
    if (a.row_stats) {
        float *red = pre_s;
        nll_acc = wave_sum(nll_acc);
        cnt_acc = wave_sum(cnt_acc);
        if (lane == 0) { red[2 * wave] = nll_acc; red[2 * wave + 1] = cnt_acc; }
        __syncthreads();
        if (tid == 0) {
            float n0 = 0.f, n1 = 0.f;
            for (int i = 0; i < kAnyWaves; ++i) { n0 += red[2 * i]; n1 += red[2 * i + 1]; }
            a.row_stats[4 * b + 0] = n0;
            a.row_stats[4 * b + 1] = n1;
            a.row_stats[4 * b + 3] = 1.f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward through time: same outputs as decoder_bwd_kernel (decoder.hip) — per-step pre-activation gradients
// delta = [gate deltas (4H) | dzq (H)], dqt, dqv, the score-path key gradients, the energy-vector gradients per row,
// d h0 — and dS holds the head's part only: the LSTM-input and conditional-query parts of the context gradients are
// added by the dense product behind this kernel, as for the fast kernel.
// ------------------------------------------------------------------------------------------
struct AnyLdsB { int dh, dc, dl, dlp, dctx, dq, dqv, q, v1, v2, al, sc, datt, red, scr, total; };
__host__ __device__ inline AnyLdsB any_lds_bwd(int H, int L, int M) {
    const int HP = (H + 3) / 4 * 4, NM = ((L > M ? L : M) + 3) / 4 * 4;
    AnyLdsB o;
    int p = 0;
    o.dh = p; p += HP; o.dc = p; p += HP;
    o.dl = p; p += 5 * HP;             // delta (4H) | dzq (H; without the conditional query: d (projected visual query))
    o.dq = p; p += HP;                 // d (projected query) of the attention being processed — RIGHT BEHIND dl: [delta | dzq | dqt] is
                                       // the vector of the one-pass dh product over the concatenated weights (any_wcat)
    o.dlp = p; p += 4 * HP;            // delta unit-major (4 unit + gate): the order of the gate images' columns
    o.dctx = p; p += 2 * HP;           // d ctx_text | d ctx_vis: the head's part (the rest reaches d alpha through the gate images)
    o.dqv = p; p += HP;                // d (projected visual query), kept until the dh sum
    o.q = p; p += HP;                  // the saved projected query
    o.v1 = p; p += HP; o.v2 = p; p += HP;      // energy vectors
    o.al = p; p += NM; o.sc = p; p += NM; o.datt = p; p += NM;
    o.red = p; p += 64;
    o.scr = p; p += kAnyThreads;
    o.total = p;
    return o;
}

// one attention's backward at one step (row-local): d alpha_m = dctx . PK[m] (+ datt[m]) — or, dctx == NULL, already in
// sc[m] (the caller's, through the gate images) — softmax backward, then through
// v . tanh(q + PK[m]): dq, the score-path dPK (accumulated in global memory over the steps) and the energy-vector
// gradient.  Thread (k, p) owns feature k of the memories m = p, p + P, ... (P = kAnyThreads / CB thread groups per
// feature block, as in matvec_cols): its dPK elements are its own, its share of dq goes through `scratch`, its share
// of the energy-vector gradient stays in its register `dv` until the end of the kernel.  Requires H <= kAnyThreads.
__device__ __forceinline__ void attention_bwd_any(const float *dctx, const float *datt, const float *q_s, const float *v_s,
                                                  const float *al, const float *__restrict__ pk, float *dpk, int n, int H,
                                                  float *sc, float *dq_s, float *red, float *scratch, float &dv) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (dctx) {                                                        // d alpha_m = dctx . PK[m] (+ datt[m]), a wave per memory
        for (int m0 = 0; m0 < n; m0 += kAnyWaves) {
            const int m = m0 + wave, mc = min(m, n - 1);
            float p = 0.f;
            for (int k = lane; k < H; k += 64) p = fmaf(dctx[k], pk[(int64_t)mc * H + k], p);
            p = wave_sum(p);
            if (lane == 0 && m < n) sc[m] = p + (datt ? datt[m] : 0.f);
        }
        lds_barrier();
    }
    if (tid < 64) {                                                    // ds_m = alpha_m (dalpha_m - sum alpha dalpha)
        float s = 0.f;
        for (int m = tid; m < n; m += 64) s = fmaf(al[m], sc[m], s);
        s = wave_sum(s);
        if (tid == 0) red[0] = s;
    }
    lds_barrier();
    const float s = red[0];
    const int CB = min((H + 63) & ~63, kAnyThreads), P = kAnyThreads / CB;
    const int k = tid % CB, p = tid / CB;
    float dq = 0.f;
    if (p < P && k < H) {
        const float qk = q_s[k], vk = v_s[k];
        float dvk = 0.f;
        for (int m = p; m < n; m += P) {
            const float ds = al[m] * (sc[m] - s);
            const float th = tanhf_(qk + pk[(int64_t)m * H + k]);
            const float g = ds * vk * (1.f - th * th);
            dpk[(int64_t)m * H + k] += g;
            dq += g;
            dvk = fmaf(ds, th, dvk);
        }
        dv += dvk;
    }
    scratch[tid] = dq;
    lds_barrier();
    if (tid < CB && tid < H) {
        float sum = 0.f;
        for (int q = 0; q < P; ++q) sum += scratch[q * CB + tid];
        dq_s[tid] = sum;
    }
    lds_barrier();
}

template <bool V4>
__global__ __launch_bounds__(kAnyThreads) void decoder_bwd_any_kernel(DecoderArgs a, int H, int cond) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = a.T, L = a.L, M = a.M, V = a.V;
    const AnyLdsB o = any_lds_bwd(H, L, M);
    float *dh_s = smem + o.dh, *dc_s = smem + o.dc, *dl_s = smem + o.dl, *dlp_s = smem + o.dlp, *dctx_s = smem + o.dctx, *dq_s = smem + o.dq,
          *dqv_s = smem + o.dqv, *q_s = smem + o.q, *vt_s = smem + o.v1, *vv_s = smem + o.v2, *al_s = smem + o.al, *sc = smem + o.sc,
          *datt_s = smem + o.datt, *red = smem + o.red, *scr = smem + o.scr;
    int len = a.cmd_lengths[b];
    len = max(1, min(len, L));
    const float *pk_t = a.pk_t + (int64_t)b * L * H, *pk_v = a.pk_v + (int64_t)b * M * H;
    float *dpk_t = a.dpk_t + (int64_t)b * L * H, *dpk_v = a.dpk_v + (int64_t)b * M * H;
    const float *u_t = a.u_t + (int64_t)b * L * 4 * H, *u2_t = a.u2_t + (int64_t)b * L * H, *u_v = a.u_v + (int64_t)b * M * 4 * H;
    const float *W_hh = a.any_w_hh, *W_ih = a.any_w_ih, *W_qt = a.any_w_qt, *W_qv = a.any_w_qv, *W_q2k = a.any_w_q2k;
    const bool use_u = a.any_use_u != 0;
    const AnyResidency res = any_residency(H, L, M, cond != 0, use_u, true, o.total, a.any_lds_floats);
    float *stamp_s = red + 32;                      // diagnostic phase stamps (ANY_STAMP): slots 16.. of the stamp buffer
    long long stamp_prev = 0;
    if (tid < 16) stamp_s[tid] = 0.f;
    // transposed weight products: 16-byte loads along the rows when the hidden size is a multiple of 4 (anyshape.h)
    // — for TALL products only: the quad form pays eight barriers for its exchange, the lane-per-column form two, and a round
    // trip of the latter moves 16 KB (cheap); from ~20 dependent round trips on the quad form wins (W_hh^T from hidden 144 on,
    // everything at hidden 200+; measured: hidden 128 1.39 -> 1.63 ms with every product on it, 200 3.54 -> 3.23)
    auto wcols = [&](const float *W, int ldw, int c0, int R, int C, const float *x, auto store) {
        const int CB = min((C + 63) & ~63, kAnyThreads), trips = R / (4 * (kAnyThreads / CB));
        if (V4 && trips >= 20) matvec_cols4(W, ldw, c0, R, C, x, scr, store);
        else matvec_cols(W, ldw, c0, R, C, x, scr, store);
    };

    // ---- seeds and the head's backward (as decoder_bwd_kernel's prologue): dlogits, dS = Wc^T dlogits
    float aux_scale = (a.seeds && a.daux) ? a.seeds[1] : 1.f;
    float scl = a.seeds ? a.seeds[0] : 1.f;
    if (a.nll_mode) {
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
        for (int r = tid; r < a.B; r += kAnyThreads) {
            const float4 x = *reinterpret_cast<const float4 *>(a.row_stats + 4 * r);
            p0 += x.x; p1 += x.y; p2 += x.z;
        }
        p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
        if (lane == 0) { red[3 * wave] = p0; red[3 * wave + 1] = p1; red[3 * wave + 2] = p2; }
        __syncthreads();
        p0 = p1 = p2 = 0.f;
        for (int i = 0; i < kAnyWaves; ++i) { p0 += red[3 * i]; p1 += red[3 * i + 1]; p2 += red[3 * i + 2]; }
        scl = (a.nll_mode == 2) ? 1.f : 1.f / p1;
        aux_scale = a.aux_saved ? ((a.nll_mode == 2) ? a.w_aux : a.w_aux / (float)a.B) : 0.f;
        if (b == 0 && tid == 0) {
            a.stats_out[0] = p0; a.stats_out[1] = p1; a.stats_out[2] = p2; a.stats_out[3] = (float)a.B;
            a.seeds_out[0] = scl; a.seeds_out[1] = aux_scale;
            a.seeds_out[2] = p0 * scl + p2 * aux_scale;
        }
        __syncthreads();
    }
    for (int tt = tid; tt < T; tt += kAnyThreads) {
        const int64_t bt = (int64_t)b * T + tt;
        const float *y = a.logp_saved + bt * V;
        if (a.nll_mode) {
            const int64_t tgt = (tt + 1 < T) ? a.targets[(int64_t)b * T + tt + 1] : (int64_t)0;
            const bool live = tgt != a.pad_tgt && tgt >= 0 && tgt < V;
            for (int j = 0; j < V; ++j) a.dlogits[bt * V + j] = live ? scl * (expf(y[j]) - (j == tgt ? 1.f : 0.f)) : 0.f;
        } else {
            const float *dy = a.dlogp + bt * V;
            float sum = 0.f;
            for (int j = 0; j < V; ++j) sum += dy[j];
            for (int j = 0; j < V; ++j) a.dlogits[bt * V + j] = scl * (dy[j] - expf(y[j]) * sum);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < T * 4 * H; idx += kAnyThreads) {
        const int tt = idx / (4 * H), col = idx - tt * 4 * H;
        const float *dl = a.dlogits + ((int64_t)b * T + tt) * V;
        float acc = 0.f;
        for (int v = 0; v < V; ++v) acc = fmaf(dl[v], a.head_wc[(int64_t)v * 4 * H + col], acc);
        a.ds[((int64_t)b * T + tt) * 4 * H + col] = acc;
    }
    // auxiliary head backward: d att_sum = seed1 * (daux - exp(aux_logp) sum daux)  (model.py:205)
    {
        const bool has_aux = a.nll_mode ? (a.aux_saved != nullptr) : (a.daux != nullptr);
        const int64_t pos = a.positions ? a.positions[b] : (int64_t)-1;
        if (tid < 64) {
            float sum = 0.f;
            if (has_aux)
                for (int m = tid; m < M; m += 64) sum += a.nll_mode ? (m == pos ? -1.f : 0.f) : a.daux[(int64_t)b * M + m];
            sum = wave_sum(sum);
            if (tid == 0) red[0] = sum;
        }
        __syncthreads();
        const float sum = red[0];
        for (int m = tid; m < M; m += kAnyThreads) {
            float da = 0.f;
            if (has_aux) {
                const float dy = a.nll_mode ? (m == pos ? -1.f : 0.f) : a.daux[(int64_t)b * M + m];
                da = aux_scale * (dy - expf(a.aux_saved[(int64_t)b * M + m]) * sum);
            }
            datt_s[m] = da;
        }
    }
    for (int k = tid; k < H; k += kAnyThreads) { dh_s[k] = 0.f; dc_s[k] = 0.f; vt_s[k] = a.v_t[k]; vv_s[k] = a.v_v[k]; }
    {   // score-path key gradients: accumulated over the steps in LDS when they fit (written out once, below), else in global memory
        float *zv = (res.flags & kAnyMemDPK) ? smem + res.dpkv : dpk_v, *zt = (res.flags & kAnyMemDPK) ? smem + res.dpkt : dpk_t;
        for (int i = tid; i < M * H; i += kAnyThreads) zv[i] = 0.f;
        for (int i = tid; i < L * H; i += kAnyThreads) zt[i] = 0.f;
    }
    if (res.flags & kAnyMemPK) { any_stage(smem + res.pkt, pk_t, L * H); any_stage(smem + res.pkv, pk_v, M * H); }
    if (res.flags & kAnyMemUT) { any_stage(smem + res.ut, u_t, L * 4 * H); if (cond) any_stage(smem + res.u2t, u2_t, L * H); }
    if (res.flags & kAnyMemUV) any_stage(smem + res.uv, u_v, M * 4 * H);
    float dvt = 0.f, dvv = 0.f;                     // this thread's share of the energy-vector gradients (attention_bwd_any)
    __syncthreads();

    int Hl = H, Ll = L, Ml = M, lenl = len;
    for (int t = T - 1; t >= 0; --t) {
        LAUNDER(Hl); LAUNDER(Ll); LAUNDER(Ml); LAUNDER(lenl);
        const int H = Hl, L = Ll, M = Ml, len = lenl;
        const float *pk_t = (res.flags & kAnyMemPK) ? smem + res.pkt : a.pk_t + (int64_t)b * L * H;
        const float *pk_v = (res.flags & kAnyMemPK) ? smem + res.pkv : a.pk_v + (int64_t)b * M * H;
        float *dpk_t = (res.flags & kAnyMemDPK) ? smem + res.dpkt : a.dpk_t + (int64_t)b * L * H;
        float *dpk_v = (res.flags & kAnyMemDPK) ? smem + res.dpkv : a.dpk_v + (int64_t)b * M * H;
        const float *u_t = (res.flags & kAnyMemUT) ? smem + res.ut : a.u_t + (int64_t)b * L * 4 * H;
        const float *u2_t = (res.flags & kAnyMemUT) ? smem + res.u2t : a.u2_t + (int64_t)b * L * H;
        const float *u_v = (res.flags & kAnyMemUV) ? smem + res.uv : a.u_v + (int64_t)b * M * 4 * H;
        const int64_t bt = (int64_t)b * T + t;
        // ---- LSTM cell backward (dh_t = head part + what step t+1 passed back)
        if (t == T - 1 && a.stamps && blockIdx.x == 0 && tid == 0) stamp_prev = clock64();
        for (int u = tid; u < H; u += kAnyThreads) {
            const float dh = a.ds[bt * 4 * H + 3 * H + u] + dh_s[u];
            const float ig = a.gates[bt * 4 * H + u], fg = a.gates[bt * 4 * H + H + u], gg = a.gates[bt * 4 * H + 2 * H + u],
                        og = a.gates[bt * 4 * H + 3 * H + u];
            const float c = a.cells[bt * H + u];
            const float c_prev = t > 0 ? a.cells[(bt - 1) * H + u]
                                       : (a.c0 ? a.c0[(int64_t)b * H + u] : a.hprev[(int64_t)b * T * H + u]);   // c_0 = h_0 (model.py:195)
            const float tc = tanhf_(c);
            const float dct = dc_s[u] + dh * og * (1.f - tc * tc);
            const float di = dct * gg * ig * (1.f - ig), df = dct * c_prev * fg * (1.f - fg), dg = dct * ig * (1.f - gg * gg),
                        d_o = dh * tc * og * (1.f - og);
            dc_s[u] = dct * fg;
            dl_s[u] = di; dl_s[H + u] = df; dl_s[2 * H + u] = dg; dl_s[3 * H + u] = d_o;
            if (use_u) { dlp_s[4 * u] = di; dlp_s[4 * u + 1] = df; dlp_s[4 * u + 2] = dg; dlp_s[4 * u + 3] = d_o; }
            a.delta[bt * 5 * H + u] = di; a.delta[bt * 5 * H + H + u] = df;
            a.delta[bt * 5 * H + 2 * H + u] = dg; a.delta[bt * 5 * H + 3 * H + u] = d_o;
        }
        lds_barrier();
        ANY_STAMP(0)
        // ---- d alpha_vis[m] = delta . U_vis[m] + d ctx_vis(head) . PK_vis[m] + d att_sum[m]: the LSTM-input part of the context
        //      gradient reaches d alpha through the gate images (delta . U[m] = (W_ih[:, ctx]^T delta) . PK[m]); the context
        //      gradients themselves (the value path of the keys) are completed by the dS += product behind this kernel
        //      (narrow hidden sizes / many memories, use_u false: d ctx = head part + W_ih[:, ctx]^T delta formed here, streamed)
        if (use_u) for (int c = tid; c < 2 * H; c += kAnyThreads) dctx_s[c] = a.ds[bt * 4 * H + H + c];
        else wcols(W_ih, 3 * H, H, 4 * H, 2 * H, dl_s, [&](int c, float v) { dctx_s[c] = v + a.ds[bt * 4 * H + H + c]; });
        for (int m = tid; m < M; m += kAnyThreads) al_s[m] = a.alpha_s[bt * M + m];
        for (int k = tid; k < H; k += kAnyThreads) q_s[k] = a.qv[bt * H + k];
        lds_barrier();
        ANY_STAMP(1)
        if (use_u && V4) {          // one pass, a wave per cell (rows_few)
            rows_few(FewRows{u_v, 4 * H, 4 * H, dlp_s}, FewRows{pk_v, H, H, dctx_s + H}, FewRows{pk_v, H, 0, dctx_s}, M,
                     [&](int m, float v) { sc[m] = v + datt_s[m]; });
            lds_barrier();
        } else if (use_u) {
            matvec_rows<true>(u_v, 4 * H, M, 4 * H, dlp_s, [&](int m, float v) { sc[m] = v + datt_s[m]; });
            lds_barrier();
            matvec_rows<V4>(pk_v, H, M, H, dctx_s + H, [&](int m, float v) { sc[m] += v; });
            lds_barrier();
        }
        // ---- visual attention backward
        ANY_STAMP(2)
        attention_bwd_any(use_u ? nullptr : dctx_s + H, datt_s, q_s, vv_s, al_s, pk_v, dpk_v, M, H, sc, dq_s, red, scr, dvv);
        for (int k = tid; k < H; k += kAnyThreads) {
            const float v = dq_s[k];
            dqv_s[k] = v;
            a.dqv[bt * H + k] = v;
            if (!cond) dl_s[4 * H + k] = v;                             // the middle block of the dh product's vector (any_wcat)
        }
        lds_barrier();
        ANY_STAMP(3)
        if (cond) {
            // d q2 = W_qv^T dqv, through tanh; the conditional query's share of d ctx_text
            wcols(W_qv, H, 0, H, H, dqv_s, [&](int c, float v) {
                const float q = a.q2[bt * H + c];
                const float dz = v * (1.f - q * q);
                dl_s[4 * H + c] = dz;
                a.delta[bt * 5 * H + 4 * H + c] = dz;
            });
            if (!use_u) wcols(W_q2k, 2 * H, H, H, H, dl_s + 4 * H, [&](int c, float v) { dctx_s[c] += v; });
        }
        for (int m = tid; m < L; m += kAnyThreads) al_s[m] = a.alpha_c[bt * L + m];
        for (int k = tid; k < H; k += kAnyThreads) q_s[k] = a.qt[bt * H + k];
        lds_barrier();
        ANY_STAMP(4)
        // ---- d alpha_text[m] = delta . U_text[m] + dzq . U2_text[m] + d ctx_text(head) . PK_text[m]
        if (use_u && V4) {
            rows_few(FewRows{u_t, 4 * H, 4 * H, dlp_s}, FewRows{u2_t, H, cond ? H : 0, dl_s + 4 * H}, FewRows{pk_t, H, H, dctx_s}, len,
                     [&](int m, float v) { sc[m] = v; });
            lds_barrier();
        } else if (use_u) {
            matvec_rows<true>(u_t, 4 * H, len, 4 * H, dlp_s, [&](int m, float v) { sc[m] = v; });
            lds_barrier();
            if (cond) {
                matvec_rows<V4>(u2_t, H, len, H, dl_s + 4 * H, [&](int m, float v) { sc[m] += v; });
                lds_barrier();
            }
            matvec_rows<V4>(pk_t, H, len, H, dctx_s, [&](int m, float v) { sc[m] += v; });
            lds_barrier();
        }
        // ---- textual attention backward
        ANY_STAMP(5)
        attention_bwd_any(use_u ? nullptr : dctx_s, nullptr, q_s, vt_s, al_s, pk_t, dpk_t, len, H, sc, dq_s, red, scr, dvt);
        for (int k = tid; k < H; k += kAnyThreads) a.dqt[bt * H + k] = dq_s[k];
        ANY_STAMP(6)
        // ---- dh_{t-1} = W_hh^T delta + W_qt^T dqt + (W_q2k[:, :H]^T dzq  or  W_qv^T dqv)
        if (V4 && use_u) {  // one transposed product over the concatenated rows: x = [delta | dzq or dqv | dqt] = dl_s .. dq_s (adjacent)
            matvec_cols4(a.any_wcat, H, 0, 6 * H, H, dl_s, scr, [&](int c, float v) { dh_s[c] = v; });
            ANY_STAMP(7)
        } else if (V4) {    // one pass over the 6H concatenated rows (matvec_cols4_sets)
            matvec_cols4_sets(ColsRows{W_hh, H, 4 * H, dl_s}, ColsRows{W_qt, H, H, dq_s},
                              cond ? ColsRows{W_q2k, 2 * H, H, dl_s + 4 * H} : ColsRows{W_qv, H, H, dqv_s}, H, scr,
                              [&](int c, float v) { dh_s[c] = v; });
            ANY_STAMP(7)
        } else {
            wcols(W_hh, H, 0, 4 * H, H, dl_s, [&](int c, float v) { dh_s[c] = v; });
            lds_barrier();
            ANY_STAMP(7)
            wcols(W_qt, H, 0, H, H, dq_s, [&](int c, float v) { dh_s[c] += v; });
            lds_barrier();
            if (cond) wcols(W_q2k, 2 * H, 0, H, H, dl_s + 4 * H, [&](int c, float v) { dh_s[c] += v; });
            else wcols(W_qv, H, 0, H, H, dqv_s, [&](int c, float v) { dh_s[c] += v; });
            lds_barrier();
        }
        ANY_STAMP(8)
    }
    // ---- epilogue: initial-state gradient through the bridge tanh (h0 = c0 = tanh(.), model.py:195), energy vectors
    if (a.stamps && blockIdx.x == 0 && tid < 16) a.stamps[tid] = stamp_s[tid];
    for (int k = tid; k < H; k += kAnyThreads) {
        const float h0 = a.hprev[(int64_t)b * T * H + k];
        a.dh0[(int64_t)b * H + k] = (dh_s[k] + dc_s[k]) * (1.f - h0 * h0);
    }
    if (res.flags & kAnyMemDPK) {                      // the key gradients accumulated in LDS: written once (padded memories: zero)
        __syncthreads();
        for (int i = tid; i < M * H; i += kAnyThreads) dpk_v[i] = smem[res.dpkv + i];
        for (int i = tid; i < L * H; i += kAnyThreads) dpk_t[i] = i < len * H ? smem[res.dpkt + i] : 0.f;
    } else {
        for (int i = tid; i < (L - len) * H; i += kAnyThreads) dpk_t[(int64_t)len * H + i] = 0.f;
    }
    {   // energy-vector gradients of the row: the thread groups' shares of feature k
        const int CB = min((H + 63) & ~63, kAnyThreads), P = kAnyThreads / CB;
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
            scr[tid] = which ? dvv : dvt;
            __syncthreads();
            if (tid < H) {
                float sum = 0.f;
                for (int q = 0; q < P; ++q) sum += scr[q * CB + tid];
                (which ? a.dv_v : a.dv_t)[(int64_t)b * H + tid] = sum;
            }
        }
    }
}

// [W_hh ; mid ; W_query_text] (mid = W_q2k[:, :H], row stride 2H, or W_query_vis) as ONE [6H, H] matrix, written twice per step:
// row-major (`plain`: dh_{t-1} of a reverse step is one transposed product over its rows) and trip-major (`stream`,
// matvec_rows_image: the three products with h_{t-1} of a forward step are one stream of consecutive kilobytes).
__global__ void any_wcat_kernel(const float *__restrict__ w_hh, const float *__restrict__ w_mid, int ld_mid,
                                const float *__restrict__ w_qt, int H, float *__restrict__ plain, float *__restrict__ stream) {
    const int R = 6 * H, ncc = (H + 63) >> 6;
    auto at = [&](int r, int c) -> float {
        if (r >= R || c >= H) return 0.f;
        return r < 4 * H ? w_hh[(int64_t)r * H + c] : (r < 5 * H ? w_mid[(int64_t)(r - 4 * H) * ld_mid + c] : w_qt[(int64_t)(r - 5 * H) * H + c]);
    };
    const int64_t total = (int64_t)any_stream_image_floats(R, H);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3), tid = (int)((i >> 2) % kAnyThreads);
        const int tu = (int)((i >> 2) / kAnyThreads), u = tu & 3, t = tu >> 2;
        const int rb = t / ncc, cc = t - rb * ncc;
        stream[i] = at(rb * 256 + (tid >> 4) + u * 64, cc * 64 + 4 * (tid & 15) + e);
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R * H; i += gridDim.x * blockDim.x) plain[i] = at(i / H, i % H);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
constexpr int kAnyMaxHidden = kAnyThreads;         // the backward pass gives every feature of an attention a thread

// The streaming kernels read the gate images U = PK . W_ih[:, ctx]^T ((L + M) . 4H floats of the ROW's own data per step)
// in place of W_ih's context columns (8 H^2 floats every row's workgroup streams): a gain for wide hidden sizes with few
// memories — B = 256, T = 20, ms per step: H 160 2.93 -> 2.70, H 200 4.9 -> 3.8, H 256 6.0 -> 4.3 — and a loss or a draw
// otherwise (H 144 2.54 -> 2.56, H 128 1.57 -> 1.68; H 100 on a 12 x 12 grid 2.0 -> 2.2).  GSCAN_ANY_U=0/1 forces the choice (tests).
// Round 5: and whenever EVERY memory of the row, gate images included, fits LDS beside the kernels' vectors (hidden 128 on a
// 6 x 6 grid: 123 KB + 23 KB of key gradients in the backward kernel): the images are then read from LDS and what is left
// to stream is 7 H^2 floats instead of 16 H^2.
static bool any_all_resident(int H, int L, int M) {
    static const int off = [] { const char *e = getenv("GSCAN_ANY_RESIDENT"); return e ? atoi(e) == 0 : false; }();
    if (off) return false;
    const AnyResidency f = any_residency(H, L, M, true, true, false, any_lds_fwd(H, L, M).total, kAnyLdsFloats);
    const AnyResidency b = any_residency(H, L, M, true, true, true, any_lds_bwd(H, L, M).total, kAnyLdsFloats);
    return (f.flags & kAnyMemUV) && (b.flags & kAnyMemUV);
}
bool decoder_any_uses_gate_images(int H, int L, int M) {
    static const int forced = [] { const char *e = getenv("GSCAN_ANY_U"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return forced != 0;
    // (round 5: with the images' sums and dot products as single passes the form pays from FEW memories on, not only for wide
    // hidden sizes — ms per step at B = 256: hidden 128 1.44 -> 1.18, 144 2.13 -> 1.83, 160 2.24 -> 2.03, 200 3.42 -> 3.01; it still
    // loses where the memories outnumber the features: a 12 x 12 grid at hidden 100 1.55 -> 1.66, a 128-token command 4.61 -> 4.94)
    return L + M <= H || any_all_resident(H, L, M);
}

int decoder_run_any(bool backward, int B, int H, bool cond, const DecoderArgs &a_in, hipStream_t stream) {
    DecoderArgs a = a_in;
    a.any_use_u = decoder_any_uses_gate_images(H, a.L, a.M) ? 1 : 0;
    GSCAN_CHECK(!a.any_use_u || (a.u_t && a.u_v && (!cond || a.u2_t)), "decoder (any shape): gate images missing");
    GSCAN_CHECK(H >= 1 && H <= kAnyMaxHidden, "decoder: decoder_hidden_size %d is outside 1..%d", H, kAnyMaxHidden);
    GSCAN_CHECK(a.any_w_hh && a.any_w_ih && a.any_w_qt && a.any_w_qv && (!cond || a.any_w_q2k),
                "decoder (any shape): the parameter pointers are missing");
    const bool greedy = !backward && a.tokens_out != nullptr;
    const int base = backward ? any_lds_bwd(H, a.L, a.M).total : any_lds_fwd(H, a.L, a.M).total;
    GSCAN_CHECK((size_t)base * sizeof(float) <= 160 * 1024, "decoder (any shape): %zu bytes of LDS per row (hidden %d, %d + %d memories)",
                (size_t)base * sizeof(float), H, a.L, a.M);
    // the memories this launch keeps in LDS (GSCAN_ANY_RESIDENT=0: none, round 4's kernels); the kernel derives the same plan
    static const int resident = [] { const char *e = getenv("GSCAN_ANY_RESIDENT"); return e ? atoi(e) : 1; }();
    a.any_lds_floats = resident ? kAnyLdsFloats : 0;
    const AnyResidency res = any_residency(H, a.L, a.M, cond, a.any_use_u != 0, backward, base, a.any_lds_floats);
    const size_t bytes = (size_t)(((base + 3) & ~3) + res.floats) * sizeof(float);
    if (greedy) {
        GSCAN_CHECK(a.head_wc && a.dec_emb && a.steps_out, "greedy decoder: missing tables");
        GSCAN_CHECK(a.V <= 4 * H, "greedy decoder (any shape): a vocabulary of %d needs decoder_hidden_size >= %d", a.V, (a.V + 3) / 4);
    }
    const bool v4 = H % 4 == 0;
    const double macs = (double)H * H + 2.0 * a.L * H + (cond ? 2.0 * H * H : 0.0) + (double)H * H + 2.0 * a.M * H +
                        4.0 * H * 3.0 * H + 4.0 * H * H + (double)H * a.V;
    ProbeScope probe(backward ? P_DECODER_BWD : P_DECODER_FWD, stream, 2.0 * macs * B * a.T);
    auto launch = [&](auto kernel, const char *name) -> int {
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
        hipLaunchKernelGGL(kernel, dim3(B), dim3(kAnyThreads), bytes, stream, a, H, cond ? 1 : 0);
        GSCAN_LAUNCHED(name);
        return 0;
    };
    if (!backward && v4 && a.any_use_u) {       // [W_hh ; W_q2k[:, :H] or W_query_vis ; W_query_text] for this step's two launches
        GSCAN_CHECK(a.any_wcat != nullptr, "decoder (any shape): no room for the concatenated weights");
        GSCAN_CHECK(a.any_wcat_stream != nullptr, "decoder (any shape): no room for the streamed weights");
        hipLaunchKernelGGL(any_wcat_kernel, dim3((int)std::min<int64_t>(cdiv((int64_t)any_stream_image_floats(6 * H, H), 256), 4096)),
                           dim3(256), 0, stream, a.any_w_hh, cond ? a.any_w_q2k : a.any_w_qv, cond ? 2 * H : H, a.any_w_qt, H,
                           a.any_wcat, a.any_wcat_stream);
        GSCAN_LAUNCHED("any_wcat_kernel");
    }
    if (backward) return v4 ? launch(decoder_bwd_any_kernel<true>, "decoder_bwd_any_kernel") : launch(decoder_bwd_any_kernel<false>, "decoder_bwd_any_kernel");
    if (greedy) return v4 ? launch(decoder_fwd_any_kernel<true, true>, "decoder_fwd_any_kernel") : launch(decoder_fwd_any_kernel<false, true>, "decoder_fwd_any_kernel");
    return v4 ? launch(decoder_fwd_any_kernel<true, false>, "decoder_fwd_any_kernel") : launch(decoder_fwd_any_kernel<false, false>, "decoder_fwd_any_kernel");
}

GSCAN_TRACE_TU(decoder_any)

}  // namespace gscan
