// Joint-attention LSTM decoder over all T steps (seq2seq/seq2seq_model.py:359-492), forward
// and backward, as two persistent kernels: ONE WORKGROUP PER BATCH ROW, time loop inside.
//
// Why this shape on MI355X.  The decoder is a chain of T dependent steps per row, and rows
// never interact, so the latency of one step — not FLOPs — bounds the training step
// (SURVEY.md §7 hard part 1).  With a row per workgroup, 256 rows fill the 256 CUs and a
// step needs no inter-workgroup traffic at all:
//   * every matrix that multiplies the recurrent state ( W_hh | W_query_text | W_q2k[:, :H]
//     and W_query_vis: 7*H*H floats = 280 KB at H=100) stays in VGPRs for all T steps — 2 MB of
//     registers per CU is the largest on-chip store, LDS (160 KB) could not hold them.  A weight
//     row is split over a LANE PAIR (each lane holds half of it, up to three rows per pair), so
//     the workgroup is 8 waves = 2 per SIMD with a 256-VGPR budget: room to keep a dozen LDS
//     reads in flight, which is what the 100-deep dot products need.  The two halves are added
//     with one DPP quad_perm, not through LDS;
//   * the row's memories stay in LDS for all T steps: projected keys PK_text [L,H] and
//     PK_vis [G*G,H], plus U = PK . W_ih[:, ctx]^T ([L,4H] and [G*G,4H]).  Because an attention
//     context is a convex combination of projected keys, W_ih[:,ctx] . ctx = sum_m alpha_m U[m]:
//     the context part of the LSTM input GEMM becomes an M-term LDS reduction with no weights
//     at all (U comes from one dense MFMA GEMM before the loop).  The embedding part of the
//     LSTM input is known for all t (teacher forcing) and is also a GEMM before the loop;
//   * tanh(q + PK) score tiles are recomputed in backward instead of being saved
//     (46*H floats per step per row would make the path HBM-bound).
// The output head does not feed back: it runs once for the row's T steps as the epilogue of the forward kernel
// and its backward as the prologue of the backward kernel (no launches, no HBM round trip in between).
//
// Weight rows ("tasks") r = slot*256 + pair, H = hidden size:
//   forward   [0,4H) W_hh[r,:] -> gate r      [4H,5H) W_query_text[k,:]      [5H,6H) W_q2k[k,:H] (conditional)
//             or W_query_vis[k,:];            [6H,7H) W_query_vis[k,:] applied to the conditional query
//   backward  the transposes: task (seg,k) holds column k of block seg of [W_hh (4 blocks) | W_query_text |
//             W_q2k[:, :H] or W_query_vis] and of W_query_vis; dh_{t-1}[k] = sum over the six blocks.
// The step prologue kernel writes both register images once per step ([slot][i][thread], coalesced; step.h).
#include "step.h"

#ifndef GSCAN_DEC_PART
#define GSCAN_DEC_PART 0      // which quarter of the hidden sizes this translation unit instantiates (step.h)
#endif

namespace gscan {

// Diagnostic phase stamps, compiled in with -DGSCAN_DEC_STAMPS only (tools/decoder_stamps.py runs on such a variant, tools/
// variants.py): thread 0 of workgroup 0 adds the cycles since the previous stamp to slot i.  Shares, not absolute times,
// are what to read from them.  Until round 5 the stamps were a run-time test of DecoderArgs::stamps in the shipped kernels:
// eight exec-mask branches per step, two SGPRs for the condition and two VGPRs for the previous clock value, in kernels
// that spill scalar registers.
#ifdef GSCAN_DEC_STAMPS
#define GSCAN_STAMP(i)                                                         \
    if (a.stamps && blockIdx.x == 0 && tid == 0) {                             \
        const long long now_ = clock64();                                      \
        stamp_acc[i] += (float)(now_ - stamp_prev);                            \
        stamp_prev = now_;                                                     \
    }
// Once-per-launch stamps (prologue / epilogue pieces) go straight to slots 10..15 of the stamp buffer.
#define GSCAN_STAMP_ONCE(i)                                                    \
    if (a.stamps && blockIdx.x == 0 && tid == 0) {                             \
        const long long now_ = clock64();                                      \
        a.stamps[i] = (float)(now_ - stamp_prev);                              \
        stamp_prev = now_;                                                     \
    }
#define GSCAN_STAMPS_ON 1
#else
#define GSCAN_STAMP(i)
#define GSCAN_STAMP_ONCE(i)
#define GSCAN_STAMPS_ON 0
#endif

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Returns x behind a compiler barrier.  Inside the time loops it keeps per-thread addresses that are cheap to
// recompute from being hoisted out of the loop, where they would be spilled to scratch and reloaded (with a full
// vmcnt drain) in every phase.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// The two lanes of a pair (decoder_pair_of, step.h: lane s and lane 7-s of a group of eight) exchange on the DPP datapath
// (row_half_mirror); both lanes must be active and both end up with the sum
__device__ __forceinline__ float pair_sum(float v) { return v + dpp_move<0x141, 0xf>(v); }

// Dot products of register-resident half rows with an LDS vector (K0 % 4 == 0), as packed FMAs
// (v_pk_fma_f32).  The vector is read in chunks of four float4; the next chunk's reads are issued before the
// current chunk's FMAs, so the LDS latency is paid about once per dot instead of once per read.
constexpr int kWaitVmcnt0 = 0x0F70;   // s_waitcnt vmcnt(0) only (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)
#ifndef GSCAN_DEC_DOT_CHUNK
#define GSCAN_DEC_DOT_CHUNK 4
#endif
constexpr int kDotChunk = GSCAN_DEC_DOT_CHUNK;
#ifndef GSCAN_DEC_BWD_CHUNK
#define GSCAN_DEC_BWD_CHUNK 4
#endif
constexpr int kBwdDotChunk = GSCAN_DEC_BWD_CHUNK;   // the backward dots are single-row: fewer reads in flight, fewer registers

template <int K0, int CH>
__device__ __forceinline__ void load_chunk(float4 (&x)[CH], const float4 *v4, int c) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
        if (c * CH + j < K0 / 4) x[j] = v4[c * CH + j];
}

// NS dots that share the same vector: out[s] = sum_i w[s][i] * v[i]
template <int NS, int K0, int CH = kDotChunk>
__device__ __forceinline__ void shared_dots(const float (&w)[NS][K0], const float *v, float (&out)[NS]) {
    constexpr int NQ = K0 / 4, NC = (NQ + CH - 1) / CH;
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    f32x2 a01[NS], a23[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { a01[s] = f32x2{0.f, 0.f}; a23[s] = f32x2{0.f, 0.f}; }
    float4 xa[CH], xb[CH];
    load_chunk<K0, CH>(xa, v4, 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float4 (&cur)[CH] = (c & 1) ? xb : xa;
        float4 (&nxt)[CH] = (c & 1) ? xa : xb;
        if (c + 1 < NC) load_chunk<K0, CH>(nxt, v4, c + 1);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int q = c * CH + j;
            if (q < NQ) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    a01[s] += f32x2{w[s][4 * q + 0], w[s][4 * q + 1]} * f32x2{cur[j].x, cur[j].y};
                    a23[s] += f32x2{w[s][4 * q + 2], w[s][4 * q + 3]} * f32x2{cur[j].z, cur[j].w};
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) out[s] = (a01[s].x + a01[s].y) + (a23[s].x + a23[s].y);
}

template <int K0, int CH = kDotChunk>
__device__ __forceinline__ float half_dot(const float (&w)[K0], const float *v) {
    const float(&w1)[1][K0] = reinterpret_cast<const float(&)[1][K0]>(w);
    float out[1];
    shared_dots<1, K0, CH>(w1, v, out);
    return out[0];
}

// Additive-attention scores s_m = v . tanh(q + PK_m) for m < n.  Wave w takes m = w, w+nwave, ... in balanced rounds
// of G <= 5 memories; a round is ONE straight-line block (G is a template parameter, reads of memories past the end
// are clamped to the last one and only the final write is guarded), so the scheduler interleaves the G tanh chains
// and the G DPP reductions instead of running them one after the other between branches — a lone reduction is a
// ~140-cycle dependent chain, and the 36 cells of a 6x6 grid are one round of five.  A lane owns the feature
// indices lane and lane+64; v and q for them are read once.
#ifndef GSCAN_DEC_SCORE_GROUP
#define GSCAN_DEC_SCORE_GROUP 5
#endif
constexpr int kScoreGroup = GSCAN_DEC_SCORE_GROUP;
#ifndef GSCAN_DEC_SCORE_PAIRS
#define GSCAN_DEC_SCORE_PAIRS 1   // 1 (round 6): a lane owns the ADJACENT features (2 lane, 2 lane + 1): one 8-byte LDS read per memory, v
#endif                            // and q one each (they were features lane and lane + 64: two 4-byte reads each); 0: round 5's layout
template <int H, int G>
__device__ __forceinline__ void score_round(float v1, float v2, float q1, float q2, int k1, int k2, const float *pk,
                                            int n, float *sc_s, int m0, int nwave, int lane) {
    float x1[G], x2[G], p[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int m = min(m0 + i * nwave, n - 1);                  // scalar arithmetic (m0 is an SGPR)
#if GSCAN_DEC_SCORE_PAIRS
        const f32x2 x = *reinterpret_cast<const f32x2 *>(pk + m * H + k1);
        x1[i] = x.x; x2[i] = x.y;
#else
        x1[i] = pk[m * H + k1];
        x2[i] = pk[m * H + k2];
#endif
    }
#pragma unroll
    for (int i = 0; i < G; ++i) p[i] = fmaf(v1, tanhf_(q1 + x1[i]), v2 * tanhf_(q2 + x2[i]));
    wave_sum_n<G>(p);
#pragma unroll
    for (int i = 0; i < G; ++i)
        if (lane == 0 && m0 + i * nwave < n) sc_s[m0 + i * nwave] = p[i];
}

// GSCAN_DEC_RFORM=1 (round 6): the FORWARD score terms in terms of r = 1 / (1 + e^{-2 (q + PK)}) instead of tanh = 2 r - 1 — the
// same two transcendentals, two plain instructions fewer per term (five for seven):
//   * the queries arrive PRE-SCALED, qs = -2 log2(e) q, written beside q by the lane that produces it, so the exponent is ONE
//     fma(PK, -2 log2 e, qs) instead of an add and a multiply;
//   * v . tanh(.) = sum_k 2 v_k r_k - sum_k v_k, and the constant is dropped (the softmax over the memories does not see it;
//     its shift bound becomes 2 sum_k max(v_k, 0)).
// Exact rewrites (saturation as before: 2^(+big) = inf -> r = 0, 2^(-big) = 0 -> r = 1).
// (The round's other attempt, TABLES of e^{-2 PK} in LDS against e^{-2 q} — one transcendental per term, pair -3.5 us — is not
// shipped: a factorised form is wrong when a query and a key are both huge and cancel, and every way of handing such rows to
// the plain form cost the gain or did not run: LABBOOK 0, profiles/r06_decoder_score_tables_ab.txt.)
#ifndef GSCAN_DEC_RFORM
#define GSCAN_DEC_RFORM 1
#endif
constexpr float kM2Log2e = -2.8853900817779268f;                // e^{-2x} = 2^(kM2Log2e x)
__device__ __forceinline__ float r_of(float pk, float qs) {     // 1 / (1 + e^{-2 (q + pk)}), qs = kM2Log2e q
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(pk, kM2Log2e, qs)));
}

// GSCAN_DEC_SCORE_HALF=1 (round 6 A/B): a HALF wave per memory — a lane owns FOUR adjacent features (one 16-byte read of
// the key, of v and of q), a wave scores two memories per round and ONE five-step DPP sum serves both (wave-per-memory:
// six steps + a readlane per memory, five of them interleaved).  R rounds straight-line (all reads in flight first).
#ifndef GSCAN_DEC_SCORE_HALF
#define GSCAN_DEC_SCORE_HALF 1
#endif
template <int H, int R>
__device__ __forceinline__ void score_rounds_half(const float4 &v4, const float4 &q4, int k, const float *pk, int n,
                                                  float *sc_s, int m0, int nwave, int l32, int hf) {
    float4 x[R];
    float p[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int mc = min(m0 + 2 * nwave * r + hf, n - 1);
        x[r] = *reinterpret_cast<const float4 *>(pk + __mul24(mc, H) + k);
    }
#if GSCAN_DEC_RFORM       // (here v4 = 2 v and q4 = the pre-scaled queries)
#pragma unroll
    for (int r = 0; r < R; ++r)
        p[r] = fmaf(v4.x, r_of(x[r].x, q4.x), v4.y * r_of(x[r].y, q4.y)) + fmaf(v4.z, r_of(x[r].z, q4.z), v4.w * r_of(x[r].w, q4.w));
#else
#pragma unroll
    for (int r = 0; r < R; ++r)
        p[r] = fmaf(v4.x, tanhf_(q4.x + x[r].x), v4.y * tanhf_(q4.y + x[r].y)) +
               fmaf(v4.z, tanhf_(q4.z + x[r].z), v4.w * tanhf_(q4.w + x[r].w));
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) p[r] += dpp_move<0xb1, 0xf>(p[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) p[r] += dpp_move<0x4e, 0xf>(p[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) p[r] += dpp_move<0x124, 0xf>(p[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) p[r] += dpp_move<0x128, 0xf>(p[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) p[r] += dpp_move<0x142, 0xa>(p[r]);     // lanes 16-31 / 48-63: the half's sum
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = m0 + 2 * nwave * r + hf;
        if (l32 == 31 && m < n) sc_s[m] = p[r];
    }
}

template <int H>
__device__ __forceinline__ void attention_scores(const float *v_s, const float *q_s, const float *pk, int n,
                                                 float *sc_s, int wave_v, int nwave, int lane, const float *qs_s = nullptr) {
    static_assert(H <= 128, "two feature indices per lane");
    const int wave = __builtin_amdgcn_readfirstlane(wave_v);
#if GSCAN_DEC_SCORE_HALF
    {
        static_assert(H % 4 == 0, "feature quads");
        const int l32 = lane & 31, hf = lane >> 5;
        const bool has = 4 * l32 < H;
        const int k = has ? 4 * l32 : 0;
        float4 v4 = *reinterpret_cast<const float4 *>(v_s + k);
        const float4 q4 = *reinterpret_cast<const float4 *>((GSCAN_DEC_RFORM ? qs_s : q_s) + k);
        if (!has) v4 = float4{0.f, 0.f, 0.f, 0.f};
        if (GSCAN_DEC_RFORM) v4 = float4{2.f * v4.x, 2.f * v4.y, 2.f * v4.z, 2.f * v4.w};
#ifndef GSCAN_DEC_SCORE_OWN_ROUNDS
#define GSCAN_DEC_SCORE_OWN_ROUNDS 1    // 1: a wave runs ITS rounds (36 cells = 18 pairs: waves 0-1 three, the others two; a command
#endif                                  // of ten: waves 5-7 none) instead of the busiest wave's, clamped - the SIMD's other wave issues
        const int pairs = (n + 1) >> 1;
        const int rounds = GSCAN_DEC_SCORE_OWN_ROUNDS ? (pairs - wave + nwave - 1) / nwave      // scalar (wave is an SGPR)
                                                      : (pairs + nwave - 1) / nwave;
        int m0 = 2 * wave;
        for (int left = rounds; left > 0;) {
            if (left >= 3) { score_rounds_half<H, 3>(v4, q4, k, pk, n, sc_s, m0, nwave, l32, hf); left -= 3; m0 += 6 * nwave; }
            else if (left == 2) { score_rounds_half<H, 2>(v4, q4, k, pk, n, sc_s, m0, nwave, l32, hf); left = 0; }
            else { score_rounds_half<H, 1>(v4, q4, k, pk, n, sc_s, m0, nwave, l32, hf); left = 0; }
        }
        return;
    }
#endif
#if GSCAN_DEC_SCORE_PAIRS
    static_assert(H % 2 == 0, "feature pairs");
    const bool has = 2 * lane < H;
    const int k1 = has ? 2 * lane : 0, k2 = k1 + 1;
    const f32x2 vv = *reinterpret_cast<const f32x2 *>(v_s + k1), qq = *reinterpret_cast<const f32x2 *>(q_s + k1);
    const float v1 = has ? vv.x : 0.f, v2 = has ? vv.y : 0.f;
    const float q1 = qq.x, q2 = qq.y;
#else
    const bool has2 = lane + 64 < H, has1 = lane < H;
    const int k1 = has1 ? lane : 0, k2 = has2 ? lane + 64 : 0;
    const float v1 = has1 ? v_s[k1] : 0.f, v2 = has2 ? v_s[k2] : 0.f;
    const float q1 = q_s[k1], q2 = q_s[k2];
#endif
    const int per = (n + nwave - 1) / nwave;                        // memories of the busiest wave
    const int rounds = (per + kScoreGroup - 1) / kScoreGroup;
    const int grp = (per + rounds - 1) / rounds;                    // memories per wave per round
    for (int m0 = wave; m0 < n; m0 += grp * nwave) {
        switch (grp) {
            case 1: score_round<H, 1>(v1, v2, q1, q2, k1, k2, pk, n, sc_s, m0, nwave, lane); break;
            case 2: score_round<H, 2>(v1, v2, q1, q2, k1, k2, pk, n, sc_s, m0, nwave, lane); break;
            case 3: score_round<H, 3>(v1, v2, q1, q2, k1, k2, pk, n, sc_s, m0, nwave, lane); break;
            case 4: score_round<H, 4>(v1, v2, q1, q2, k1, k2, pk, n, sc_s, m0, nwave, lane); break;
            default: score_round<H, 5>(v1, v2, q1, q2, k1, k2, pk, n, sc_s, m0, nwave, lane); break;
        }
    }
}

// ------------------------------------------------------------------------------------------
// geometry shared by the kernels, the weight-image kernel and the host
// ------------------------------------------------------------------------------------------
#if GSCAN_DEC_PART == 0
DecoderGeometry decoder_geometry(int H, bool cond) {
    DecoderGeometry g;
    g.rows = (cond ? 7 : 6) * H;
    g.slots = (g.rows + kDecPairs - 1) / kDecPairs;
    g.k0 = ((H / 2 + 3) / 4) * 4;
    g.image_floats = (int64_t)g.slots * g.k0 * kDecThreads;
    return g;
}
#endif

constexpr int kHeadWcRows = 16;        // vocabulary rows of the head's LDS image (one MFMA tile of logits)
constexpr int kHeadDlStride = 20;      // backward: row stride of the chunk's [32][16] dlogits tile (sixteen rows -> eight bank groups)
__host__ __device__ inline int head_fwd_scratch_floats(int H) {
    const int SS = 4 * H + 4, sch = kHeadChunk * SS > 2 * 8 * 256 ? kHeadChunk * SS : 2 * 8 * 256;
    return sch + kHeadWcRows * SS + kHeadChunk * 16 + kHeadChunk;
}
struct DecoderLds {
    int uv, pkv, ut, pkt, u2t, dpkv, dpkt, vec, total;
};
// uv_in_lds = false: the gate images of the visual memories U_vis [M,4H] (the largest resident block: 102 KB for an
// 8x8 grid at H = 100) stay in global memory and the two phases that read them stream them from L2 every step.
__host__ __device__ inline DecoderLds decoder_lds(int H, int L, int M, int V, bool cond, bool backward,
                                                  bool uv_in_lds = true) {
    const int HP = 2 * (((H / 2 + 3) / 4) * 4);       // padded length of every vector a half_dot reads
    DecoderLds o;
    int p = 0;
    o.uv = p;  p += uv_in_lds ? M * 4 * H : 0;
    o.pkv = p; p += M * H;
    o.ut = p;  p += L * 4 * H;
    o.pkt = p; p += L * H;
    o.u2t = p; p += cond ? L * H : 0;
    o.dpkv = p; p += backward ? M * H : 0;
    o.dpkt = p; p += backward ? L * H : 0;
    o.vec = p;
    p += (backward ? 7 * HP + 19 * H : 2 * HP + 8 * H) + 256;
    // scratch of the fused output head, overlaid on the memories (forward: after the loop; backward: before staging)
    const int head = backward ? kHeadChunk * kHeadDlStride + kHeadWcRows * 4 * H + 32 : head_fwd_scratch_floats(H);
    o.total = p > head ? p : head;
    return o;
}

// float4 helpers
__device__ __forceinline__ float4 fma4(float a, const float4 &x, const float4 &acc) {
    return float4{fmaf(a, x.x, acc.x), fmaf(a, x.y, acc.y), fmaf(a, x.z, acc.z), fmaf(a, x.w, acc.w)};
}
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b, float acc) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, fmaf(a.w, b.w, acc))));
}

// Column quad q (columns 4q..4q+3) of the row-memory images of one attention, [U (4H cols) | PK (H) | U2 (H)]:
// LDS float offset of memory 0 and the stride between memories.
__device__ __forceinline__ void quad_offset(int u_off, int pk_off, int u2_off, int H, int q, int &off, int &stride) {
    const int col = 4 * q;
    if (col < 4 * H) { off = u_off + col; stride = 4 * H; }
    else if (col < 5 * H) { off = pk_off + (col - 4 * H); stride = H; }
    else { off = u2_off + (col - 5 * H); stride = H; }
}

// global -> LDS copy of up to five arrays (each n % 4 == 0 floats, 16-byte aligned on both sides) as ONE index space:
// a thread's 16-byte loads are issued eight at a time whatever array they fall into, so the row's memories arrive in
// two or three round trips to L2 instead of one or two per array (five arrays copied one after the other, four loads
// in flight each: ~9 dependent round trips at the benchmark shape, most of the kernels' 8 us of set-up).  The source
// pointers are cast to the global address space: picked from a table they would be generic, and FLAT loads count
// on lgkmcnt too, which ties every LDS store's wait to all outstanding loads.
struct StageList {       // plain scalars: arrays indexed by a per-lane segment number would live in scratch memory
    const float *p0, *p1, *p2, *p3, *p4;
    int d0, d1, d2, d3, d4;     // LDS float offsets
    int e0, e1, e2, e3, e4;     // cumulative sizes in 16-byte units
};
using fvec4 = __attribute__((ext_vector_type(4))) float;
// Round 5: the copies go from global memory STRAIGHT to LDS (global_load_lds_dwordx4, gfx950): no staging registers —
// the register image of the weights (156 VGPRs) is in flight at the same time — and no wait per batch of loads: a thread
// issues all of its loads (14 at the benchmark shape) in one go, behind whatever the kernel requested before, and the
// caller waits ONCE (s_waitcnt vmcnt(0), then the workgroup barrier) before anyone reads the memories.  A wave
// instruction writes 64 consecutive 16-byte units from a wave-uniform LDS base (M0), so each array is walked in chunks of
// 64 units, wave w taking the chunks w, w + 8, ...; lanes past the array's end are masked off.
__device__ __forceinline__ void stage_array(float *smem, const float *src, int dst, int units, int tid) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int c = wave; c * 64 < units; c += kDecThreads / 64) {
        const int u = c * 64 + lane;
        if (u < units)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * (int64_t)u),
                                             (__attribute__((address_space(3))) void *)(smem + dst + 256 * c), 16, 0, 0);
    }
}
__device__ __forceinline__ void stage_all(float *smem, const StageList &sl, int tid) {
    stage_array(smem, sl.p0, sl.d0, sl.e0, tid);
    stage_array(smem, sl.p1, sl.d1, sl.e1 - sl.e0, tid);
    stage_array(smem, sl.p2, sl.d2, sl.e2 - sl.e1, tid);
    stage_array(smem, sl.p3, sl.d3, sl.e3 - sl.e2, tid);
    stage_array(smem, sl.p4, sl.d4, sl.e4 - sl.e3, tid);
}
// every load a wave issued before this point has landed (registers and LDS alike), and every wave's has: what the copies
// above wrote may be read
__device__ __forceinline__ void staged_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ StageList stage_list(const DecoderArgs &a, const DecoderLds &o, int b, int H, int L, int M,
                                                bool cond, bool uv_lds) {
    StageList sl;
    sl.p0 = a.u_v + (int64_t)b * M * 4 * H; sl.d0 = o.uv;  sl.e0 = uv_lds ? M * H : 0;            // 16-byte units
    sl.p1 = a.pk_v + (int64_t)b * M * H;    sl.d1 = o.pkv; sl.e1 = sl.e0 + M * H / 4;
    sl.p2 = a.u_t + (int64_t)b * L * 4 * H; sl.d2 = o.ut;  sl.e2 = sl.e1 + L * H;
    sl.p3 = a.pk_t + (int64_t)b * L * H;    sl.d3 = o.pkt; sl.e3 = sl.e2 + L * H / 4;
    sl.p4 = cond ? a.u2_t + (int64_t)b * L * H : sl.p3; sl.d4 = o.u2t; sl.e4 = sl.e3 + (cond ? L * H / 4 : 0);
    return sl;
}

// Role of thread tid in the column-sum phases.  A group of eight lanes 8g..8g+7 holds the gate rows of two hidden
// units: its lower quad (lanes 8g+j) unit g, its upper quad (lanes 8g+4+j) unit 64+g, lane j of a quad = gate j
// (i, f, g, o).  The dot phase pairs lane s with lane 7-s of the same eight (DPP row_half_mirror) and the weight
// image gives pair (g, j) the W_hh row of (unit g, gate j) in slot 0 and of (unit 64+g, gate 3-j) in slot 1 — so
// after the pair sum every lane already holds the W_hh.h term of ITS (unit, gate): slot 0 in the lower quad, slot
// 1 in the upper one (decoder_image_element, step.h).
struct GateLane { int unit, gate; bool valid, upper; };
template <int H>
__device__ __forceinline__ GateLane gate_lane(int tid) {
    GateLane r;
    r.upper = (tid >> 2) & 1;
    r.gate = tid & 3;
    r.unit = (r.upper ? 64 : 0) + (tid >> 3);
    r.valid = r.unit < H;
    return r;
}

// quad (four adjacent lanes) butterfly sum: every lane of the quad ends up with the total
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_move<0xb1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_move<0x4e, 0xf>(v);        // quad_perm [2,3,0,1]
    return v;
}
template <int I>
__device__ __forceinline__ float quad_bcast(float v) { return dpp_move<I * 0x55, 0xf>(v); }   // lane I of the quad

// sum_m alpha_m * X[m][4c .. 4c+3] for ONE column quad c of a row-memory image X, by the four lanes of a quad: lane j
// takes the memories m = j, j+4, ... (16-byte LDS reads, three in flight), the quad adds up on the DPP datapath and
// lane j returns column 4c + j.  alpha_m lives in lane m of `alpha` (every wave holds the whole distribution) and is
// fetched with ds_bpermute, so the loop runs the same number of rounds in every lane; threads without a column quad
// pass stride = 0 and ignore the result.  n is uniform in the workgroup.
#ifndef GSCAN_DEC_QCS_U
#define GSCAN_DEC_QCS_U 3
#endif
// The same sum when n is a whole number of rounds (n % (4 U) == 0: the 36 cells of a 6 x 6 grid are three rounds of
// twelve): no clamped indices, no masked weights, addresses by addition — per memory one LDS read, one ds_bpermute and two
// packed FMAs instead of those plus a min, a 24-bit multiply, two shifts, a compare and a select (round 6: the column-sum
// phases are bound by instruction issue, two waves per SIMD).
#ifndef GSCAN_DEC_QCS_EXACT
#define GSCAN_DEC_QCS_EXACT 1
#endif
__device__ __forceinline__ float quad_column_sum_exact(const float *base, int stride, int n, int j, float alpha) {
    constexpr int U = GSCAN_DEC_QCS_U;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float *ptr = base + j * stride;
    const int step = 4 * stride;
    int from = 4 * j;                                            // byte address of lane j + 4 u + i0 for ds_bpermute
    for (int i0 = 0; i0 < n; i0 += 4 * U) {
        f32x4 x[U];
        float am[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = *reinterpret_cast<const f32x4 *>(ptr + u * step);
            am[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(from + 16 * u, __float_as_int(alpha)));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += am[u] * x[u];
        ptr += U * step;
        from += 16 * U;
    }
    const float s0 = quad_sum(acc[0]), s1 = quad_sum(acc[1]), s2 = quad_sum(acc[2]), s3 = quad_sum(acc[3]);
    return j == 0 ? s0 : j == 1 ? s1 : j == 2 ? s2 : s3;
}

template <bool GLOBAL = false>
__device__ __forceinline__ float quad_column_sum(const float *base, int stride, int n, int j, float alpha) {
    constexpr int U = GSCAN_DEC_QCS_U;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < n; i0 += 4 * U) {
        f32x4 x[U];
        float am[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = i0 + 4 * u + j, mc = min(m, n - 1);
            if (GLOBAL) x[u] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(
                               (const __attribute__((address_space(1))) float *)base + (int64_t)mc * stride);
            else x[u] = *reinterpret_cast<const f32x4 *>(base + __mul24(mc, stride));   // 24-bit multiply: full rate (v_mul_lo_u32 is quarter rate)
            const float al = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * mc, __float_as_int(alpha)));
            am[u] = m < n ? al : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += am[u] * x[u];
    }
    const float s0 = quad_sum(acc[0]), s1 = quad_sum(acc[1]), s2 = quad_sum(acc[2]), s3 = quad_sum(acc[3]);
    return j == 0 ? s0 : j == 1 ? s1 : j == 2 ? s2 : s3;
}

// Two such sums over the SAME memories with one fetch of the attention weights (the command's gate images and its
// [PK | U2] columns in phase C: the waves that own both ran the two loops one after the other).
__device__ __forceinline__ void quad_column_sum2(const float *base_a, int stride_a, const float *base_b, int stride_b, int n,
                                                 int j, float alpha, float &out_a, float &out_b) {
    constexpr int U = GSCAN_DEC_QCS_U;
    f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < n; i0 += 4 * U) {
        f32x4 xa[U], xb[U];
        float am[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = i0 + 4 * u + j, mc = min(m, n - 1);
            xa[u] = *reinterpret_cast<const f32x4 *>(base_a + __mul24(mc, stride_a));
            xb[u] = *reinterpret_cast<const f32x4 *>(base_b + __mul24(mc, stride_b));
            const float al = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * mc, __float_as_int(alpha)));
            am[u] = m < n ? al : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc_a += am[u] * xa[u]; acc_b += am[u] * xb[u]; }
    }
    const float a0 = quad_sum(acc_a[0]), a1 = quad_sum(acc_a[1]), a2 = quad_sum(acc_a[2]), a3 = quad_sum(acc_a[3]);
    const float b0 = quad_sum(acc_b[0]), b1 = quad_sum(acc_b[1]), b2 = quad_sum(acc_b[2]), b3 = quad_sum(acc_b[3]);
    out_a = j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3;
    out_b = j == 0 ? b0 : j == 1 ? b1 : j == 2 ? b2 : b3;
}

// The output head of a row's T steps (see the comment in decoder_fwd_body where it is called), on the matrix cores
// (round 5): per chunk of 32 steps, logits[32, V] = S[32, 4H] . Wc^T as two 16 x 16 MFMA tiles whose K extent (H steps of
// four) is dealt over the eight waves; the waves' partial tiles are added in wave order through LDS (a fixed order:
// bitwise reproducible), then log_softmax per step and the row's loss terms.  Rounds 1-4 took the same dots as four lanes
// per logit with both operands from LDS: two 16-byte LDS reads per four multiply-adds, 6 k cycles of LDS bandwidth.
template <int H, int NT>
__device__ __forceinline__ void head_epilogue(const DecoderArgs &a, float *smem, int b, int tid, int lane, int wave, int T) {
    static_assert(NT == 512, "eight waves: the partial tiles are [8][2][256]");
    __syncthreads();                                         // the row's S is complete and visible to the workgroup
    constexpr int SS = 4 * H + 4;                            // row stride of both LDS images (16-byte aligned rows)
    constexpr int SCH = kHeadChunk * SS > 2 * 8 * 256 ? kHeadChunk * SS : 2 * 8 * 256;
    const int V = a.V;
    float *S_ch = smem, *red = smem, *wc_s = S_ch + SCH, *lg = wc_s + kHeadWcRows * SS, *lse_s = lg + kHeadChunk * 16;
    for (int i = tid; i < kHeadWcRows * H; i += NT) {        // 16-byte units; rows past the vocabulary are zero
        const int v = i / H, c4 = i - v * H;
        *reinterpret_cast<float4 *>(wc_s + v * SS + 4 * c4) =
            v < V ? reinterpret_cast<const float4 *>(a.head_wc)[i] : float4{0.f, 0.f, 0.f, 0.f};
    }
    float nll_acc = 0.f, cnt_acc = 0.f;                      // get_loss terms of this thread's (step mod 32, logit)
    const int fi = lane & 15, fk = lane >> 4;                // MFMA 16x16x4: A[i = fi][k = fk] = S row, B[k = fk][j = fi] = Wc[j][k]
    for (int t0 = 0; t0 < T; t0 += kHeadChunk) {
        const int n = min(kHeadChunk, T - t0);
        const unsigned bt0 = (unsigned)b * T + t0;
        {
            const float4 *src4 = reinterpret_cast<const float4 *>(a.s + bt0 * 4 * H);
            for (int idx = tid; idx < n * H; idx += NT) {
                const int row = idx / H, c4 = idx - row * H;
                *reinterpret_cast<float4 *>(S_ch + row * SS + 4 * c4) = src4[idx];
            }
        }
        lds_barrier();
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        {
            const float *a0 = S_ch + fi * SS + fk, *a1 = a0 + 16 * SS, *bw = wc_s + fi * SS + fk;
            if (n > 16) {
                for (int ks = wave; ks < H; ks += NT / 64) {
                    const float bv = bw[4 * ks];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * ks], bv, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * ks], bv, acc1, 0, 0, 0);
                }
            } else {
                for (int ks = wave; ks < H; ks += NT / 64) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * ks], bw[4 * ks], acc0, 0, 0, 0);
            }
        }
        lds_barrier();                                       // every wave is done with the S rows: their place takes the partial tiles
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[(wave * 2 + 0) * 256 + r * 64 + lane] = acc0[r];
            red[(wave * 2 + 1) * 256 + r * 64 + lane] = acc1[r];
        }
        lds_barrier();
        {   // thread -> element of a C fragment: tile = tid >> 8, register = (tid >> 6) & 3, lane = tid & 63;
            // row = 4 (lane >> 4) + register, column = lane & 15
            const int tile = tid >> 8, r = (tid >> 6) & 3, l = tid & 63;
            float sum = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NT / 64; ++w8) sum += red[(w8 * 2 + tile) * 256 + (tid & 255)];
            lg[(16 * tile + 4 * (l >> 4) + r) * 16 + (l & 15)] = sum;
        }
        lds_barrier();
        {   // log_softmax of every step of the chunk at once: thread (step tid >> 4, logit tid & 15), the sixteen lanes of a
            // step side by side in a wave: maximum and sum over the vocabulary on the DPP datapath (four in-row steps)
            const int tt = tid >> 4, v = tid & 15;
            const bool mine = tt < n && v < V;
            const float x = mine ? lg[tt * 16 + v] : -INFINITY;
            float mx = x;
            mx = fmaxf(mx, dpp_move<0xb1, 0xf>(mx));
            mx = fmaxf(mx, dpp_move<0x4e, 0xf>(mx));
            mx = fmaxf(mx, dpp_move<0x124, 0xf>(mx));
            mx = fmaxf(mx, dpp_move<0x128, 0xf>(mx));
            float sum = mine ? __expf(x - mx) : 0.f;
            sum += dpp_move<0xb1, 0xf>(sum);
            sum += dpp_move<0x4e, 0xf>(sum);
            sum += dpp_move<0x124, 0xf>(sum);
            sum += dpp_move<0x128, 0xf>(sum);
            const float lse = mx + __logf(sum);
            if (mine) {
                const unsigned at = (bt0 + tt) * V + v;
                a.logits[at] = x;
                a.logp_saved[at] = x - lse;
                a.logp_out[at] = x - lse;
            }
            if (a.row_stats) {
                // the target of position t is token t+1, literal 0 after the last one (model.py:108-115);
                // pad targets are ignored (nn.NLLLoss(ignore_index=pad), model.py:100)
                const int t = t0 + tt;
                const int64_t tgt = (tt < n && t + 1 < T) ? a.targets[(int64_t)b * T + t + 1] : (int64_t)0;
                if (mine && tgt != a.pad_tgt && tgt >= 0 && tgt == v) { nll_acc += lse - x; cnt_acc += 1.f; }
            }
        }
        lds_barrier();
    }
    if (a.row_stats) {                                       // every thread holds the terms of its (step mod 32, logit)
        nll_acc = wave_sum(nll_acc);
        cnt_acc = wave_sum(cnt_acc);
        if (lane == 0) { lse_s[2 * wave] = nll_acc; lse_s[2 * wave + 1] = cnt_acc; }
        lds_barrier();
        if (tid == 0) {
            float n0 = 0.f, n1 = 0.f;
            for (int i = 0; i < NT / 64; ++i) { n0 += lse_s[2 * i]; n1 += lse_s[2 * i + 1]; }
            a.row_stats[4 * b + 0] = n0;
            a.row_stats[4 * b + 1] = n1;
            a.row_stats[4 * b + 3] = 1.f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// GREEDY = false: teacher forcing over the T given target tokens, everything backward needs is saved (training /
// scoring).  GREEDY = true (predict.py:101-112): the row feeds its own argmax back for up to T steps and stops at
// <EOS>; the embedding part of the gates is a row of a [V, 4H] table, the output head is the composite
// W_h2o . W_o2h ([V, 4H], in LDS) applied every step, nothing is saved but tokens and attention rows.
// UVL = false: the visual gate images are streamed from L2 (decoder_lds); compiled for the hidden sizes whose 8x8-grid
// memories overflow LDS only.
//
// One step = six phases, a workgroup barrier behind each:
//   A  every product with h_{t-1}: W_hh h (stays in the owning lanes' registers), W_query_text h -> LDS,
//      W_q2k[:, :H] h (or W_query_vis h) -> LDS
//   B  textual scores v . tanh(q + PK_m), a wave per memory
//   C  softmax (in every wave's registers), then column sums over the command: the gate images U_t (kept in the
//      registers of the lane that owns the gate), the textual context, and the conditional query tanh(. + U2_t)
//   D  visual query from the conditional query (conditional attention only)
//   E  visual scores
//   F  softmax, column sums over the grid cells (U_v -> owning lanes; PK_v -> visual context), gate activations —
//      one gate per lane —, cell update by lane 0 of each quad, h_t -> LDS
// ------------------------------------------------------------------------------------------
template <int H, bool COND, bool GREEDY, bool UVL = true>
__device__ __forceinline__ void decoder_fwd_body(const DecoderArgs &a) {
    constexpr int R = (COND ? 7 : 6) * H, NS = (R + kDecPairs - 1) / kDecPairs, K0 = ((H / 2 + 3) / 4) * 4,
                  HP = 2 * K0;
    constexpr int NQ2 = (COND ? 2 : 1) * H / 4;   // column quads of [PK_t | U2_t]
#ifndef GSCAN_DEC_DEFER0
#define GSCAN_DEC_DEFER0 1
#endif
    // Round 5: the dot products of slot 0 — W_hh rows only once 4H >= 256 — leave phase A for phase E: their result is not
    // needed before the gates of phase F, and phase A (the densest in FMAs) no longer waits for them.  Phase A 1 269 -> 909
    // cycles, phase E 2 125 -> 2 557, the other phases -170: forward kernel 101.5 -> 100.4 us (profiles/
    // r05_decoder_fwd_deferred_slot0_ab.txt; in phase B or D, or staggered between a SIMD's two waves, the same dot costs
    // what it saves).
    constexpr bool DEFER0 = GSCAN_DEC_DEFER0 && 4 * H >= kDecPairs && NS >= 2 && !GREEDY;
    static_assert(H <= 128 && 4 * NQ2 <= kDecThreads, "hidden size too large for the quad roles");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = kDecThreads / 64;
    const int pair = decoder_pair_of(tid), half = decoder_half_of(tid);
    const int T = a.T, L = a.L, M = a.M;
    constexpr bool uv_lds = UVL;
    const DecoderLds o = decoder_lds(H, L, M, a.V, COND, false, uv_lds);
    float *PKv = smem + o.pkv, *PKt = smem + o.pkt;
    float *vec = smem + o.vec;
    float *h_s = vec;                                       // dot input, zero-padded to HP
    float *qt_s = vec + HP, *qv_s = qt_s + H, *vt_s = qv_s + H, *vv_s = vt_s + H;
    float *qh_s = vv_s + H;                                 // W_q2k[:, :H] h (conditional)
    float *ctxt_s = qh_s + H;                               // textual context (greedy head)
    float *qst_s = ctxt_s + H, *qsv_s = qst_s + H;          // the two queries pre-scaled (GSCAN_DEC_RFORM), written with them
    float *q2_s = qsv_s + H;                                // dot input, zero-padded to HP
    float *sc_s = q2_s + HP, *bq_s = sc_s + 64;
    [[maybe_unused]] float *stamp_acc = sc_s + 192;
    // greedy decoding only: composite head [V,4H] (S order), embedding part of the logits [V,V], visual context,
    // logits, current token (behind everything else in LDS)
    float *wc_s = smem + o.total, *le_s = wc_s + a.V * 4 * H, *ctxv_s = le_s + a.V * a.V, *logit_s = ctxv_s + H;
    int *tok_s = reinterpret_cast<int *>(logit_s + 64);
    [[maybe_unused]] long long stamp_prev = (GSCAN_STAMPS_ON && a.stamps) ? clock64() : 0;
    int len = a.cmd_lengths[b];
    len = max(1, min(len, L));
    // roles in the column-sum phases
    const GateLane gl = gate_lane<H>(tid);
    const int j4 = tid & 3;
    const float *ut_col = smem + o.ut + (gl.valid ? 4 * gl.unit : 0);           // U_t column quad of this lane's unit
    const int ut_stride = gl.valid ? 4 * H : 0;
    // [PK_t | U2_t] column quad (threads 0 .. 4 NQ2 - 1): columns < H -> textual context, the rest -> conditional query
    const int cq2 = tid >> 2;
    const bool has2 = cq2 < NQ2;
    const float *k2_col = smem + (!has2 ? o.pkt : (4 * cq2 < H ? o.pkt + 4 * cq2 : o.u2t + (4 * cq2 - H)));
    const int k2_stride = has2 ? H : 0;
    // visual phase: lanes with a unit sum its U_v column quad; the first H/4 quads WITHOUT a unit (upper quads of the
    // groups 8g, g >= H - 64) sum the PK_v column quads = the visual context
    const int spare = gl.upper && !gl.valid ? (tid >> 3) - max(0, H - 64) : -1;
    const bool ctx_role = spare >= 0 && spare < H / 4;
    const float *uv_col = smem + (gl.valid ? o.uv + 4 * gl.unit : o.pkv + (ctx_role ? 4 * spare : 0));
    const int uv_stride = gl.valid ? 4 * H : (ctx_role ? H : 0);
    static_assert(H <= 64 || 4 * (128 - H) >= H, "not enough spare quads for the visual context");
    // activation of gate j as a * sigmoid(s x) + c: tanh x = 2 sigmoid(2x) - 1 for the cell candidate (gate 2)
    const float act_s = gl.gate == 2 ? -2.8853900817779268f : -1.4426950408889634f;
    const float act_a = gl.gate == 2 ? 2.f : 1.f, act_c = gl.gate == 2 ? -1.f : 0.f;

    // ---- one-time loads, ONE queue and one wait (round 5): first the small per-row values into registers (initial state,
    //      energy vectors, query bias), then the row's memories straight to LDS (they come from HBM: the longest
    //      latency goes first), then the register image of the weights (L2 hits, 39 16-byte loads per thread).  Round 4
    //      staged the memories through registers behind the image and fetched the small values behind a barrier after
    //      that: three dependent round trips where one is enough.
    const int64_t h0_at = (int64_t)b * (GREEDY ? 1 : T) * H;
    const float h0_reg = tid < H ? a.hprev[h0_at + tid] : 0.f;
    const float vt_reg = tid < H ? a.v_t[tid] : 0.f, vv_reg = tid < H ? a.v_v[tid] : 0.f;
    const float bq_reg = (COND && tid < H) ? a.b_q2k[tid] : 0.f;
    float c = 0.f;                                          // cell state: lane 0 of the quad of each unit
    if (gl.valid && gl.gate == 0)                           // c0 = h0 unless given (seq2seq_model.py:494-504)
        c = a.c0 ? a.c0[(int64_t)b * H + gl.unit] : a.hprev[h0_at + gl.unit];
    stage_all(smem, stage_list(a, o, b, H, L, M, COND, uv_lds), tid);
    if (GREEDY) stage_array(smem, a.head_wc, (int)(wc_s - smem), a.V * H, tid);   // the whole head [V, 4H] behind the memories
    float w[NS][K0];
    {   // 16-byte loads: a workgroup's request for one image step spans 8 KB (four times as many L2 channels busy as
        // with dword loads while all 256 workgroups walk the same image at the same time)
        static_assert(K0 % 4 == 0, "a thread's half-row is a whole number of 16-byte groups");
        const float4 *img4 = reinterpret_cast<const float4 *>(a.w_image);
#pragma unroll
        for (int q = 0; q < NS * K0 / 4; ++q) {
            const float4 v = img4[q * kDecThreads + tid];
            w[(4 * q) / K0][(4 * q) % K0] = v.x; w[(4 * q) / K0][(4 * q) % K0 + 1] = v.y;
            w[(4 * q) / K0][(4 * q) % K0 + 2] = v.z; w[(4 * q) / K0][(4 * q) % K0 + 3] = v.w;
        }
    }
    if (tid < HP) { h_s[tid] = tid < H ? h0_reg : 0.f; q2_s[tid] = 0.f; }   // dot inputs, their padding zero
    if (tid < H) { vt_s[tid] = vt_reg; vv_s[tid] = vv_reg; }
    if (COND && tid < H) bq_s[tid] = bq_reg;
    if (GSCAN_STAMPS_ON && tid < 16) stamp_acc[tid] = 0.f;
    if (GREEDY && tid == 0) tok_s[0] = a.sos;
    float att_acc = 0.f;                                    // wave 0, lane m
    // The weight registers are complete from here on (and the compiler's wait-count bookkeeping knows: without an
    // explicit wait it re-checks "the first weight load may still be in flight" at the top of EVERY iteration with
    // s_waitcnt vmcnt(2), which in steady state drains the previous step's stores: vmcnt counts in order), and so are the
    // memories in LDS.
    staged_barrier();
    if (GREEDY) {   // le[v][v'] = Wc[v', 0:H] . Emb[v]: what the token fed in contributes to the logits
        const int V = a.V;
        for (int i = tid; i < V * V; i += kDecThreads) {
            const int v = i / V, vo = i - v * V;
            float acc = 0.f;
            for (int k = 0; k < H; ++k) acc = fmaf(wc_s[vo * 4 * H + k], a.dec_emb[v * H + k], acc);
            le_s[i] = acc;
        }
        lds_barrier();
    }
    // Softmax shift (GSCAN_DEC_SOFTMAX_BOUND=1): a score v . tanh(.) lies in [-|v|_1, |v|_1], so |v|_1 can stand in for the
    // maximum the reference's softmax subtracts (seq2seq_model.py:136; the quotient is the same number) — one 64-lane
    // reduction chain (six DPP steps + a readlane, ~90 cycles) per attention and step fewer — as long as exp(-2 |v|_1)
    // stays a normal float: |v|_1 < 40; beyond that the maximum is taken as before (wave-uniform choice per launch).
#ifndef GSCAN_DEC_SOFTMAX_BOUND
#define GSCAN_DEC_SOFTMAX_BOUND 1
#endif
    float shift_t = 0.f, shift_v = 0.f;
    bool bound_t = false, bound_v = false;
    if (GSCAN_DEC_SOFTMAX_BOUND) {
        const float at1 = (lane < H ? fabsf(vt_s[lane]) : 0.f) + (lane + 64 < H ? fabsf(vt_s[lane + 64]) : 0.f);
        const float av1 = (lane < H ? fabsf(vv_s[lane]) : 0.f) + (lane + 64 < H ? fabsf(vv_s[lane + 64]) : 0.f);
        shift_t = wave_sum(at1);
        shift_v = wave_sum(av1);
        bound_t = shift_t < 40.f;
        bound_v = shift_v < 40.f;
        if (GSCAN_DEC_RFORM && GSCAN_DEC_SCORE_HALF) {   // the r form's score is sum_k 2 v_k r_k, r in (0, 1): at most 2 sum_k max(v_k, 0)
            const float pt1 = (lane < H ? fmaxf(vt_s[lane], 0.f) : 0.f) + (lane + 64 < H ? fmaxf(vt_s[lane + 64], 0.f) : 0.f);
            const float pv1 = (lane < H ? fmaxf(vv_s[lane], 0.f) : 0.f) + (lane + 64 < H ? fmaxf(vv_s[lane + 64], 0.f) : 0.f);
            shift_t = 2.f * wave_sum(pt1);
            shift_v = 2.f * wave_sum(pv1);
        }
    }
    // GSCAN_DEC_PRIO=1: the second-dispatched half of the workgroup (waves 4-7: the younger wave of every SIMD, which loses
    // the issue arbitration by age) runs at priority 1 for the whole loop; 2: the older half instead (A/B).
#ifndef GSCAN_DEC_PRIO
#define GSCAN_DEC_PRIO 0
#endif
    if (GSCAN_DEC_PRIO == 1) { if (__builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1); }
    if (GSCAN_DEC_PRIO == 2) { if (__builtin_amdgcn_readfirstlane(wave) < 4) __builtin_amdgcn_s_setprio(1); }
    int steps_done = 0;
    GSCAN_STAMP_ONCE(10)

    for (int t = 0; t < T; ++t) {
        GSCAN_STAMP(0)
        const unsigned bt = (unsigned)b * T + t;            // 32-bit offsets: B*T*4H < 2^31 is checked on the host
        // embedding part of this lane's gate (+ both biases): issued first, consumed in phase C.  Unconditional (a
        // guarded load makes the compiler drain vmcnt first); greedy: row of the [V, 4H] table of the token fed in
        const unsigned ge_row = GREEDY ? (unsigned)tok_s[0] : bt;
        const float ge = a.ge[ge_row * 4 * H + (gl.valid ? gl.gate * H + gl.unit : 0)];

        // ---- A: everything that multiplies h_{t-1} -------------------------------------------
        float gh[NS];
        if constexpr (DEFER0) {                             // slot 0 (W_hh rows only) waits for phase E
            gh[0] = 0.f;
            float gh12[NS - 1];
            shared_dots<NS - 1, K0>(reinterpret_cast<const float(&)[NS - 1][K0]>(w[1]), h_s + half * K0, gh12);
#pragma unroll
            for (int s = 1; s < NS; ++s) gh[s] = gh12[s - 1];
        } else {
            shared_dots<NS, K0>(w, h_s + half * K0, gh);    // rows >= 6H (phase D rows) compute an unused value
        }
        float ghh = 0.f;                                    // W_hh h of this lane's (unit, gate)
#pragma unroll
        for (int s = DEFER0 ? 1 : 0; s < NS; ++s) {
            const int r = s * kDecPairs + pair;
            const float acc = pair_sum(gh[s]);
            if (s * kDecPairs < 4 * H && s < 2) { if (gl.upper == (s == 1)) ghh = acc; }
            if (r >= 4 * H && r < 6 * H && half == 0) {
                if (r < 5 * H) { qt_s[r - 4 * H] = acc; if (GSCAN_DEC_RFORM) qst_s[r - 4 * H] = kM2Log2e * acc; }
                else if (COND) qh_s[r - 5 * H] = acc;
                else { qv_s[r - 5 * H] = acc; if (GSCAN_DEC_RFORM) qsv_s[r - 5 * H] = kM2Log2e * acc; }
            }
        }
        lds_barrier();
        GSCAN_STAMP(1)

        // ---- B: textual scores s_m = v . tanh(q + PK_m), m < len (seq2seq_model.py:129-135) --
        attention_scores<H>(vt_s, qt_s, PKt, len, sc_s, wave, nwave, lane, qst_s);
        lds_barrier();
        GSCAN_STAMP(2)
        // ---- C: softmax over the command (seq2seq_model.py:136-137) in every wave's registers: lane m holds
        // alpha_m; then the column sums sum_m alpha_m [U_t | PK_t | U2_t][m, :]
        float alpha;
        {
            const float x = (lane < len) ? sc_s[lane] : -INFINITY;
            const float mx = bound_t ? shift_t : wave_max(x);
            const float e = (lane < len) ? __expf(x - mx) : 0.f;
            alpha = e * __builtin_amdgcn_rcpf(wave_sum(e));
        }
#ifndef GSCAN_DEC_QCS2
#define GSCAN_DEC_QCS2 1          // 1 (round 6): the waves that sum the [PK_t | U2_t] columns too do both sums in ONE loop
#endif
        float uct, s2 = 0.f;                                                      // this lane's gate, textual part
        if (GSCAN_DEC_QCS2 && wave < (4 * NQ2 + 63) / 64) quad_column_sum2(ut_col, ut_stride, k2_col, k2_stride, len, j4, alpha, uct, s2);
        else uct = quad_column_sum(ut_col, ut_stride, len, j4, alpha);
        // This is the one place where a wave waits for a global load (ge): vmcnt counts stores too, so every store
        // of phases A-C is issued behind this wait and has most of a step to be acknowledged before the next one
        float pre = ge + ghh + uct;
        asm volatile("" : "+v"(pre));
        if (wave < (4 * NQ2 + 63) / 64) {
            if (!GSCAN_DEC_QCS2) s2 = quad_column_sum(k2_col, k2_stride, len, j4, alpha);
            const int col = 4 * cq2 + j4;
            if (has2) {
                if (col < H) {
                    ctxt_s[col] = s2;
                    if (!GREEDY) { a.s[bt * 4 * H + H + col] = s2; a.qt[bt * H + col] = qt_s[col]; }
                } else {                                    // conditional query (seq2seq_model.py:394-396)
                    const float q = tanhf_(s2 + qh_s[col - H] + bq_s[col - H]);
                    q2_s[col - H] = q;
                    if (!GREEDY) a.q2[bt * H + col - H] = q;
                }
            }
        }
        if (wave == 0 && lane < L) a.alpha_c[bt * L + lane] = alpha;
        if (COND) {
            lds_barrier();
            GSCAN_STAMP(3)
            // ---- D: visual query from the conditional query ---------------------------------
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int r = s * kDecPairs + pair;
                if (r >= 6 * H && r < 7 * H) {
                    const float acc = pair_sum(half_dot<K0>(w[s], q2_s + half * K0));
                    if (half == 0) {
                        qv_s[r - 6 * H] = acc;
                        if (GSCAN_DEC_RFORM) qsv_s[r - 6 * H] = kM2Log2e * acc;
                        if (!GREEDY) a.qv[bt * H + r - 6 * H] = acc;
                    }
                }
            }
        } else if (!GREEDY && tid < H) {
            a.qv[bt * H + tid] = qv_s[tid];
        }
        lds_barrier();
        GSCAN_STAMP(5)

        // ---- E: visual scores over all M cells (no mask: every row has M memories) -----------
        if constexpr (DEFER0) {   // slot 0's W_hh rows against h_{t-1} (still in h_s: phase F writes h_t)
            const float g0 = pair_sum(half_dot<K0>(w[0], h_s + half * K0));
            if (!gl.upper) pre += g0;
        }
        attention_scores<H>(vv_s, qv_s, PKv, M, sc_s, wave, nwave, lane, qsv_s);
        lds_barrier();
        GSCAN_STAMP(6)
        // ---- F: softmax, column sums over the cells, gates, cell update (seq2seq_model.py:414) ----
        {
            const float x = (lane < M) ? sc_s[lane] : -INFINITY;
            const float mx = bound_v ? shift_v : wave_max(x);
            const float e = (lane < M) ? __expf(x - mx) : 0.f;
            alpha = e * __builtin_amdgcn_rcpf(wave_sum(e));
            if (wave == 0) {
                if (lane < M) a.alpha_s[bt * M + lane] = alpha;
                att_acc += alpha;                              // seq2seq_model.py:479,490
            }
        }
        float ucv;
        if (uv_lds) {
            if (GSCAN_DEC_QCS_EXACT && M % (4 * GSCAN_DEC_QCS_U) == 0) ucv = quad_column_sum_exact(uv_col, uv_stride, M, j4, alpha);
            else ucv = quad_column_sum(uv_col, uv_stride, M, j4, alpha);
        } else {                                               // U_vis streamed from L2 (it did not fit LDS)
            const float from_l2 = quad_column_sum<true>(a.u_v + (int64_t)b * M * 4 * H + (gl.valid ? 4 * gl.unit : 0),
                                                        gl.valid ? 4 * H : 0, M, j4, alpha);
            const float from_lds = quad_column_sum(smem + o.pkv + (ctx_role ? 4 * spare : 0), ctx_role ? H : 0, M, j4, alpha);
            ucv = gl.valid ? from_l2 : from_lds;
        }
        {
            // one gate per lane: i, f, o = sigmoid, g = tanh, as a * sigmoid(s x) + c
            const float act = fmaf(act_a, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(act_s * (pre + ucv))), act_c);
            const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
            if (gl.valid) {
                if (!GREEDY) a.gates[bt * 4 * H + gl.gate * H + gl.unit] = act;
                if (gl.gate == 0) {
                    c = gf * c + gi * gg;
                    const float h = go * tanhf_(c);
                    h_s[gl.unit] = h;
                    if (!GREEDY) {
                        a.cells[bt * H + gl.unit] = c;
                        a.s[bt * 4 * H + 3 * H + gl.unit] = h;
                        if (t + 1 < T) a.hprev[(bt + 1) * H + gl.unit] = h;
                    }
                }
            } else if (ctx_role) {
                const int col = 4 * spare + j4;
                if (GREEDY) ctxv_s[col] = ucv;
                else a.s[bt * 4 * H + 2 * H + col] = ucv;
            }
        }
        lds_barrier();
        GSCAN_STAMP(7)
        if (GREEDY) {
            // output head on [e | ctx_text | ctx_vis | h] (seq2seq_model.py:421-424) as the composite W_h2o . W_o2h,
            // argmax (= argmax of log_softmax, predict.py:106-107; the first of equal maxima), feed back, stop at <EOS>
            const int V = a.V, tok = tok_s[0];
            for (int v = wave; v < V; v += nwave) {
                const float *wrow = wc_s + v * 4 * H;
                float p = 0.f;
                for (int k = lane; k < H; k += 64)
                    p += wrow[H + k] * ctxt_s[k] + wrow[2 * H + k] * ctxv_s[k] + wrow[3 * H + k] * h_s[k];
                p = wave_sum(p);
                if (lane == 0) logit_s[v] = p + le_s[tok * V + v];
            }
            lds_barrier();
            if (tid == 0) {
                int best = 0;
                float top = logit_s[0];
                for (int v = 1; v < V; ++v)
                    if (logit_s[v] > top) { top = logit_s[v]; best = v; }
                tok_s[0] = best;
                a.tokens_out[(int64_t)b * T + t] = best;
            }
            steps_done = t + 1;
            lds_barrier();
            if (tok_s[0] == a.eos) break;                  // uniform: every thread reads the same LDS word
        }
    }
    if (GREEDY) {
        if (tid == 0) a.steps_out[b] = steps_done;
        if (wave == 0 && lane < M) a.att_sum[(int64_t)b * M + lane] = att_acc;
        return;
    }
    if (a.h_last && tid < H) a.h_last[(int64_t)b * H + tid] = h_s[tid];   // h_T (own LDS write, no barrier needed)
    if (wave == 0) {
        if (lane < M) a.att_sum[(int64_t)b * M + lane] = att_acc;
        if (a.aux_saved) {                                   // auxiliary head: log_softmax over the cells (model.py:205)
            const float x = (lane < M) ? att_acc : -INFINITY;
            const float mx = wave_max(x);
            const float lse = mx + logf(wave_sum((lane < M) ? expf(x - mx) : 0.f));
            if (lane < M) {
                a.aux_saved[(int64_t)b * M + lane] = x - lse;
                a.aux_out[(int64_t)b * M + lane] = x - lse;
            }
            if (a.row_stats) {                               // get_auxiliary_loss (model.py:162-164), this row's term
                const int64_t pos = a.positions ? a.positions[b] : (int64_t)-1;
                const float nll = wave_sum((lane < M && lane == pos) ? lse - x : 0.f);
                if (lane == 0) a.row_stats[4 * b + 2] = nll;
            }
        } else if (a.row_stats && lane == 0) {
            a.row_stats[4 * b + 2] = 0.f;
        }
    }
    if (GSCAN_STAMPS_ON && a.stamps && blockIdx.x == 0 && tid < 10) a.stamps[tid] = stamp_acc[tid];

    // ---- output head of the row's T steps (it does not feed back, seq2seq_model.py:421-424).  The reference applies
    //      hidden_to_output(output_to_hidden(.)), two bias-free Linears with NOTHING between them: their product
    //      Wc = W_h2o . W_o2h ([V, 4H], columns in S order; written by the step prologue) IS the head, so
    //      logits_t = Wc . S_t costs V . 4H multiply-adds per step instead of H . 4H + V . H (11 x fewer at V = 9), and the
    //      H-wide intermediate is never formed (its weight gradients follow from d Wc = dlogits^T . S, step.hip).
    //      Four lanes per (t, v) logit, each a quarter of the 4H-deep dot from LDS, added on the DPP datapath; then
    //      logp_t = log_softmax(logits_t) (model.py:203) and the row's NLL partial sums by thread t.
    head_epilogue<H, kDecThreads>(a, smem, b, tid, lane, wave, T);
    GSCAN_STAMP_ONCE(11)
}

template <int H, bool COND, bool GREEDY, bool UVL = true>
__global__ __launch_bounds__(kDecThreads) void decoder_fwd_kernel(DecoderArgs a) {
    TraceScope trace_scope(TK_DECODER_FWD);
    decoder_fwd_body<H, COND, GREEDY, UVL>(a);
}

// Backward of s_m = v . tanh(q + PK_m) for one attention.  Lane m of `dsm` holds d s_m.  Wave w owns the memories
// w, w+nwave, ...; a lane owns features (lane, lane+64):  dPK[m][k] += ds_m v_k (1 - th^2)  (accumulated over the
// T steps in LDS), the same term summed over this wave's memories goes to part_s[wave][k] (-> d q_k after the
// cross-wave sum), and dv_k += ds_m th is kept per lane.  (Round 5 A/B: the two attentions' sums in LDS instead — four
// registers fewer across the time loop, and no VGPR spill — cost the reverse kernel 1.8 us: two more LDS
// read-modify-writes per phase outweigh one scratch store and load per step.)
// x += v on an LDS word that no other lane touches in this phase (the wave-per-memory form).  GSCAN_DEC_DPK_ATOMIC=1: as ONE ds_add_f32 without a
// return value instead of read, add, write.  Measured in round 6 and OFF: with no two lanes on one address the float
// atomic still takes the score-backward phases from 2 940 + 1 626 to 11 617 + 3 770 cycles per step (reverse kernel
// 118.7 -> 213.9 us, profiles/r06_decoder_bwd_quad_layout_ab.txt) — the LDS executes them far below the rate of plain
// reads and writes (round 3 saw the same with eight waves adding onto the same words and blamed the collisions).
#ifndef GSCAN_DEC_DPK_ATOMIC
#define GSCAN_DEC_DPK_ATOMIC 0
#endif
__device__ __forceinline__ void lds_accumulate(float *x, float v) {
#if GSCAN_DEC_DPK_ATOMIC
    __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)x, v, 0, 0, false);
#else
    *x += v;
#endif
}

// Round 6: the same backward with a thread per (feature k = tid >> 2, group of memories g = tid & 3: m = g, g + 4, ...)
// instead of a wave per memory.  The terms of d q_k then sit in the four lanes of a QUAD and are added with two DPP steps:
// no per-wave partial sums through LDS, no summing phase, no barrier behind it (two phases and two barriers of the nine
// a reverse step had: 436 + 428 of its 12.9 k cycles).  ds_m comes from lane m of dsm (ds_bpermute; every wave holds the
// whole distribution), the energy-vector gradient dv_k is one loop-carried register per attention (it was two).
// Returns d q_k in every lane of the quad of feature k (undefined for k >= H).
#ifndef GSCAN_DEC_SB_QUAD
#define GSCAN_DEC_SB_QUAD 2       // 0: round 5's wave per memory; 1: quads (a feature per lane); 2: groups of eight (a feature pair)
#endif
#ifndef GSCAN_DEC_SBO_UV
#define GSCAN_DEC_SBO_UV 5        // rounds per straight-line block of the group-of-eight form: grid cells (36 = 5 rounds of 8)
#endif
#ifndef GSCAN_DEC_SBO_UT
#define GSCAN_DEC_SBO_UT 2        // ... command tokens (10 = 2 rounds of 8)
#endif
template <int H>
__device__ __forceinline__ float score_backward_quads(float dsm, const float *q_s, const float *v_s, const float *pk,
                                                      float *dpk, int n, float &dv_acc, int tid) {
    const int t4 = opaque(tid);
    const int k = t4 >> 2, g = t4 & 3;
    const bool has = k < H;
    const int kk = has ? k : 0;
    const float v = has ? v_s[kk] : 0.f;
    const float q = q_s[kk];
    float pdq = 0.f;
    // Rounds of U memories per lane, straight-line: every LDS read of a round (projected key, ds through ds_bpermute, the
    // running key gradient) is issued before the first tanh — as a loop over single memories with a guarded
    // read-add-write the compiler kept one dependent LDS round trip + transcendental chain + round trip per memory
    // (phase 3: 3 248 cycles for 9 memories per lane).  Lanes without a memory (m >= n) or a feature (k >= H) read a
    // clamped address and store nothing.
#ifndef GSCAN_DEC_SBQ_U
#define GSCAN_DEC_SBQ_U 3
#endif
    constexpr int U = GSCAN_DEC_SBQ_U;
    for (int m0 = 0; m0 < n; m0 += 4 * U) {                     // the same number of rounds in every lane
        float x[U], dsl[U], old[U];
        int at[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + 4 * u + g, mc = min(m, n - 1);
            live[u] = m < n;
            at[u] = __mul24(mc, H) + kk;
            x[u] = pk[at[u]];
            dsl[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * mc, __float_as_int(dsm)));
            old[u] = dpk[at[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float ds = live[u] ? dsl[u] : 0.f;
            const float th = tanhf_(q + x[u]);
            const float t = ds * v * (1.f - th * th);
            if (has && live[u]) dpk[at[u]] = old[u] + t;
            pdq += t;
            dv_acc = fmaf(ds, th, dv_acc);
        }
    }
    return quad_sum(pdq);
}

// The same with a thread per (feature PAIR k2 = tid >> 3, group of memories g = tid & 7: m = g, g + 8, ...): the projected
// keys, the running key gradients and their updates are 8-byte LDS accesses (half as many LDS instructions per term as
// the quad form: the phase is bound by LDS issue and transcendentals about equally), U rounds straight-line, and the two
// d q sums are added over the eight lanes of a group with three DPP steps (two in the quad, one row_half_mirror).
// Returns (d q_{2 k2}, d q_{2 k2 + 1}) in every lane of the group.
template <int H, int U>
__device__ __forceinline__ f32x2 score_backward_octs(float dsm, const float *q_s, const float *v_s, const float *pk,
                                                     float *dpk, int n, f32x2 &dv_acc, int tid) {
    static_assert(H % 2 == 0 && H <= 128, "feature pairs over 64 groups of eight lanes");
    const int t8 = opaque(tid);
    const int k2 = t8 >> 3, g = t8 & 7;
    const bool has = 2 * k2 < H;
    const int kk = has ? 2 * k2 : 0;
    const f32x2 q = *reinterpret_cast<const f32x2 *>(q_s + kk);
    f32x2 v = *reinterpret_cast<const f32x2 *>(v_s + kk);
    if (!has) v = f32x2{0.f, 0.f};
    f32x2 pdq = {0.f, 0.f};
    for (int m0 = 0; m0 < n; m0 += 8 * U) {                     // the same number of rounds in every lane
        f32x2 x[U], old[U];
        float dsl[U];
        int at[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + 8 * u + g, mc = min(m, n - 1);
            live[u] = m < n;
            at[u] = __mul24(mc, H) + kk;                          // 24-bit multiply: full rate (v_mul_lo_u32 is quarter rate)
            x[u] = *reinterpret_cast<const f32x2 *>(pk + at[u]);
            dsl[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * mc, __float_as_int(dsm)));
            old[u] = *reinterpret_cast<const f32x2 *>(dpk + at[u]);
        }
        // (the r form of the forward scores, GSCAN_DEC_RFORM, measured here too — 4 r (1 - r) for 1 - tanh^2, 2 ds r - ds for
        // ds tanh, pre-scaled queries: three instructions fewer per term and no change, 96.2-96.8 -> 96.9-97.1 us: this phase
        // is not bound by its arithmetic.  profiles/r06_decoder_score_tables_ab.txt, run 10)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float ds = live[u] ? dsl[u] : 0.f;
            const f32x2 th = {tanhf_(q.x + x[u].x), tanhf_(q.y + x[u].y)};
            const f32x2 t = {ds * v.x * (1.f - th.x * th.x), ds * v.y * (1.f - th.y * th.y)};
            if (has && live[u]) *reinterpret_cast<f32x2 *>(dpk + at[u]) = old[u] + t;
            pdq += t;
            dv_acc.x = fmaf(ds, th.x, dv_acc.x);
            dv_acc.y = fmaf(ds, th.y, dv_acc.y);
        }
    }
    pdq.x = quad_sum(pdq.x);
    pdq.y = quad_sum(pdq.y);
    pdq.x += dpp_move<0x141, 0xf>(pdq.x);                       // row_half_mirror: the other quad of the eight
    pdq.y += dpp_move<0x141, 0xf>(pdq.y);
    return pdq;
}

template <int H>
__device__ __forceinline__ void score_backward(float dsm, const float *q_s, const float *v_s, const float *pk,
                                               float *dpk, int n, float *part_s, f32x2 &dv_acc, int wave, int nwave,
                                               int lane) {
    const bool has2 = lane + 64 < H, has1 = lane < H;
    const int k1 = has1 ? lane : 0, k2 = has2 ? lane + 64 : 0;
    const float v1 = has1 ? v_s[k1] : 0.f, v2 = has2 ? v_s[k2] : 0.f;
    const float q1 = q_s[k1], q2 = q_s[k2];
    float pdq1 = 0.f, pdq2 = 0.f;
#ifndef GSCAN_DEC_SB_UNROLL
#define GSCAN_DEC_SB_UNROLL 2
#endif
#pragma unroll GSCAN_DEC_SB_UNROLL
    for (int m = wave; m < n; m += nwave) {
        const float ds = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dsm), m));
        const float th1 = tanhf_(q1 + pk[m * H + k1]), th2 = tanhf_(q2 + pk[m * H + k2]);
        const float t1 = ds * v1 * (1.f - th1 * th1), t2 = ds * v2 * (1.f - th2 * th2);
        if (has1) lds_accumulate(dpk + m * H + k1, t1);
        if (has2) lds_accumulate(dpk + m * H + k2, t2);
        pdq1 += t1;
        pdq2 += t2;
        dv_acc.x = fmaf(ds, th1, dv_acc.x);
        dv_acc.y = fmaf(ds, th2, dv_acc.y);
    }
    if (has1) part_s[wave * H + k1] = pdq1;
    if (has2) part_s[wave * H + k2] = pdq2;
}

// d alpha[m] = delta . U[m] (+ dzq . U2[m]) + dctx(ext) . PK[m] for the memories of one attention, one wave per
// memory, 16-byte LDS reads.  float4 index idx < H covers the gate deltas of unit idx (dperm_s, unit-major like U), the next H/4 the
// external context gradient against PK, the last H/4 (conditional, textual only) dzq against U2.  The left-hand
// vectors do not depend on m: each lane reads its (up to three) float4 of them once.
// u_global != NULL: the U rows are read from global memory ([n, 4H]) instead of LDS.
template <int H, int HP, bool WITH_U2, bool UGLOBAL = false>
__device__ __forceinline__ void dalpha_rows(const float *smem, const float *d_s, const float *dperm_s, const float *ext_s, int u_off,
                                            int pk_off, int u2_off, int n, const float *add, float *sc_s, int wave,
                                            int nwave, int lane, const float *u_global = nullptr) {
    constexpr int Q = H / 4, NQ = (WITH_U2 ? 6 : 5) * Q, NI = (NQ + 63) / 64;
    lane = opaque(lane);
    float4 x[NI];
    int yoff[NI], ystr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = lane + 64 * i;
        x[i] = float4{0.f, 0.f, 0.f, 0.f};
        yoff[i] = u_off;
        ystr[i] = 0;
        if (idx < 4 * Q) {              // the U images are unit-major: column quad idx = the four gates of unit idx
            x[i] = *reinterpret_cast<const float4 *>(dperm_s + 4 * idx);
            yoff[i] = u_off + 4 * idx; ystr[i] = 4 * H;
        } else if (idx < 5 * Q) {
            x[i] = *reinterpret_cast<const float4 *>(ext_s + 4 * (idx - 4 * Q));
            yoff[i] = pk_off + 4 * (idx - 4 * Q); ystr[i] = H;
        } else if (idx < NQ) {
            x[i] = *reinterpret_cast<const float4 *>(d_s + 5 * HP + 4 * (idx - 5 * Q));
            yoff[i] = u2_off + 4 * (idx - 5 * Q); ystr[i] = H;
        }
    }
#ifndef GSCAN_DEC_DALPHA_HALF
#define GSCAN_DEC_DALPHA_HALF 1   // 1 (round 6): a HALF wave per memory, two memories per wave round: four or five 16-byte
#endif                            // reads per lane and ONE five-step DPP sum for both (it was a wave per memory: two or three
                                  // reads, then a six-step sum + readlane per memory, one after the other)
#if GSCAN_DEC_DALPHA_HALF
    if constexpr (!UGLOBAL) {
        constexpr int NH = (NQ + 31) / 32;
        wave = __builtin_amdgcn_readfirstlane(wave);               // uniform: a scalar loop, not an exec-mask loop
        const int l32 = lane & 31, hf = lane >> 5;
        float4 xh[NH];
        int yo[NH], ys[NH];
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int idx = l32 + 32 * i;
            xh[i] = float4{0.f, 0.f, 0.f, 0.f};
            yo[i] = u_off; ys[i] = 0;
            if (idx < 4 * Q) { xh[i] = *reinterpret_cast<const float4 *>(dperm_s + 4 * idx); yo[i] = u_off + 4 * idx; ys[i] = 4 * H; }
            else if (idx < 5 * Q) { xh[i] = *reinterpret_cast<const float4 *>(ext_s + 4 * (idx - 4 * Q)); yo[i] = pk_off + 4 * (idx - 4 * Q); ys[i] = H; }
            else if (idx < NQ) { xh[i] = *reinterpret_cast<const float4 *>(d_s + 5 * HP + 4 * (idx - 5 * Q)); yo[i] = u2_off + 4 * (idx - 5 * Q); ys[i] = H; }
        }
        for (int m0 = 2 * wave; m0 < n; m0 += 2 * nwave) {          // memories m0 (lanes 0-31) and m0 + 1 (lanes 32-63)
            const int m = m0 + hf, mc = min(m, n - 1);
            // every read of the round in flight before the first multiply (written as one expression the compiler issued
            // read, wait, four dependent FMAs, four times over: four LDS round trips per round), the products as
            // independent chains per read, and the additive term fetched with them
            float4 y[NH];
#pragma unroll
            for (int j = 0; j < NH; ++j) y[j] = *reinterpret_cast<const float4 *>(smem + yo[j] + __mul24(mc, ys[j]));
            const float extra = add ? add[mc] : 0.f;
            asm volatile("" : "+v"(y[0].x));                        // (keeps the reads above the arithmetic)
            float pj[NH];
#pragma unroll
            for (int j = 0; j < NH; ++j) pj[j] = dot4(xh[j], y[j], 0.f);
            float p = l32 == 0 ? extra : 0.f;                       // ONE lane of the half brings the additive term into the sum
#pragma unroll
            for (int j = 0; j < NH; ++j) p += pj[j];
            p += dpp_move<0xb1, 0xf>(p);                            // quad_perm [1,0,3,2]
            p += dpp_move<0x4e, 0xf>(p);                            // quad_perm [2,3,0,1]
            p += dpp_move<0x124, 0xf>(p);                           // row_ror:4
            p += dpp_move<0x128, 0xf>(p);                           // row_ror:8: every lane of a 16-lane row holds the row's sum
            p += dpp_move<0x142, 0xa>(p);                           // row_bcast:15 into rows 1 and 3: lanes 16-31 / 48-63 hold a half's sum
            if (l32 == 31 && m < n) sc_s[m] = p;
        }
        return;
    }
#endif
#ifndef GSCAN_DEC_DALPHA_G
#define GSCAN_DEC_DALPHA_G 0      // > 0 (round 6 A/B, lost: 3 -> +1.5 us, 2 -> flat): G memories per straight-line round, their wave sums interleaved
#endif
#if GSCAN_DEC_DALPHA_G > 0
    if constexpr (!UGLOBAL) {
        constexpr int G = GSCAN_DEC_DALPHA_G;
        for (int m0 = wave; m0 < n; m0 += G * nwave) {
            float4 y[G][NI];
            float p[G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int m = min(m0 + i * nwave, n - 1);
#pragma unroll
                for (int j = 0; j < NI; ++j) y[i][j] = *reinterpret_cast<const float4 *>(smem + yoff[j] + m * ystr[j]);
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                p[i] = 0.f;
#pragma unroll
                for (int j = 0; j < NI; ++j) p[i] = dot4(x[j], y[i][j], p[i]);
            }
            wave_sum_n<G>(p);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int m = m0 + i * nwave;
                if (lane == 0 && m < n) sc_s[m] = p[i] + (add ? add[m] : 0.f);
            }
        }
        return;
    }
#endif
    for (int m0 = wave; m0 < n; m0 += 2 * nwave) {
        float p[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = min(m0 + i * nwave, n - 1);
            p[i] = 0.f;
            if (!UGLOBAL) {      // a template parameter, not a test of the pointer: that test breaks the back end in some builds
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    p[i] = dot4(x[j], *reinterpret_cast<const float4 *>(smem + yoff[j] + m * ystr[j]), p[i]);
            } else {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int idx = lane + 64 * j;
                    const float4 y = idx < 4 * Q ? *reinterpret_cast<const float4 *>(u_global + (int64_t)m * 4 * H + 4 * idx)
                                                 : *reinterpret_cast<const float4 *>(smem + yoff[j] + m * ystr[j]);
                    p[i] = dot4(x[j], y, p[i]);
                }
            }
        }
        // (the two sums one after the other, each only where its memory exists: interleaving them as wave_sum_n<2> —
        // unconditionally, for the straight-line code that needs — made the reverse kernel 5.7 us SLOWER, round 5 A/B)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + i * nwave;
            if (m < n) {
                const float tsum = wave_sum(p[i]);
                if (lane == 0) sc_s[m] = tsum + (add ? add[m] : 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backward through time.  Emits per-step pre-activation gradients (delta for the LSTM gates,
// dzq for the conditional query, dqt / dqv for the projected queries); every parameter
// gradient is a dense GEMM over those afterwards.  Key gradients along the score path are
// accumulated in LDS across the T steps and written once; the value path
// (dPK += alpha^T . dctx) is a separate batched product outside (it needs W_ih^T . delta,
// which is again a dense GEMM).
// ------------------------------------------------------------------------------------------
template <int H, bool COND, bool UVL = true>
__device__ __forceinline__ void decoder_bwd_body(const DecoderArgs &a) {
    constexpr int R = (COND ? 7 : 6) * H, NS = (R + kDecPairs - 1) / kDecPairs, K0 = ((H / 2 + 3) / 4) * 4,
                  HP = 2 * K0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = kDecThreads / 64;
    const int pair = decoder_pair_of(tid), half = decoder_half_of(tid);
    const int T = a.T, L = a.L, M = a.M;
    constexpr bool uv_lds = UVL;
    const DecoderLds o = decoder_lds(H, L, M, a.V, COND, true, uv_lds);
    float *Uv = smem + o.uv, *PKv = smem + o.pkv, *Ut = smem + o.ut, *PKt = smem + o.pkt, *U2t = smem + o.u2t;
    float *dPKv = smem + o.dpkv, *dPKt = smem + o.dpkt;
    float *vec = smem + o.vec;
#ifndef GSCAN_DEC_P1_LATE
#define GSCAN_DEC_P1_LATE 1       // 1 (round 6): the five global stores of phase 1 leave its two busy waves for waves 4-5, which
#endif                            // copy them out of LDS at the head of phase 2 (they have a memory fewer to score there)
    // (not where the visual gate images are streamed from L2: that instantiation has no register to spare)
    constexpr bool kP1Late = GSCAN_DEC_P1_LATE && UVL && kDecThreads >= 256 + H;
    float *d_s = vec;                 // [6][HP]: delta (4 blocks) | dqt | dzq or dqv, each zero-padded to HP
    float *dqv_s = vec + 6 * HP;      // [HP]
    float *qt_s = vec + 7 * HP, *q2_s = qt_s + H, *qv_s = q2_s + H, *vt_s = qv_s + H, *vv_s = vt_s + H,
          *exc_s = vv_s + H, *exs_s = exc_s + H, *part_s = exs_s + H;    // part_s: 8H
    float *dperm_s = part_s + 8 * H;  // [H][4] the gate deltas again, unit-major (the layout of the U images)
    float *sc_s = dperm_s + 4 * H, *al_s = sc_s + 64, *datt_s = sc_s + 128;
    [[maybe_unused]] float *stamp_acc = sc_s + 192;
    [[maybe_unused]] long long stamp_prev = (GSCAN_STAMPS_ON && a.stamps) ? clock64() : 0;
    int len = a.cmd_lengths[b];
    len = max(1, min(len, L));

    float aux_scale = (a.seeds && a.daux) ? a.seeds[1] : 1.f;
    // ---- backward of the output head for the row's T steps: dlogits_t = seed * (dlogp_t - exp(logp_t) sum dlogp_t)
    //      through log_softmax, then through the composite head Wc = W_h2o . W_o2h ([V, 4H], S order; see the forward
    //      epilogue): dS_t = Wc^T dlogits_t.  dlogits is kept for the weight gradients (d Wc = dlogits^T . S, a GEMM
    //      behind this kernel); dS is what the loop below (and the LSTM-input product after it) starts from.
    //      Round 5: a thread per (step, logit) of a 32-step chunk — coalesced loads, one exp each (it was a thread per
    //      step, nine exps and stores in a row) — and dS[32, 4H] = dlogits[32, 16] . Wc[16, 4H] on the matrix cores, a
    //      wave per 16-column tile (it was nine LDS float4 reads per output float4).
    {
        static_assert(H % 4 == 0, "4H is a whole number of 16-column MFMA tiles");
        constexpr int DS = kHeadDlStride;                    // row stride of the dlogits tile in LDS
        const int V = a.V;
        float *dl_ch = smem, *wc_s = dl_ch + kHeadChunk * DS, *red = wc_s + kHeadWcRows * 4 * H;
        for (int i = tid; i < kHeadWcRows * H; i += kDecThreads)       // 16-byte units; rows past the vocabulary are zero
            reinterpret_cast<float4 *>(wc_s)[i] = i < V * H ? reinterpret_cast<const float4 *>(a.head_wc)[i] : float4{0.f, 0.f, 0.f, 0.f};
        float sc = a.seeds ? a.seeds[0] : 1.f;
        if (a.nll_mode) {
            // the loss is mean-over-live-tokens NLL (+ w * mean-over-rows auxiliary NLL): every workgroup sums the
            // per-row partials of the forward pass in the same fixed order and seeds its row with 1/tokens, w/rows
            float p0 = 0.f, p1 = 0.f, p2 = 0.f;
            for (int r = tid; r < a.B; r += kDecThreads) {
                const float4 x = *reinterpret_cast<const float4 *>(a.row_stats + 4 * r);
                p0 += x.x; p1 += x.y; p2 += x.z;
            }
            p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
            if (lane == 0) { red[3 * wave] = p0; red[3 * wave + 1] = p1; red[3 * wave + 2] = p2; }
            lds_barrier();
            p0 = p1 = p2 = 0.f;
            for (int i = 0; i < kDecThreads / 64; ++i) { p0 += red[3 * i]; p1 += red[3 * i + 1]; p2 += red[3 * i + 2]; }
            sc = (a.nll_mode == 2) ? 1.f : 1.f / p1;
            aux_scale = a.aux_saved ? ((a.nll_mode == 2) ? a.w_aux : a.w_aux / (float)a.B) : 0.f;
            if (b == 0 && tid == 0) {
                a.stats_out[0] = p0; a.stats_out[1] = p1; a.stats_out[2] = p2; a.stats_out[3] = (float)a.B;
                a.seeds_out[0] = sc; a.seeds_out[1] = aux_scale;
                a.seeds_out[2] = p0 * sc + p2 * aux_scale;
            }
        }
        static_assert(kDecThreads == kHeadChunk * 16, "a thread per element of the chunk's 32 x 16 dlogits tile");
        const int tt = tid >> 4, v16 = tid & 15;             // this thread's (step of the chunk, logit)
        const int fi = lane & 15, fk = lane >> 4;            // MFMA 16x16x4 fragment coordinates
        for (int t0 = 0; t0 < T; t0 += kHeadChunk) {
            const int n = min(kHeadChunk, T - t0);
            const unsigned bt0 = (unsigned)b * T + t0;
            const bool mine = tt < n && v16 < V;
            const unsigned at = (bt0 + tt) * V + v16;
            float dl = 0.f;
            if (a.nll_mode) {                                // d(sum NLL)/d(logp) is -1 at the live target
                if (mine) {
                    const int t = t0 + tt;
                    const int64_t tgt = (t + 1 < T) ? a.targets[(int64_t)b * T + t + 1] : (int64_t)0;
                    const bool live = tgt != a.pad_tgt && tgt >= 0 && tgt < V;
                    dl = live ? sc * (expf(a.logp_saved[at]) - (v16 == tgt ? 1.f : 0.f)) : 0.f;
                }
            } else {
                // sum of the step's incoming gradients: the sixteen lanes of a step sit side by side in a wave
                const float dy = mine ? a.dlogp[at] : 0.f;
                float sum = dy;
                sum += dpp_move<0xb1, 0xf>(sum);             // quad_perm [1,0,3,2]
                sum += dpp_move<0x4e, 0xf>(sum);             // quad_perm [2,3,0,1]
                sum += dpp_move<0x124, 0xf>(sum);            // row_ror:4
                sum += dpp_move<0x128, 0xf>(sum);            // row_ror:8: every lane of the 16-lane row holds the total
                if (mine) dl = sc * (dy - expf(a.logp_saved[at]) * sum);
            }
            dl_ch[tt * DS + v16] = dl;                       // zero outside the chunk / the vocabulary
            if (mine) a.dlogits[at] = dl;
            lds_barrier();
            const int ksteps = (V + 3) / 4;
            for (int ct = wave; ct < H / 4; ct += kDecThreads / 64) {          // 16-column tiles of dS
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                for (int ks = 0; ks < ksteps; ++ks) {
                    const float bv = wc_s[(4 * ks + fk) * 4 * H + 16 * ct + fi];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dl_ch[fi * DS + 4 * ks + fk], bv, acc0, 0, 0, 0);
                    if (n > 16) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dl_ch[(16 + fi) * DS + 4 * ks + fk], bv, acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                // C fragment: row = 4 (lane >> 4) + r, column = lane & 15
                    const int row = 4 * fk + r;
                    if (row < n) a.ds[(bt0 + row) * 4 * H + 16 * ct + fi] = acc0[r];
                    if (16 + row < n) a.ds[(bt0 + 16 + row) * 4 * H + 16 * ct + fi] = acc1[r];
                }
            }
            lds_barrier();
        }
        __syncthreads();                                     // the row's dS is visible to the workgroup
    }
    GSCAN_STAMP_ONCE(10)

    // ---- set-up.  Everything small that the loop needs from global memory is requested FIRST, into registers — the
    //      energy vectors, the auxiliary head's inputs, the saved activations of the last step — then the register image
    //      of the weights (39 16-byte loads per thread) and the memories: one queue of loads, one wait.  (Round 4 issued
    //      the small loads behind the wait for the image: two more dependent round trips to L2 / HBM, ~4 k cycles.)
    float dc = 0.f;
    // saved activations of step t are fetched one iteration ahead (their HBM/L2 latency hides behind step t+1)
    float pf[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float alpha_v_pf = 0.f, alpha_c_pf = 0.f, alpha_v_nx = 0.f, alpha_c_nx = 0.f;   // lane m of every wave
    // every saved-activation access is base (SGPR pair) + one unsigned 32-bit offset, so that no per-thread
    // 64-bit pointer has to stay live (or spill) across the loop
    const unsigned utid = (unsigned)tid, uH = (unsigned)H, row0 = (unsigned)b * T;
    auto prefetch = [&](int t) {
        const unsigned bt = row0 + t;
        const int ln = opaque(lane);
        alpha_v_nx = a.alpha_s[bt * M + min(ln, M - 1)];    // lanes >= M / L hold a copy; masked where it is used
        alpha_c_nx = a.alpha_c[bt * L + min(ln, L - 1)];
        if (tid < H) {
            const unsigned g = bt * 4 * uH + utid;
            pf[0] = a.gates[g]; pf[1] = a.gates[g + uH]; pf[2] = a.gates[g + 2 * uH]; pf[3] = a.gates[g + 3 * uH];
            pf[4] = a.cells[bt * uH + utid];
            const float *prev = (t > 0) ? a.cells : a.hprev;          // c_{t-1}, or c_0 = h_0 (model.py:195)
            pf[5] = prev[(t > 0 ? bt - 1 : row0) * uH + utid];
            pf[6] = a.ds[g + 3 * uH];
        } else if (tid >= 128 && tid < 128 + H) {
            const unsigned kk = utid - 128;
            pf[0] = a.ds[bt * 4 * uH + uH + kk];
            pf[1] = a.ds[bt * 4 * uH + 2 * uH + kk];
            pf[2] = a.qt[bt * uH + kk];
            pf[3] = a.qv[bt * uH + kk];
            if (COND) pf[4] = a.q2[bt * uH + kk];
        }
    };
    prefetch(T - 1);                                        // (the row's dS is this workgroup's own, complete behind the barrier above)
    const float vt_reg = tid < H ? a.v_t[tid] : 0.f, vv_reg = tid < H ? a.v_v[tid] : 0.f;
    // auxiliary head backward: d att_sum = seed1 * (daux - exp(aux_logp) sum daux)  (model.py:205): wave 0, lane = cell
    const bool has_aux = a.nll_mode ? (a.aux_saved != nullptr) : (a.daux != nullptr);
    float aux_dy = 0.f, aux_y = 0.f;
    if (tid < 64 && has_aux) {
        if (a.nll_mode) {
            const int64_t pos = a.positions ? a.positions[b] : (int64_t)-1;
            aux_dy = (tid < M && tid == pos) ? -1.f : 0.f;
        } else {
            aux_dy = (tid < M) ? a.daux[(int64_t)b * M + tid] : 0.f;
        }
        aux_y = (tid < M) ? a.aux_saved[(int64_t)b * M + tid] : 0.f;
    }

    float wt[NS][K0];
    {
        static_assert(K0 % 4 == 0, "a thread's half-column is a whole number of 16-byte groups");
        const float4 *img4 = reinterpret_cast<const float4 *>(a.w_image);
#pragma unroll
        for (int q = 0; q < NS * K0 / 4; ++q) {
            const float4 v = img4[q * kDecThreads + tid];
            wt[(4 * q) / K0][(4 * q) % K0] = v.x; wt[(4 * q) / K0][(4 * q) % K0 + 1] = v.y;
            wt[(4 * q) / K0][(4 * q) % K0 + 2] = v.z; wt[(4 * q) / K0][(4 * q) % K0 + 3] = v.w;
        }
    }
    stage_all(smem, stage_list(a, o, b, H, L, M, COND, uv_lds), tid);
    for (int i = tid; i < M * H; i += kDecThreads) dPKv[i] = 0.f;
    for (int i = tid; i < L * H; i += kDecThreads) dPKt[i] = 0.f;
    for (int i = tid; i < 7 * HP; i += kDecThreads) vec[i] = 0.f;        // d_s / dqv_s incl. padding
    for (int i = tid; i < 8 * H; i += kDecThreads) part_s[i] = 0.f;      // dh_T = 0 (summed at the top of the loop)
#if GSCAN_DEC_SB_QUAD == 2
    f32x2 dvv_acc = {0.f, 0.f}, dvt_acc = {0.f, 0.f};       // thread (feature pair tid >> 3, memories tid & 7, + 8, ...)
#elif GSCAN_DEC_SB_QUAD
    float dvv_acc = 0.f, dvt_acc = 0.f;                     // thread (feature tid >> 2, memories tid & 3, + 4, ...)
#else
    f32x2 dvv_acc = {0.f, 0.f}, dvt_acc = {0.f, 0.f};
#endif
    if (tid < H) { vt_s[tid] = vt_reg; vv_s[tid] = vv_reg; }
    if (tid < 64) {
        const float sum = wave_sum(aux_dy);
        datt_s[tid] = (has_aux && tid < M) ? aux_scale * (aux_dy - expf(aux_y) * sum) : 0.f;
    }
    if (GSCAN_STAMPS_ON && tid >= 64 && tid < 80) stamp_acc[tid - 64] = 0.f;
    staged_barrier();                                       // weights, memories and the small loads above are complete
    if (GSCAN_DEC_PRIO == 1) { if (__builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1); }
    if (GSCAN_DEC_PRIO == 2) { if (__builtin_amdgcn_readfirstlane(wave) < 4) __builtin_amdgcn_s_setprio(1); }
    GSCAN_STAMP_ONCE(11)

    for (int t = T - 1; t >= 0; --t) {
        GSCAN_STAMP(0)
        const unsigned bt = (unsigned)b * T + t;
        alpha_v_pf = alpha_v_nx;
        alpha_c_pf = alpha_c_nx;
        // consumed HERE (a step after their loads were issued), not at their first use in phases 3 / 6, where the
        // in-order vmcnt wait would also cover this iteration's prefetch and stores
        asm volatile("" : "+v"(alpha_v_pf), "+v"(alpha_c_pf));
        // ---- 1: dh_t = sum of the six partial products of step t+1; LSTM cell backward ------------
        if (tid < H) {
            float dh = pf[6];
#pragma unroll
            for (int sgi = 0; sgi < 6; ++sgi) dh += part_s[sgi * H + tid];
            const float ig = pf[0], fg = pf[1], gg = pf[2], og = pf[3];
            const float c = pf[4];
            const float c_prev = pf[5];
            const float tc = tanhf_(c);
            const float dct = dc + dh * og * (1.f - tc * tc);
            const float di = dct * gg * ig * (1.f - ig);
            const float df = dct * c_prev * fg * (1.f - fg);
            const float dg = dct * ig * (1.f - gg * gg);
            const float d_o = dh * tc * og * (1.f - og);
            dc = dct * fg;
            d_s[tid] = di; d_s[HP + tid] = df; d_s[2 * HP + tid] = dg; d_s[3 * HP + tid] = d_o;
            *reinterpret_cast<float4 *>(dperm_s + 4 * tid) = float4{di, df, dg, d_o};
            if constexpr (!kP1Late) {
                const unsigned dp = bt * 5 * uH + utid;      // rows of [delta (4H) | dzq (H)]
                a.delta[dp] = di; a.delta[dp + uH] = df; a.delta[dp + 2 * uH] = dg; a.delta[dp + 3 * uH] = d_o;
                // dqt of step t+1, kept in LDS by this thread since phase 7 of that step: stores go out right after
                // the wait for the prefetched activations above (vmcnt counts them: the next wait is a full step away)
                if (t + 1 < T) a.dqt[(bt + 1) * uH + utid] = d_s[4 * HP + tid];
            }
        } else if (tid >= 128 && tid < 128 + H) {
            // external gradients wrt the two contexts (output head) and the saved queries
            const int kk = tid - 128;
            exc_s[kk] = pf[0];
            exs_s[kk] = pf[1];
            qt_s[kk] = pf[2];
            qv_s[kk] = pf[3];
            if (COND) q2_s[kk] = pf[4];
        }
        if (t > 0) prefetch(t - 1);
        lds_barrier();
        GSCAN_STAMP(1)

        if constexpr (kP1Late) {
            if (tid >= 256 && tid < 256 + H) {
                const unsigned k = utid - 256, dp = bt * 5 * uH + k;
                const float4 dd = *reinterpret_cast<const float4 *>(dperm_s + 4 * k);      // (di, df, dg, do) of unit k
                a.delta[dp] = dd.x; a.delta[dp + uH] = dd.y; a.delta[dp + 2 * uH] = dd.z; a.delta[dp + 3 * uH] = dd.w;
                if (t + 1 < T) a.dqt[(bt + 1) * uH + k] = d_s[4 * HP + k];
            }
        }
        // ---- 2: d alpha_vis[m] = delta . U_vis[m] + dctx_vis(ext) . PK_vis[m] + d att_sum[m] ---
        dalpha_rows<H, HP, false, !UVL>(smem, d_s, dperm_s, exs_s, o.uv, o.pkv, o.pkv, M, datt_s, sc_s, wave, nwave, lane,
                                        a.u_v + (int64_t)b * M * 4 * H);
        lds_barrier();
        GSCAN_STAMP(2)
        // ---- 3: softmax backward in every wave's registers (lane m: ds_m = alpha_m (dalpha_m - sum alpha dalpha)),
        //         then through tanh(q + PK): wave w owns memories w, w+8, ...; a lane owns features lane, lane+64
        {
            const float al = (lane < M) ? alpha_v_pf : 0.f;
            const float da = (lane < M) ? sc_s[lane] : 0.f;
            const float dsm = al * (da - wave_sum(al * da));
#if GSCAN_DEC_SB_QUAD == 2
            const f32x2 dq = score_backward_octs<H, GSCAN_DEC_SBO_UV>(dsm, qv_s, vv_s, PKv, dPKv, M, dvv_acc, tid);
            if ((tid & 7) == 0 && 2 * (tid >> 3) < H) {      // lane 0 of the eight of feature pair k2 holds both sums
                const int k = 2 * (tid >> 3);
                *reinterpret_cast<f32x2 *>(dqv_s + k) = dq;
                *reinterpret_cast<f32x2 *>(a.dqv + bt * H + k) = dq;
                if (!COND) *reinterpret_cast<f32x2 *>(d_s + 5 * HP + k) = dq;   // visual query came straight from h
            }
#elif GSCAN_DEC_SB_QUAD
            const float dq = score_backward_quads<H>(dsm, qv_s, vv_s, PKv, dPKv, M, dvv_acc, tid);
            if ((tid & 3) == 0 && (tid >> 2) < H) {          // lane 0 of the quad of feature k holds d q_k
                const int k = tid >> 2;
                dqv_s[k] = dq;
                a.dqv[bt * H + k] = dq;
                if (!COND) d_s[5 * HP + k] = dq;             // visual query came straight from h
            }
#else
            score_backward<H>(dsm, qv_s, vv_s, PKv, dPKv, M, part_s, dvv_acc, wave, nwave, lane);
#endif
        }
        lds_barrier();
        GSCAN_STAMP(3)
#if !GSCAN_DEC_SB_QUAD
        if (tid < H) {
            float dq = 0.f;
#pragma unroll
            for (int cch = 0; cch < kDecThreads / 64; ++cch) dq += part_s[cch * H + tid];
            dqv_s[tid] = dq;
            a.dqv[bt * H + tid] = dq;
            if (!COND) d_s[5 * HP + tid] = dq;               // visual query came straight from h
        }
        lds_barrier();
#endif
        GSCAN_STAMP(4)

        // ---- 4: conditional query: dq2 = W_qv^T dqv, through tanh ------------------------------
        if (COND) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int r = s * kDecPairs + pair;
                if (r >= 6 * H && r < 7 * H) {
                    const float dq2 = pair_sum(half_dot<K0, kBwdDotChunk>(wt[s], dqv_s + half * K0));
                    const float q = q2_s[r - 6 * H];
                    const float dz = dq2 * (1.f - q * q);
                    if (half == 0) { d_s[5 * HP + r - 6 * H] = dz; a.delta[bt * 5 * H + 4 * H + r - 6 * H] = dz; }
                }
            }
            lds_barrier();
        }
        GSCAN_STAMP(5)

        // ---- 5: d alpha_text[m] = delta . U_text[m] + dzq . U2_text[m] + dctx_text(ext) . PK_text[m]
        dalpha_rows<H, HP, COND>(smem, d_s, dperm_s, exc_s, o.ut, o.pkt, o.u2t, len, nullptr, sc_s, wave, nwave, lane);
        lds_barrier();
        GSCAN_STAMP(6)
        // ---- 6: the same for the textual attention -----------------------------------------------
        {
            const float al = (lane < len) ? alpha_c_pf : 0.f;
            const float da = (lane < len) ? sc_s[lane] : 0.f;
            const float dsm = al * (da - wave_sum(al * da));
#if GSCAN_DEC_SB_QUAD == 2
            const f32x2 dq = score_backward_octs<H, GSCAN_DEC_SBO_UT>(dsm, qt_s, vt_s, PKt, dPKt, len, dvt_acc, tid);
            if ((tid & 7) == 0 && 2 * (tid >> 3) < H) *reinterpret_cast<f32x2 *>(d_s + 4 * HP + 2 * (tid >> 3)) = dq;
#elif GSCAN_DEC_SB_QUAD
            const float dq = score_backward_quads<H>(dsm, qt_s, vt_s, PKt, dPKt, len, dvt_acc, tid);
            if ((tid & 3) == 0 && (tid >> 2) < H) d_s[4 * HP + (tid >> 2)] = dq;
#else
            score_backward<H>(dsm, qt_s, vt_s, PKt, dPKt, len, part_s, dvt_acc, wave, nwave, lane);
#endif
        }
        lds_barrier();
        GSCAN_STAMP(7)
#if !GSCAN_DEC_SB_QUAD
        if (tid < H) {
            float dq = 0.f;
#pragma unroll
            for (int cch = 0; cch < kDecThreads / 64; ++cch) dq += part_s[cch * H + tid];
            d_s[4 * HP + tid] = dq;
        }
        lds_barrier();
#endif
        GSCAN_STAMP(8)

        // ---- 7: dh_{t-1} = [W_hh | W_qt | W_q2k_h or W_qv]^T . [delta | dqt | dzq or dqv] ------
        //         (six partial products per unit; they are summed at the top of the next iteration)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int r = s * kDecPairs + pair;
            if (r < 6 * H) {
                const int sg = r / H;
                const float part = pair_sum(half_dot<K0, kBwdDotChunk>(wt[s], d_s + sg * HP + half * K0));
                if (half == 0) part_s[r] = part;             // r = sg*H + k
            }
        }
        lds_barrier();
        GSCAN_STAMP(9)
    }

    // ---- epilogue: initial-state gradient through the bridge tanh, key and energy gradients ----
    if (tid < H) {
        a.dqt[row0 * uH + utid] = d_s[4 * HP + tid];     // step 0's dqt (deferred like the others)
        float dh = 0.f;
#pragma unroll
        for (int sgi = 0; sgi < 6; ++sgi) dh += part_s[sgi * H + tid];
        const float h0 = a.hprev[(int64_t)b * T * H + tid];
        a.dh0[(int64_t)b * H + tid] = (dh + dc) * (1.f - h0 * h0);   // h0 = c0 = tanh(.) (model.py:195)
    }
    for (int i = tid; i < M * H; i += kDecThreads) a.dpk_v[(int64_t)b * M * H + i] = dPKv[i];
    for (int i = tid; i < L * H; i += kDecThreads) a.dpk_t[(int64_t)b * L * H + i] = (i / H < len) ? dPKt[i] : 0.f;
    // energy-vector gradients of this ROW; the rows are added up by a leaf launch (head_grad_finish).  256 workgroups
    // adding into the same 2H addresses with atomics kept this kernel's last writes — and the whole critical chain behind
    // it — waiting ~9 us (profiles/r03_c_*).
#if GSCAN_DEC_SB_QUAD == 2
    {   // the eight lanes of a group hold the sums over their memories of the feature pair tid >> 3
        f32x2 xv = {quad_sum(dvv_acc.x), quad_sum(dvv_acc.y)}, xt = {quad_sum(dvt_acc.x), quad_sum(dvt_acc.y)};
        xv.x += dpp_move<0x141, 0xf>(xv.x); xv.y += dpp_move<0x141, 0xf>(xv.y);
        xt.x += dpp_move<0x141, 0xf>(xt.x); xt.y += dpp_move<0x141, 0xf>(xt.y);
        if ((tid & 7) == 0 && 2 * (tid >> 3) < H) {
            *reinterpret_cast<f32x2 *>(a.dv_v + (int64_t)b * H + 2 * (tid >> 3)) = xv;
            *reinterpret_cast<f32x2 *>(a.dv_t + (int64_t)b * H + 2 * (tid >> 3)) = xt;
        }
    }
#elif GSCAN_DEC_SB_QUAD
    {   // the four lanes of a quad hold the sums over their memories of feature tid >> 2
        const float xv = quad_sum(dvv_acc), xt = quad_sum(dvt_acc);
        if ((tid & 3) == 0 && (tid >> 2) < H) {
            a.dv_v[(int64_t)b * H + (tid >> 2)] = xv;
            a.dv_t[(int64_t)b * H + (tid >> 2)] = xt;
        }
    }
#else
    // (every wave holds partial sums for features (lane, lane+64))
    lds_barrier();
    if (lane < H) part_s[wave * H + lane] = dvv_acc.x;
    if (lane + 64 < H) part_s[wave * H + lane + 64] = dvv_acc.y;
    lds_barrier();
    if (tid < H) {
        float x = 0.f;
        for (int cch = 0; cch < kDecThreads / 64; ++cch) x += part_s[cch * H + tid];
        a.dv_v[(int64_t)b * H + tid] = x;
    }
    lds_barrier();
    if (lane < H) part_s[wave * H + lane] = dvt_acc.x;
    if (lane + 64 < H) part_s[wave * H + lane + 64] = dvt_acc.y;
    lds_barrier();
    if (tid < H) {
        float x = 0.f;
        for (int cch = 0; cch < kDecThreads / 64; ++cch) x += part_s[cch * H + tid];
        a.dv_t[(int64_t)b * H + tid] = x;
    }
#endif
    if (GSCAN_STAMPS_ON && a.stamps && blockIdx.x == 0 && tid < 10) a.stamps[tid] = stamp_acc[tid];
    GSCAN_STAMP_ONCE(12)
}

template <int H, bool COND, bool UVL = true>
__global__ __launch_bounds__(kDecThreads) void decoder_bwd_kernel(DecoderArgs a) {
    TraceScope trace_scope(TK_DECODER_BWD);
    decoder_bwd_body<H, COND, UVL>(a);
}

// Register images of the decoder weights are written once per step by the step prologue kernel
// (decoder_image_element in step.h; layout described there).

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
constexpr size_t kLdsLimit = 160 * 1024;

constexpr int kStreamMinHidden = 84;   // hidden sizes from here on have a U_vis-streaming instantiation

template <int H, bool COND>
static int launch_decoder(bool backward, int B, const DecoderArgs &a, hipStream_t stream) {
    const bool greedy = !backward && a.tokens_out != nullptr;
    const size_t extra = greedy ? (size_t)a.V * 4 * H + (size_t)a.V * a.V + H + 64 + 16 : 0;
    // the row's memories stay in LDS for all T steps; when they do not fit, the largest block (the gate images of
    // the visual memories) stays in global memory and is streamed from L2 by the two phases that read it
    DecoderLds o = decoder_lds(H, a.L, a.M, a.V, COND, backward, true);
    const bool uvl = ((size_t)o.total + extra) * sizeof(float) <= kLdsLimit || H < kStreamMinHidden;
    if (!uvl) o = decoder_lds(H, a.L, a.M, a.V, COND, backward, false);
    const size_t bytes = ((size_t)o.total + extra) * sizeof(float);
    GSCAN_CHECK(bytes <= kLdsLimit,
                "decoder: a row's memories need %zu bytes of LDS (> 160 KiB)%s: grid cells=%d command length=%d hidden=%d",
                bytes, uvl ? "" : " even with the visual gate images left in global memory", a.M, a.L, H);
    GSCAN_CHECK(a.w_image != nullptr, "decoder: weight image missing");
    // Algorithmic MACs of one decoder step that this kernel owns (SURVEY.md §8d MAC_step minus the
    // embedding part of the LSTM input and the output head, which run as GEMMs outside the loop):
    // query projections, both score/context reductions, and the [ctx_text|ctx_vis|h] part of the LSTM.
    const double macs = (double)H * H + 2.0 * a.L * H + (COND ? 2.0 * H * H : 0.0) + (double)H * H +
                        2.0 * a.M * H + 4.0 * H * 3.0 * H + 4.0 * H * H + (double)H * a.V;   // + the fused head
    // What the kernel EXECUTES per step is about half of that: the context part of the LSTM input product is a weighted
    // sum of the gate images (L + M rows of 4H instead of 4H x 2H), the head is one [V, 4H] matrix instead of H x 4H +
    // V x H.  Forward: 6 H^2 dots (+ H^2 conditional), scores L H + M H, column sums of [U | PK (| U2)], head;
    // backward: the same matrices transposed, d alpha rows, two passes over the score terms.
    const double hh = (double)H * H, lh = (double)a.L * H, mh = (double)a.M * H;
    const double exec_macs = backward ? (COND ? 7.0 : 6.0) * hh + 7.0 * mh + (COND ? 8.0 : 7.0) * lh + 4.0 * H * a.V
                                      : (COND ? 7.0 : 6.0) * hh + 6.0 * mh + (COND ? 7.0 : 6.0) * lh + 4.0 * H * a.V;
    ProbeScope probe(backward ? P_DECODER_BWD : P_DECODER_FWD, stream, 2.0 * exec_macs * B * a.T, 2.0 * macs * B * a.T);
    auto launch = [&](auto kernel, const char *name) -> int {
        static bool attr_set = false;                          // one flag per kernel instantiation (generic lambda)
        if (!attr_set) {
            GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
            attr_set = true;
        }
        hipLaunchKernelGGL(kernel, dim3(B), dim3(kDecThreads), bytes, stream, a);
        GSCAN_LAUNCHED(name);
        return 0;
    };
    if (greedy) GSCAN_CHECK(a.V <= 64 && a.head_wc && a.dec_emb && a.steps_out, "greedy decoder: missing tables or V > 64");
    if constexpr (H >= kStreamMinHidden) {
        if (!uvl) {
            if (backward) return launch(decoder_bwd_kernel<H, COND, false>, "decoder_bwd_kernel");
            if (greedy) return launch(decoder_fwd_kernel<H, COND, true, false>, "decoder_fwd_kernel");
            return launch(decoder_fwd_kernel<H, COND, false, false>, "decoder_fwd_kernel");
        }
    }
    if (backward) return launch(decoder_bwd_kernel<H, COND, true>, "decoder_bwd_kernel");
    if (greedy) return launch(decoder_fwd_kernel<H, COND, true, true>, "decoder_fwd_kernel");
    return launch(decoder_fwd_kernel<H, COND, false, true>, "decoder_fwd_kernel");
}


// the sizes of this translation unit's part
#define GSCAN_DEC_CAT_(a, b) a##b
#define GSCAN_DEC_CAT(a, b) GSCAN_DEC_CAT_(a, b)
#define GSCAN_DEC_THIS_PART(X) GSCAN_DEC_CAT(GSCAN_DEC_HIDDEN_PART, GSCAN_DEC_PART)(X)
int GSCAN_DEC_CAT(decoder_run_part, GSCAN_DEC_PART)(bool backward, int B, int H, bool cond, const DecoderArgs &a, hipStream_t stream) {
    switch (H) {
#define X(n) case n: return cond ? launch_decoder<n, true>(backward, B, a, stream) : launch_decoder<n, false>(backward, B, a, stream);
        GSCAN_DEC_THIS_PART(X)
#undef X
        default: break;
    }
    return -1;       // not a size of this part
}

#if GSCAN_DEC_PART == 0
int decoder_run_part1(bool backward, int B, int H, bool cond, const DecoderArgs &a, hipStream_t stream);
int decoder_run_part2(bool backward, int B, int H, bool cond, const DecoderArgs &a, hipStream_t stream);
int decoder_run_part3(bool backward, int B, int H, bool cond, const DecoderArgs &a, hipStream_t stream);

bool decoder_hidden_supported(int h) {
#define X(n) if (h == n) return true;
    GSCAN_DEC_HIDDEN_SIZES(X)
#undef X
    return false;
}

// bytes of LDS a row's workgroup needs
size_t decoder_lds_bytes(int H, int L, int M, int V, bool cond, bool backward) {
    const size_t full = (size_t)decoder_lds(H, L, M, V, cond, backward, true).total * sizeof(float);
    if (full <= kLdsLimit || H < kStreamMinHidden) return full;
    return (size_t)decoder_lds(H, L, M, V, cond, backward, false).total * sizeof(float);
}

// The kernels of this file hold a row's memories in LDS and its attention distributions in the 64 lanes of a wave, and
// are compiled for the hidden sizes whose recurrent weights fit the register file; every other shape runs on
// decoder_any.hip's kernels (GSCAN_DECODER_ANY=1: every shape does, for tests).
bool decoder_fast_supported(int H, int L, int M, int V, bool cond) {
    static const int force_any = [] { const char *e = getenv("GSCAN_DECODER_ANY"); return e ? atoi(e) : 0; }();
    if (force_any || !decoder_hidden_supported(H) || L > 64 || M > 64 || V > kHeadWcRows) return false;   // V: one MFMA tile of logits
    // greedy decoding keeps the whole head and a [V, V] table behind the memories: it must fit too, if need be with the
    // visual gate images streamed (a vocabulary of 64 at H = 96 did not, and the launch failed instead of taking the
    // streaming kernels: found by the planner sanitizer run, tests/planner)
    const size_t greedy_extra = ((size_t)V * 4 * H + (size_t)V * V + H + 64 + 16) * sizeof(float);
    const size_t fwd_least = (size_t)decoder_lds(H, L, M, V, cond, false, H < kStreamMinHidden).total * sizeof(float);
    return decoder_lds_bytes(H, L, M, V, cond, true) <= kLdsLimit && decoder_lds_bytes(H, L, M, V, cond, false) <= kLdsLimit &&
           fwd_least + greedy_extra <= kLdsLimit;
}

int decoder_run(bool backward, int B, int H, bool cond, const DecoderArgs &a, hipStream_t stream) {
    GSCAN_CHECK(B > 0 && a.T > 0 && a.L > 0 && a.M > 0, "decoder: bad dims B=%d T=%d L=%d M=%d", B, a.T, a.L, a.M);
    if (!decoder_fast_supported(H, a.L, a.M, a.V, cond)) return decoder_run_any(backward, B, H, cond, a, stream);
    int rc = decoder_run_part0(backward, B, H, cond, a, stream);
    if (rc < 0) rc = decoder_run_part1(backward, B, H, cond, a, stream);
    if (rc < 0) rc = decoder_run_part2(backward, B, H, cond, a, stream);
    if (rc < 0) rc = decoder_run_part3(backward, B, H, cond, a, stream);
    if (rc >= 0) return rc;
    GSCAN_CHECK(false, "decoder_hidden_size %d has no compiled kernel (supported: " GSCAN_DEC_HIDDEN_LIST ")", H);
}

GSCAN_TRACE_TU(decoder)
#else
// parts 1-3: their own trace pointer (a per-translation-unit device variable), set with part 0's
#define GSCAN_DEC_TRACE_NAME GSCAN_DEC_CAT(decoder_p, GSCAN_DEC_PART)
int GSCAN_DEC_CAT(trace_set_decoder_p, GSCAN_DEC_PART)(unsigned long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

}  // namespace gscan
