// Command encoder recurrence (seq2seq/seq2seq_model.py:62-88), forward and backward.
//
// The reference sorts rows by length, packs them and calls nn.LSTM.  Rows are independent,
// so here one workgroup owns one command (and direction) for its whole length: the recurrent
// matrix W_hh (4He x He floats) lives in the VGPRs of the workgroup for all time steps (a thread
// holds two rows), h is broadcast from LDS, and the command's projections and saved activations stay in LDS for
// the whole loop (see "LDS residency" below).  Length
// masking replaces packing: a row simply stops after its own length (the reverse direction
// starts at its last real token), outputs at padded positions stay zero.
//
// Backward keeps the transposed ownership: thread (s, k) holds W_hh[s*He .. s*He+He-1][k], so
// dh_{t-1}[k] = sum_j W_hh[j][k] * delta[j] is four in-register partial dots plus one LDS sum.
// The kernel emits only the gate pre-activation gradients delta[b,t,dir,4He]; all weight
// gradients are dense GEMMs over those afterwards.
#include "anyshape.h"

namespace gscan {

// R rows of weights against one LDS vector: the vector is read once (same address in every lane: LDS broadcast,
// 8 cycles of LDS bandwidth per 16-byte wave read whatever the lanes hold) and feeds R dot products.  The
// recurrences are bound by exactly these reads, so a thread keeps TWO weight rows (R = 2) when the register file
// allows (He <= 100): half as many waves read the same vector.
template <int HE, int R>
__device__ __forceinline__ void dots_lds(const float (&w)[R][HE], const float *v, float (&out)[R]) {
    static_assert(HE % 4 == 0, "hidden size must be a multiple of 4");
    // packed FMAs (v_pk_fma_f32: two lanes of a register pair per instruction): with two workgroups per CU the
    // recurrence's step was bound by the issue of these FMAs (2 waves x 200 scalar FMAs x 4 cycles of the ~2 000 per step)
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    f32x2 a01[R], a23[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { a01[r] = f32x2{0.f, 0.f}; a23[r] = f32x2{0.f, 0.f}; }
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
#pragma unroll
    for (int i = 0; i < HE / 4; ++i) {
        const float4 x = v4[i];
#pragma unroll
        for (int r = 0; r < R; ++r) {              // R * 2 independent chains: enough to cover the FMA latency
            a01[r] += f32x2{w[r][4 * i + 0], w[r][4 * i + 1]} * f32x2{x.x, x.y};
            a23[r] += f32x2{w[r][4 * i + 2], w[r][4 * i + 3]} * f32x2{x.z, x.w};
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = (a01[r][0] + a23[r][0]) + (a01[r][1] + a23[r][1]);
}

#ifdef GSCAN_ENC_STAMPS   // experiment build: cycle stamps of workgroup (0,0), in the tail of the trace buffer
#define EST(i) if (g_trace_buf && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { const long long n_ = clock64(); g_trace_buf[1520 + i] += (unsigned long long)(n_ - est_prev); est_prev = n_; }
#else
#define EST(i)
#endif

template <int HE> struct EncShape {
    static constexpr int R = HE <= 100 ? 2 : 1;                       // weight rows (forward) / columns (backward) per thread
    static constexpr int kThreads = ((4 * HE / R + 63) / 64) * 64;
};

// LDS residency.  A wait for a prefetched load (s_waitcnt vmcnt) also waits for every store the wave issued
// before it, and under divergent control flow the compiler can only emit the full drain vmcnt(0) — with global
// loads and stores inside the time loop each step paid a store acknowledgement.  A command is short (its whole
// history is 6 He floats per step), so both recurrences keep it in LDS: everything a (row, direction) reads is
// staged before the loop with 16-byte coalesced loads, results are written in place, and one coalesced write-back
// follows the loop.  There is no global memory operation inside either time loop.
//   forward   g_s [len][4He] input projections + b_hh  ->  gate activations (in place)
//             c_s [len][He] cells, h_s [len][He] outputs, z_s [He] the zero initial state
//   backward  d_s [len][4He] gate activations  ->  gate pre-activation gradients (in place)
//             c_s [len][He] cells, o_s [len][He] gradient wrt the summed outputs, part_s [4He] partial dh
constexpr size_t kEncLdsLimit = 160 * 1024;
inline size_t encoder_fwd_lds(int L, int HE, int E = 0) { return ((size_t)L * 6 * HE + HE + (size_t)((L + 15) & ~15) * (E ? E + 1 : 0)) * sizeof(float); }
inline size_t encoder_bwd_lds(int L, int HE) { return ((size_t)L * 6 * HE + 4 * HE) * sizeof(float); }

// grid (B, D): the two directions of a row run as two workgroups (they only meet in the sums below).
// The LAST encoder layer passes `out` / `h_final` (direction sums), a layer below it passes `hcat` instead: its h per
// direction, [B, L, D*He] = the next layer's input (nn.LSTM concatenates the directions), times `hcat_mask` (the
// inter-layer dropout, or NULL), zero at padded positions.
// `out` and `h_final` must be zero on entry: each direction ADDS its h (0 + h_f + h_r in either order is the same
// float: two-operand addition commutes), which is how the directions are summed (seq2seq_model.py:77-81).
// Thread j < 4He/R owns gate rows j + r*(4He/R), r < R  (R = 2: [i | f] rows and the matching [g | o] rows).
template <int HE>
__global__ __launch_bounds__(EncShape<HE>::kThreads, 2) void encoder_lstm_fwd_kernel(int L, int D, const float *__restrict__ gx,
                                        const int32_t *__restrict__ lengths, const float *__restrict__ b_hh_f,
                                        const float *__restrict__ b_hh_r, float *__restrict__ out,
                                        float *__restrict__ h_final, float *__restrict__ gates,
                                        float *__restrict__ cells, float *__restrict__ hprev,
                                        const float *__restrict__ w_image, float *__restrict__ hcat,
                                        const float *__restrict__ hcat_mask, EncInput in) {
    TraceScope trace_scope(TK_ENCODER_FWD);
    constexpr int R = EncShape<HE>::R, NT = 4 * HE / R;              // owning threads
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x, nthr = blockDim.x;
    int len = lengths[b];
    len = max(0, min(len, L));
    float *g_s = lds, *c_s = g_s + len * 4 * HE, *h_s = c_s + len * HE, *z_s = h_s + len * HE;
    const bool is_gate = j < NT, is_unit = j < HE;
    const int64_t row0 = (int64_t)b * L;                             // row of (b, t, dir) = (row0 + t) * D + dir
#ifdef GSCAN_ENC_STAMPS
    long long est_prev = clock64();
#endif

    if (in.x) {
        // the first layer projects its own input: g_s[t][row] = W_ih[row] . x[b,t] + b_ih[row] + b_hh[row]
        // (seq2seq_model.py:70,74).  x = the embedded, dropped-out command (E floats per token, staged in LDS); a
        // thread computes the projections of the gate rows it owns, so nothing crosses threads but x — and the
        // launch that used to compute them for the whole batch is off the critical chain of the step.
        const int E = in.E;
        float *x_s = z_s + HE;                                       // [E][LP] time-contiguous, LP = len rounded up to 16
        const int LP = (len + 15) & ~15;
        for (int idx = j; idx < LP * (E + 1); idx += nthr) {         // row E: ones (the bias column of the image)
            const int e = idx / LP, t = idx - e * LP;
            x_s[idx] = t < len ? (e < E ? in.x[(row0 + t) * E + e] : 1.f) : 0.f;
        }
        lds_barrier();
        EST(0)
        {
            // [4He rows] x [len steps] x [E]: a tile of 16 gate rows x 16 time steps per wave on the matrix cores, A
            // fragments from the column-major image of W_ih the prologue writes (a lane group reads 16 consecutive
            // rows: 64-byte pieces), B fragments = x from LDS.  As per-thread FMAs (a thread per gate row, x broadcast
            // from LDS) this phase took 12-13 000 of the kernel's 45 000 cycles whichever way its loads were arranged
            // (tools/encoder_stamps.py): 800 FMAs and 100 broadcast b128 reads per thread, two workgroups per CU.
            using f32x4 = __attribute__((ext_vector_type(4))) float;
            const int EK = E + 1;                                    // K of the product: E inputs and the bias column
            const float *wt = in.w_ih_t + (int64_t)dir * EK * 4 * HE;
            const int lane = j & 63, wave = j >> 6, fr = lane & 15, fg = lane >> 4;
            constexpr int NTILE = 4 * HE / 16, SU = 8, NW = EncShape<HE>::kThreads / 64, TPW = (NTILE + NW - 1) / NW;
            static_assert(HE % 4 == 0, "gate rows come in whole tiles of 16");
            // every global load of a wave's tiles (A fragments of up to 32 columns) is issued before the first MFMA: a
            // tile at a time, each tile waited out its own L2 round trip (~1 200 cycles)
            for (int t0 = 0; t0 < len; t0 += 16) {
                f32x4 c[TPW];
#pragma unroll
                for (int i = 0; i < TPW; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int s0 = 0; 4 * s0 < EK; s0 += SU) {
                    float av[TPW][SU], bv[SU];
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int k = 4 * (s0 + u) + fg, kc = min(k, EK - 1);
                        bv[u] = (k < EK) ? x_s[kc * LP + t0 + fr] : 0.f;         // x_s is zero at steps >= len
#pragma unroll
                        for (int i = 0; i < TPW; ++i)
                            av[i][u] = wt[(int64_t)kc * 4 * HE + 16 * min(wave + i * NW, NTILE - 1) + fr];
                    }
#pragma unroll
                    for (int u = 0; u < SU; ++u)
#pragma unroll
                        for (int i = 0; i < TPW; ++i)
                            c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][u], bv[u], c[i], 0, 0, 0);
                }
                const int t = t0 + fr;                               // C: column = time step, rows 4 fg + r of the tile
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int tile = wave + i * NW;
                    if (t < len && tile < NTILE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) g_s[t * 4 * HE + 16 * tile + 4 * fg + r] = c[i][r];
                    }
                }
            }
        }
    } else {
        // stage the input projections of the whole command, recurrent bias added (seq2seq_model.py:74)
        const float *bias = dir ? b_hh_r : b_hh_f;                   // a parameter: no 16-byte alignment promised
        for (int idx = j; idx < len * HE; idx += nthr) {
            const int t = idx / HE, q = idx - t * HE;
            const float4 x = *reinterpret_cast<const float4 *>(gx + ((row0 + t) * D + dir) * 4 * HE + 4 * q);
            const float4 bb = {bias[4 * q], bias[4 * q + 1], bias[4 * q + 2], bias[4 * q + 3]};
            *reinterpret_cast<float4 *>(g_s + t * 4 * HE + 4 * q) = float4{x.x + bb.x, x.y + bb.y, x.z + bb.z, x.w + bb.w};
        }
    }
    EST(1)
    // padded positions: zero saved h_prev (it multiplies delta = 0 in a GEMM later); out stays zero
    for (int idx = j; idx < (L - len) * HE; idx += nthr) {
        const int t = len + idx / HE, k = idx % HE;
        hprev[((row0 + t) * D + dir) * HE + k] = 0.f;
        if (hcat) hcat[((row0 + t) * D + dir) * HE + k] = 0.f;
    }
    // register image [dir][r][k][thread]: consecutive lanes read consecutive floats (a row per lane straight from
    // W_hh would touch 64 cache lines per load)
    float w[R][HE];
    if (is_gate) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float *img = w_image + ((int64_t)(dir * R + r) * HE) * NT + j;
#pragma unroll
            for (int k = 0; k < HE; ++k) w[r][k] = img[k * NT];
        }
    }
    if (is_unit) z_s[j] = 0.f;
    float c = 0.f;
    const float *h_prev = z_s;
    lds_barrier();
    EST(2)
    for (int s = 0; s < len; ++s) {
        const int t = dir ? (len - 1 - s) : s;
        float *g = g_s + t * 4 * HE;
        if (is_gate) {
            float dot[R];
            dots_lds<HE, R>(w, h_prev, dot);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int gr = j + r * NT;                            // gate row: [i | f | g | o] blocks of He
                const float pre = g[gr] + dot[r];
                g[gr] = (gr >= 2 * HE && gr < 3 * HE) ? tanhf_(pre) : sigmoidf_(pre);
            }
        }
        lds_barrier();
        if (is_unit) {
            const float ig = g[j], fg = g[HE + j], gg = g[2 * HE + j], og = g[3 * HE + j];
            c = fg * c + ig * gg;
            c_s[t * HE + j] = c;
            h_s[t * HE + j] = og * tanhf_(c);
        }
        h_prev = h_s + t * HE;
        lds_barrier();
    }
    EST(3)
    // write-back: saved activations for the backward pass, the direction sums (atomics: two addends commute)
    for (int idx = j; idx < len * HE; idx += nthr) {
        const int t = idx / HE, q = idx - t * HE;
        *reinterpret_cast<float4 *>(gates + ((row0 + t) * D + dir) * 4 * HE + 4 * q) =
            *reinterpret_cast<const float4 *>(g_s + t * 4 * HE + 4 * q);
    }
    for (int idx = j; idx < len * HE; idx += nthr) {
        const int t = idx / HE, k = idx - t * HE;
        const int64_t row = (row0 + t) * D + dir;
        const bool first = dir ? (t == len - 1) : (t == 0);
        cells[row * HE + k] = c_s[idx];
        hprev[row * HE + k] = first ? 0.f : h_s[(dir ? t + 1 : t - 1) * HE + k];
        if (out) atomicAdd(out + (row0 + t) * HE + k, h_s[idx]);
        if (hcat) hcat[row * HE + k] = hcat_mask ? h_s[idx] * hcat_mask[row * HE + k] : h_s[idx];
    }
    if (h_final && is_unit && len > 0) atomicAdd(h_final + (int64_t)b * HE + j, h_s[(dir ? 0 : len - 1) * HE + j]);
    EST(4)
}

// Backward: thread (seg, q) owns columns q + r*(He/R), r < R, of block seg of W_hh (its R dot products share the
// delta segment they read); dh_{t-1}[k] = sum of the four blocks' partial products, taken by unit k at the top of
// the next step.
template <int HE>
__global__ __launch_bounds__(EncShape<HE>::kThreads, 2) void encoder_lstm_bwd_kernel(int L, int D, const int32_t *__restrict__ lengths,
                                        const float *__restrict__ w_hh_f, const float *__restrict__ w_hh_r,
                                        const float *__restrict__ gates, const float *__restrict__ cells,
                                        const float *__restrict__ d_out, const float *__restrict__ d_h_final,
                                        float *__restrict__ delta, int d_out_row, int d_out_dir,
                                        const float *__restrict__ d_out_mask) {
    TraceScope trace_scope(TK_ENCODER_BWD);
    constexpr int R = EncShape<HE>::R, KQ = HE / R, NT = 4 * KQ;
    static_assert(HE % R == 0, "hidden size must divide by the columns per thread");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
    int len = lengths[b];
    len = max(0, min(len, L));
    float *d_s = lds, *c_s = d_s + len * 4 * HE, *o_s = c_s + len * HE, *part_s = o_s + len * HE;
    const bool active = tid < NT, is_unit = tid < HE;
    const int seg = tid / KQ, q = tid % KQ;
    const int64_t row0 = (int64_t)b * L;
    for (int idx = tid; idx < len * HE; idx += nthr) {                // saved activations of the whole command
        const int t = idx / HE, k4 = idx - t * HE;
        *reinterpret_cast<float4 *>(d_s + t * 4 * HE + 4 * k4) =
            *reinterpret_cast<const float4 *>(gates + ((row0 + t) * D + dir) * 4 * HE + 4 * k4);
    }
    for (int idx = tid; idx < len * HE / 4; idx += nthr) {
        const int t = idx / (HE / 4), k4 = idx - t * (HE / 4);
        *reinterpret_cast<float4 *>(c_s + t * HE + 4 * k4) =
            *reinterpret_cast<const float4 *>(cells + ((row0 + t) * D + dir) * HE + 4 * k4);
        // gradient wrt this direction's h_t: the shared sum [B,L,He] for the last layer (row stride He, direction
        // offset 0), the direction's half of d(next layer's input) [B,L,D*He] times the dropout mask below it
        const int64_t at = (row0 + t) * d_out_row + dir * d_out_dir + 4 * k4;
        float4 g4 = *reinterpret_cast<const float4 *>(d_out + at);
        if (d_out_mask) {
            const float4 m4 = *reinterpret_cast<const float4 *>(d_out_mask + at);
            g4 = float4{g4.x * m4.x, g4.y * m4.y, g4.z * m4.z, g4.w * m4.w};
        }
        *reinterpret_cast<float4 *>(o_s + t * HE + 4 * k4) = g4;
    }
    for (int i = tid; i < 4 * HE; i += nthr)
        part_s[i] = (i < HE && d_h_final) ? d_h_final[(int64_t)b * HE + i] : 0.f;
    for (int idx = tid; idx < (L - len) * HE; idx += nthr) {          // padded positions get delta = 0
        const int t = len + idx / HE, k4 = idx % HE;
        *reinterpret_cast<float4 *>(delta + ((row0 + t) * D + dir) * 4 * HE + 4 * k4) = float4{0.f, 0.f, 0.f, 0.f};
    }
    const float *w_hh = dir ? w_hh_r : w_hh_f;
    float wt[R][HE];
    if (active) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int jj = 0; jj < HE; ++jj) wt[r][jj] = w_hh[(int64_t)(seg * HE + jj) * HE + q + r * KQ];
    }
    float dc = 0.f;
    lds_barrier();
    for (int s = len - 1; s >= 0; --s) {
        const int t = dir ? (len - 1 - s) : s;
        float *g = d_s + t * 4 * HE;
        if (is_unit) {
            const float dh = ((part_s[tid] + part_s[HE + tid]) + (part_s[2 * HE + tid] + part_s[3 * HE + tid])) +
                             o_s[t * HE + tid];
            const float ig = g[tid], fg = g[HE + tid], gg = g[2 * HE + tid], og = g[3 * HE + tid];
            const float c = c_s[t * HE + tid];
            const float c_prev = (s > 0) ? c_s[(dir ? t + 1 : t - 1) * HE + tid] : 0.f;
            const float tc = tanhf_(c);
            const float dct = dc + dh * og * (1.f - tc * tc);
            dc = dct * fg;
            g[tid] = dct * gg * ig * (1.f - ig);
            g[HE + tid] = dct * c_prev * fg * (1.f - fg);
            g[2 * HE + tid] = dct * ig * (1.f - gg * gg);
            g[3 * HE + tid] = dh * tc * og * (1.f - og);
        }
        lds_barrier();
        if (active) {
            float part[R];
            dots_lds<HE, R>(wt, g + seg * HE, part);
#pragma unroll
            for (int r = 0; r < R; ++r) part_s[seg * HE + q + r * KQ] = part[r];
        }
        lds_barrier();
    }
    for (int idx = tid; idx < len * HE; idx += nthr) {
        const int t = idx / HE, k4 = idx - t * HE;
        *reinterpret_cast<float4 *>(delta + ((row0 + t) * D + dir) * 4 * HE + 4 * k4) =
            *reinterpret_cast<const float4 *>(d_s + t * 4 * HE + 4 * k4);
    }
}

// ------------------------------------------------------------------------------------------
// The same two recurrences for ANY encoder hidden size and command length (the kernels above keep W_hh in registers —
// hidden sizes up to 128 — and the whole command's history in LDS — L . He <= ~6 800): W_hh (and, for the first layer,
// W_ih) streamed from L2 in the reference's layout, one step's vectors in LDS, saved activations written straight to
// global memory.  grid (B, D), 1024 threads, same arguments and outputs.
// ------------------------------------------------------------------------------------------
template <bool V4>
__global__ __launch_bounds__(kAnyThreads) void encoder_lstm_fwd_any_kernel(int L, int D, int HE, const float *__restrict__ gx,
                                        const int32_t *__restrict__ lengths, const float *__restrict__ w_hh_f,
                                        const float *__restrict__ b_hh_f, const float *__restrict__ w_hh_r,
                                        const float *__restrict__ b_hh_r, float *__restrict__ out,
                                        float *__restrict__ h_final, float *__restrict__ gates,
                                        float *__restrict__ cells, float *__restrict__ hprev, float *__restrict__ hcat,
                                        const float *__restrict__ hcat_mask, EncInput in) {
    TraceScope trace_scope(TK_ENCODER_FWD);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    int len = lengths[b];
    len = max(0, min(len, L));
    const int HP = (HE + 3) / 4 * 4;
    float *h_s = lds, *pre_s = h_s + HP, *c_s = pre_s + 4 * HP, *x_s = c_s + HP;      // x_s [4 HP]: staged x, then gx of a step
    const int64_t row0 = (int64_t)b * L;
    const float *w_hh = dir ? w_hh_r : w_hh_f, *b_hh = dir ? b_hh_r : b_hh_f;
    const float *w_ih = dir ? in.w_ih_r : in.w_ih_f, *b_ih = dir ? in.b_ih_r : in.b_ih_f;
    for (int k = tid; k < HE; k += kAnyThreads) { h_s[k] = 0.f; c_s[k] = 0.f; }
    for (int idx = tid; idx < (L - len) * HE; idx += kAnyThreads) {       // padded positions: zero saved h_prev; out stays zero
        const int t = len + idx / HE, k = idx % HE;
        hprev[((row0 + t) * D + dir) * HE + k] = 0.f;
        if (hcat) hcat[((row0 + t) * D + dir) * HE + k] = 0.f;
    }
    __syncthreads();
    // ---- the first layer's input projections of the whole command, before the recurrence: they do not depend on h, so
    //      nothing orders the steps' products (no barrier between them: many steps' loads in flight).  They go to the
    //      row's slots of `gates`, where the recurrence reads them back and leaves the activated gates.
    float *gx_s = x_s;                                        // recurrence: this step's input projection (4 HP floats)
    if (in.x) {
        const int E = in.E, chunk = max(1, (4 * HP) / E);     // the command's x, `chunk` steps at a time, in the gx_s region
        for (int t0 = 0; t0 < len; t0 += chunk) {
            const int nt = min(chunk, len - t0);
            for (int i = tid; i < nt * E; i += kAnyThreads) x_s[i] = in.x[(row0 + t0) * E + i];
            __syncthreads();
            for (int tt = 0; tt < nt; ++tt) {
                float *dst = gates + ((row0 + t0 + tt) * D + dir) * 4 * HE;
                // E is small and W_ih rows need not be 16-byte aligned (E = 25): scalar loads
                matvec_rows<false>(w_ih, E, 4 * HE, E, x_s + tt * E, [&](int r, float v) { dst[r] = v + b_ih[r] + b_hh[r]; });
            }
            __syncthreads();
        }
    }
    const float *pre_g = in.x ? gates : gx;                   // [B, L, D, 4 HE] either way
    for (int s = 0; s < len; ++s) {
        const int t = dir ? (len - 1 - s) : s;
        const int64_t row = (row0 + t) * D + dir;
        for (int k = tid; k < HE; k += kAnyThreads) hprev[row * HE + k] = h_s[k];     // h entering this step (0 at the first)
        for (int r = tid; r < 4 * HE; r += kAnyThreads) gx_s[r] = pre_g[row * 4 * HE + r] + (in.x ? 0.f : b_hh[r]);
        matvec_rows<V4>(w_hh, HE, 4 * HE, HE, h_s, [&](int r, float v) { pre_s[r] = v; });
        __syncthreads();
        for (int k = tid; k < HE; k += kAnyThreads) {
            const float ig = sigmoidf_(pre_s[k] + gx_s[k]), fg = sigmoidf_(pre_s[HE + k] + gx_s[HE + k]),
                        gg = tanhf_(pre_s[2 * HE + k] + gx_s[2 * HE + k]), og = sigmoidf_(pre_s[3 * HE + k] + gx_s[3 * HE + k]);
            const float c = fg * c_s[k] + ig * gg, h = og * tanhf_(c);
            c_s[k] = c;
            h_s[k] = h;
            gates[row * 4 * HE + k] = ig; gates[row * 4 * HE + HE + k] = fg;
            gates[row * 4 * HE + 2 * HE + k] = gg; gates[row * 4 * HE + 3 * HE + k] = og;
            cells[row * HE + k] = c;
            if (out) atomicAdd(out + (row0 + t) * HE + k, h);
            if (hcat) hcat[row * HE + k] = hcat_mask ? h * hcat_mask[row * HE + k] : h;
        }
        __syncthreads();
    }
    if (h_final && len > 0)
        for (int k = tid; k < HE; k += kAnyThreads) atomicAdd(h_final + (int64_t)b * HE + k, h_s[k]);
}

__global__ __launch_bounds__(kAnyThreads) void encoder_lstm_bwd_any_kernel(int L, int D, int HE, const int32_t *__restrict__ lengths,
                                        const float *__restrict__ w_hh_f, const float *__restrict__ w_hh_r,
                                        const float *__restrict__ gates, const float *__restrict__ cells,
                                        const float *__restrict__ d_out, const float *__restrict__ d_h_final,
                                        float *__restrict__ delta, int d_out_row, int d_out_dir,
                                        const float *__restrict__ d_out_mask) {
    TraceScope trace_scope(TK_ENCODER_BWD);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    int len = lengths[b];
    len = max(0, min(len, L));
    const int HP = (HE + 3) / 4 * 4;
    float *dh_s = lds, *dc_s = dh_s + HP, *dl_s = dc_s + HP, *scr = dl_s + 4 * HP;   // dl_s [4 HP], scr [kAnyThreads]
    const int64_t row0 = (int64_t)b * L;
    const float *w_hh = dir ? w_hh_r : w_hh_f;
    for (int k = tid; k < HE; k += kAnyThreads) { dh_s[k] = d_h_final ? d_h_final[(int64_t)b * HE + k] : 0.f; dc_s[k] = 0.f; }
    for (int idx = tid; idx < (L - len) * 4 * HE; idx += kAnyThreads)   // padded positions get delta = 0
        delta[((row0 + len + idx / (4 * HE)) * D + dir) * 4 * HE + idx % (4 * HE)] = 0.f;
    __syncthreads();
    for (int s = len - 1; s >= 0; --s) {
        const int t = dir ? (len - 1 - s) : s;
        const int64_t row = (row0 + t) * D + dir;
        for (int k = tid; k < HE; k += kAnyThreads) {
            const int64_t at = (row0 + t) * d_out_row + dir * d_out_dir + k;
            const float dh = dh_s[k] + (d_out_mask ? d_out[at] * d_out_mask[at] : d_out[at]);
            const float ig = gates[row * 4 * HE + k], fg = gates[row * 4 * HE + HE + k], gg = gates[row * 4 * HE + 2 * HE + k],
                        og = gates[row * 4 * HE + 3 * HE + k];
            const float c = cells[row * HE + k];
            const float c_prev = s > 0 ? cells[((row0 + (dir ? t + 1 : t - 1)) * D + dir) * HE + k] : 0.f;
            const float tc = tanhf_(c);
            const float dct = dc_s[k] + dh * og * (1.f - tc * tc);
            dc_s[k] = dct * fg;
            const float di = dct * gg * ig * (1.f - ig), df = dct * c_prev * fg * (1.f - fg), dg = dct * ig * (1.f - gg * gg),
                        d_o = dh * tc * og * (1.f - og);
            dl_s[k] = di; dl_s[HE + k] = df; dl_s[2 * HE + k] = dg; dl_s[3 * HE + k] = d_o;
            delta[row * 4 * HE + k] = di; delta[row * 4 * HE + HE + k] = df;
            delta[row * 4 * HE + 2 * HE + k] = dg; delta[row * 4 * HE + 3 * HE + k] = d_o;
        }
        __syncthreads();
        matvec_cols(w_hh, HE, 0, 4 * HE, HE, dl_s, scr, [&](int c, float v) { dh_s[c] = v; });
    }
}

template <typename K>
static int encoder_lds_attr(K kernel, size_t bytes, bool &attr_set) {
    GSCAN_CHECK(bytes <= kEncLdsLimit, "encoder lstm: a command of this length needs %zu bytes of LDS (limit %zu)", bytes,
                kEncLdsLimit);
    if (!attr_set) {
        GSCAN_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEncLdsLimit));
        attr_set = true;
    }
    return 0;
}

template <int HE>
static int launch_fwd(int B, int L, int D, const float *gx, const int32_t *lengths, const float *bf, const float *br,
                      float *out, float *hfin, float *gates, float *cells, float *hprev, const float *w_image,
                      float *hcat, const float *hcat_mask, const EncInput &in, hipStream_t stream) {
    const int nt = EncShape<HE>::kThreads;
    static bool attr_set = false;
    const size_t lds = encoder_fwd_lds(L, HE, in.x ? in.E : 0);
    if (int rc = encoder_lds_attr(encoder_lstm_fwd_kernel<HE>, lds, attr_set)) return rc;
    // algorithmic work: the recurrent product h.W_hh^T per (row, step, direction), plus the input projection when the
    // kernel computes it itself; padded steps counted
    ProbeScope probe(P_ENCODER_FWD, stream, 2.0 * B * L * D * 4 * HE * (HE + (in.x ? in.E : 0)));
    hipLaunchKernelGGL(encoder_lstm_fwd_kernel<HE>, dim3(B, D), dim3(nt), lds, stream, L, D, gx, lengths,
                       bf, br, out, hfin, gates, cells, hprev, w_image, hcat, hcat_mask, in);
    GSCAN_LAUNCHED("encoder_lstm_fwd_kernel");
    return 0;
}
template <int HE>
static int launch_bwd(int B, int L, int D, const int32_t *lengths, const float *wf, const float *wr,
                      const float *gates, const float *cells, const float *d_out, const float *d_hfin, float *delta,
                      int d_out_row, int d_out_dir, const float *d_out_mask, hipStream_t stream) {
    const int nt = EncShape<HE>::kThreads;
    static bool attr_set = false;
    if (int rc = encoder_lds_attr(encoder_lstm_bwd_kernel<HE>, encoder_bwd_lds(L, HE), attr_set)) return rc;
    ProbeScope probe(P_ENCODER_BWD, stream, 2.0 * B * L * D * 4 * HE * HE);
    hipLaunchKernelGGL(encoder_lstm_bwd_kernel<HE>, dim3(B, D), dim3(nt), encoder_bwd_lds(L, HE), stream, L, D, lengths, wf,
                       wr, gates, cells, d_out, d_hfin, delta, d_out_row, d_out_dir, d_out_mask);
    GSCAN_LAUNCHED("encoder_lstm_bwd_kernel");
    return 0;
}

#define GSCAN_HIDDEN_SIZES(X) GSCAN_ENC_HIDDEN_SIZES(X)

bool hidden_size_supported(int h) {
#define X(n) if (h == n) return true;
    GSCAN_HIDDEN_SIZES(X)
#undef X
    return false;
}

// image[dir][r][k][j] = W_hh_dir[j + r*NT][k], NT = 4He / rows-per-thread (the step prologue writes the same image)
__global__ void encoder_weight_image_kernel(const float *__restrict__ w_f, const float *__restrict__ w_r, int He,
                                            int D, int rows, float *__restrict__ image) {
    const int nt = 4 * He / rows, total = D * 4 * He * He;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i % nt, k = (i / nt) % He, r = (i / (nt * He)) % rows, dir = i / (4 * He * He);
        image[i] = (dir ? w_r : w_f)[(int64_t)(j + r * nt) * He + k];
    }
}

int encoder_weight_image(const float *w_hh_f, const float *w_hh_r, int He, int D, float *image, hipStream_t stream) {
    const int total = D * 4 * He * He;
    hipLaunchKernelGGL(encoder_weight_image_kernel, dim3(std::min(cdiv(total, 256), 1024)), dim3(256), 0, stream, w_hh_f,
                       w_hh_r, He, D, encoder_rows_per_thread(He), image);
    GSCAN_LAUNCHED("encoder_weight_image_kernel");
    return 0;
}

// The register/LDS-resident kernels take the compiled hidden sizes and commands whose whole history fits LDS; every
// other shape runs on the streaming kernels (GSCAN_ENCODER_ANY=1: every shape does, for tests).
bool encoder_fast_supported(int He, int L, int E) {
    static const int force_any = [] { const char *e = getenv("GSCAN_ENCODER_ANY"); return e ? atoi(e) : 0; }();
    if (force_any || !hidden_size_supported(He)) return false;
    return encoder_fwd_lds(L, He, E) <= kEncLdsLimit && encoder_bwd_lds(L, He) <= kEncLdsLimit;
}

int encoder_rows_per_thread(int He) {
    switch (He) {
#define X(n) case n: return EncShape<n>::R;
        GSCAN_HIDDEN_SIZES(X)
#undef X
        default: return 1;
    }
}

int encoder_lstm_forward(int B, int L, int He, int D, const float *gx, const int32_t *lengths, const float *w_hh_f,
                         const float *b_hh_f, const float *w_hh_r, const float *b_hh_r, float *out, float *h_final,
                         float *gates, float *cells, float *hprev, const float *w_image, hipStream_t stream,
                         float *hcat, const float *hcat_mask, const EncInput *input) {
    GSCAN_CHECK(B > 0 && L > 0 && (D == 1 || D == 2), "encoder lstm: bad dims B=%d L=%d D=%d", B, L, D);
    EncInput in{};
    if (input) in = *input;
    GSCAN_CHECK(!in.x || in.E > 0, "encoder lstm: the first layer's own input projection needs its input width");
    GSCAN_CHECK(in.x || gx, "encoder lstm: neither input projections nor inputs given");
    GSCAN_CHECK(hcat || (out && h_final), "encoder lstm: neither direction sums nor per-direction outputs requested");
    GSCAN_CHECK(D == 1 || (w_hh_r && b_hh_r), "encoder lstm: reverse weights missing");
    if (encoder_fast_supported(He, L, in.x ? in.E : 0)) {
        GSCAN_CHECK(((in.x ? 0 : (uintptr_t)gx) | (uintptr_t)gates) % 16 == 0, "encoder lstm: gx and gates must be 16-byte aligned");
        GSCAN_CHECK(w_image && (!in.x || in.w_ih_t), "encoder lstm: weight image missing (W_hh registers; the first layer's "
                    "column-major [W_ih | b_ih + b_hh])");
        switch (He) {
#define X(n) case n: return launch_fwd<n>(B, L, D, gx, lengths, b_hh_f, b_hh_r, out, h_final, gates, cells, hprev, w_image, hcat, hcat_mask, in, stream);
            GSCAN_HIDDEN_SIZES(X)
#undef X
            default: break;
        }
    }
    {   // any other hidden size / command length: weights streamed, one step's vectors in LDS
        GSCAN_CHECK(He >= 1 && w_hh_f && b_hh_f && (!in.x || (in.w_ih_f && in.b_ih_f && (D == 1 || (in.w_ih_r && in.b_ih_r)))),
                    "encoder lstm: encoder_hidden_size %d / weights missing", He);
        const int HP = (He + 3) / 4 * 4;
        const size_t lds = (size_t)(10 * HP + (in.x ? in.E : 0)) * sizeof(float);
        GSCAN_CHECK(lds <= kEncLdsLimit, "encoder lstm: encoder_hidden_size %d needs %zu bytes of LDS", He, lds);
        ProbeScope probe(P_ENCODER_FWD, stream, 2.0 * B * L * D * 4 * He * (He + (in.x ? in.E : 0)));
        if (He % 4 == 0)
            hipLaunchKernelGGL(encoder_lstm_fwd_any_kernel<true>, dim3(B, D), dim3(kAnyThreads), lds, stream, L, D, He, gx, lengths,
                               w_hh_f, b_hh_f, w_hh_r, b_hh_r, out, h_final, gates, cells, hprev, hcat, hcat_mask, in);
        else
            hipLaunchKernelGGL(encoder_lstm_fwd_any_kernel<false>, dim3(B, D), dim3(kAnyThreads), lds, stream, L, D, He, gx, lengths,
                               w_hh_f, b_hh_f, w_hh_r, b_hh_r, out, h_final, gates, cells, hprev, hcat, hcat_mask, in);
        GSCAN_LAUNCHED("encoder_lstm_fwd_any_kernel");
    }
    return 0;
}

int encoder_lstm_backward(int B, int L, int He, int D, const int32_t *lengths, const float *w_hh_f,
                          const float *w_hh_r, const float *gates, const float *cells, const float *d_out,
                          const float *d_h_final, float *delta, hipStream_t stream, int d_out_row, int d_out_dir,
                          const float *d_out_mask) {
    GSCAN_CHECK(B > 0 && L > 0 && (D == 1 || D == 2), "encoder lstm bwd: bad dims B=%d L=%d D=%d", B, L, D);
    if (d_out_row == 0) d_out_row = He;                      // the last layer: one gradient for both directions
    if (encoder_fast_supported(He, L, 0)) {
        GSCAN_CHECK(d_out_row % 4 == 0 && d_out_dir % 4 == 0 && ((uintptr_t)d_out_mask % 16) == 0,
                    "encoder lstm bwd: d_out strides / mask must keep 16-byte alignment");
        GSCAN_CHECK(((uintptr_t)gates | (uintptr_t)cells | (uintptr_t)d_out | (uintptr_t)delta) % 16 == 0,
                    "encoder lstm bwd: gates, cells, d_out and delta must be 16-byte aligned");
        switch (He) {
#define X(n) case n: return launch_bwd<n>(B, L, D, lengths, w_hh_f, w_hh_r, gates, cells, d_out, d_h_final, delta, d_out_row, d_out_dir, d_out_mask, stream);
            GSCAN_HIDDEN_SIZES(X)
#undef X
            default: break;
        }
    }
    {
        const int HP = (He + 3) / 4 * 4;
        const size_t lds = (size_t)(6 * HP + kAnyThreads) * sizeof(float);
        GSCAN_CHECK(He >= 1 && lds <= kEncLdsLimit, "encoder lstm bwd: encoder_hidden_size %d", He);
        ProbeScope probe(P_ENCODER_BWD, stream, 2.0 * B * L * D * 4 * He * He);
        hipLaunchKernelGGL(encoder_lstm_bwd_any_kernel, dim3(B, D), dim3(kAnyThreads), lds, stream, L, D, He, lengths, w_hh_f, w_hh_r,
                           gates, cells, d_out, d_h_final, delta, d_out_row, d_out_dir, d_out_mask);
        GSCAN_LAUNCHED("encoder_lstm_bwd_any_kernel");
    }
    return 0;
}

GSCAN_TRACE_TU(lstm_encoder)

}  // namespace gscan
