// Command encoder recurrence (seq2seq/seq2seq_model.py:62-88), forward and backward.
//
// The reference sorts rows by length, packs them and calls nn.LSTM.  Rows are independent,
// so here one workgroup owns one command (and direction) for its whole length: the recurrent
// matrix W_hh (4He x He floats) lives in the VGPRs of the workgroup for all time steps (a thread
// holds two rows), h is broadcast from LDS, and there is no global traffic inside the loop except the
// precomputed input projection gx[t] coming in and the saved activations going out.  Length
// masking replaces packing: a row simply stops after its own length (the reverse direction
// starts at its last real token), outputs at padded positions stay zero.
//
// Backward keeps the transposed ownership: thread (s, k) holds W_hh[s*He .. s*He+He-1][k], so
// dh_{t-1}[k] = sum_j W_hh[j][k] * delta[j] is four in-register partial dots plus one LDS sum.
// The kernel emits only the gate pre-activation gradients delta[b,t,dir,4He]; all weight
// gradients are dense GEMMs over those afterwards.
#include "step.h"

namespace gscan {

// R rows of weights against one LDS vector: the vector is read once (same address in every lane: LDS broadcast,
// 8 cycles of LDS bandwidth per 16-byte wave read whatever the lanes hold) and feeds R dot products.  The
// recurrences are bound by exactly these reads, so a thread keeps TWO weight rows (R = 2) when the register file
// allows (He <= 100): half as many waves read the same vector.
template <int HE, int R>
__device__ __forceinline__ void dots_lds(const float (&w)[R][HE], const float *v, float (&out)[R]) {
    static_assert(HE % 4 == 0, "hidden size must be a multiple of 4");
    float a0[R], a1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a0[r] = a1[r] = 0.f;
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
#pragma unroll
    for (int i = 0; i < HE / 4; ++i) {
        const float4 x = v4[i];
#pragma unroll
        for (int r = 0; r < R; ++r) {              // R * 2 independent chains: enough to cover the FMA latency
            a0[r] = fmaf(w[r][4 * i + 0], x.x, a0[r]);
            a1[r] = fmaf(w[r][4 * i + 1], x.y, a1[r]);
            a0[r] = fmaf(w[r][4 * i + 2], x.z, a0[r]);
            a1[r] = fmaf(w[r][4 * i + 3], x.w, a1[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = a0[r] + a1[r];
}

template <int HE> struct EncShape {
    static constexpr int R = HE <= 100 ? 2 : 1;                       // weight rows (forward) / columns (backward) per thread
    static constexpr int kThreads = ((4 * HE / R + 63) / 64) * 64;
};

// grid (B, D): the two directions of a row run as two workgroups (they only meet in the sums below).
// `out` and `h_final` must be zero on entry: each direction ADDS its h (0 + h_f + h_r in either order is the same
// float: two-operand addition commutes), which is how the directions are summed (seq2seq_model.py:77-81).
// Thread j < 4He/R owns gate rows j + r*(4He/R), r < R  (R = 2: [i | f] rows and the matching [g | o] rows).
template <int HE>
__global__ __launch_bounds__(EncShape<HE>::kThreads, 2) void encoder_lstm_fwd_kernel(int L, int D, const float *__restrict__ gx,
                                        const int32_t *__restrict__ lengths, const float *__restrict__ w_hh_f,
                                        const float *__restrict__ b_hh_f, const float *__restrict__ w_hh_r,
                                        const float *__restrict__ b_hh_r, float *__restrict__ out,
                                        float *__restrict__ h_final, float *__restrict__ gates,
                                        float *__restrict__ cells, float *__restrict__ hprev,
                                        const float *__restrict__ w_image) {
    constexpr int R = EncShape<HE>::R, NT = 4 * HE / R;              // owning threads
    __shared__ __attribute__((aligned(16))) float h_s[HE];
    __shared__ float gate_s[4 * HE];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    int len = lengths[b];
    len = max(0, min(len, L));
    const bool is_gate = j < NT, is_unit = j < HE;
    float w[R][HE];

    // padded positions: zero saved h_prev (it multiplies delta = 0 in a GEMM later); out stays zero
    for (int idx = j; idx < (L - len) * HE; idx += blockDim.x) {
        const int t = len + idx / HE, k = idx % HE;
        hprev[(((int64_t)b * L + t) * D + dir) * HE + k] = 0.f;
    }

    const float *w_hh = dir ? w_hh_r : w_hh_f;
    const float *b_hh = dir ? b_hh_r : b_hh_f;
    float bias[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        bias[r] = 0.f;
        if (is_gate) {
            const int row = j + r * NT;
            // register image [dir][r][k][thread]: consecutive lanes read consecutive floats (a row per lane straight
            // from W_hh would touch 64 cache lines per load; keeping both paths in one kernel cost 96 spilled VGPRs)
            const float *img = w_image + ((int64_t)(dir * R + r) * HE) * NT + j;
#pragma unroll
            for (int k = 0; k < HE; ++k) w[r][k] = img[k * NT];
            bias[r] = b_hh[row];
        }
    }
    float c = 0.f;
    if (is_unit) h_s[j] = 0.f;
    // The input projections are fetched TWO steps ahead: memory operations retire in order on this hardware, so a
    // wait for a load also waits for every store issued before it — with a one-step distance each step would wait
    // for the previous step's stores to be acknowledged.  For the same reason the direction sum (atomics) is not
    // issued inside the loop: h_t is kept in LDS and added to `out` once, after the last step.
    extern __shared__ float hist_s[];                        // [L][HE]
    float gx_a[R], gx_b[R];                                  // projections of steps s and s+1 (bias included)
    auto fetch = [&](int s, float (&dst)[R]) {
        const int t = dir ? (len - 1 - s) : s;
        const float *g = gx + (((int64_t)b * L + t) * D + dir) * 4 * HE;
#pragma unroll
        for (int r = 0; r < R; ++r) dst[r] = g[j + r * NT] + bias[r];
    };
#pragma unroll
    for (int r = 0; r < R; ++r) gx_a[r] = gx_b[r] = 0.f;
    if (is_gate && len > 0) fetch(0, gx_a);
    if (is_gate && len > 1) fetch(1, gx_b);
    lds_barrier();
    for (int s = 0; s < len; ++s) {
        const int t = dir ? (len - 1 - s) : s;
        const int64_t row = ((int64_t)b * L + t) * D + dir;
        float gx_cur[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { gx_cur[r] = gx_a[r]; gx_a[r] = gx_b[r]; }
        if (is_gate && s + 2 < len) fetch(s + 2, gx_b);
        if (is_gate) {
            float dot[R];
            dots_lds<HE, R>(w, h_s, dot);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int gr = j + r * NT;                            // gate row: [i | f | g | o] blocks of He
                const float pre = gx_cur[r] + dot[r];
                const float a = (gr >= 2 * HE && gr < 3 * HE) ? tanhf_(pre) : sigmoidf_(pre);
                gate_s[gr] = a;
                gates[row * 4 * HE + gr] = a;
            }
        }
        if (is_unit) hprev[row * HE + j] = h_s[j];
        lds_barrier();
        if (is_unit) {
            const float ig = gate_s[j], fg = gate_s[HE + j], gg = gate_s[2 * HE + j], og = gate_s[3 * HE + j];
            c = fg * c + ig * gg;
            const float h = og * tanhf_(c);
            cells[row * HE + j] = c;
            h_s[j] = h;
            hist_s[t * HE + j] = h;
        }
        lds_barrier();
    }
    if (is_unit) {
        atomicAdd(h_final + (int64_t)b * HE + j, h_s[j]);
        for (int t = 0; t < len; ++t) atomicAdd(out + ((int64_t)b * L + t) * HE + j, hist_s[t * HE + j]);
    }
}

// Backward: thread (seg, q) owns columns q + r*(He/R), r < R, of block seg of W_hh (its R dot products share the
// delta segment they read); dh_{t-1}[k] = sum of the four blocks' partial products.
template <int HE>
__global__ __launch_bounds__(EncShape<HE>::kThreads, 2) void encoder_lstm_bwd_kernel(int L, int D, const int32_t *__restrict__ lengths,
                                        const float *__restrict__ w_hh_f, const float *__restrict__ w_hh_r,
                                        const float *__restrict__ gates, const float *__restrict__ cells,
                                        const float *__restrict__ d_out, const float *__restrict__ d_h_final,
                                        float *__restrict__ delta) {
    constexpr int R = EncShape<HE>::R, KQ = HE / R, NT = 4 * KQ;
    static_assert(HE % R == 0, "hidden size must divide by the columns per thread");
    __shared__ __attribute__((aligned(16))) float delta_s[4 * HE];
    __shared__ float part_s[4 * HE];
    __shared__ float dh_s[HE];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    int len = lengths[b];
    len = max(0, min(len, L));
    const bool active = tid < NT, is_unit = tid < HE;
    const int seg = tid / KQ, q = tid % KQ;
    const float *w_hh = dir ? w_hh_r : w_hh_f;
    float wt[R][HE];
    if (active) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int jj = 0; jj < HE; ++jj) wt[r][jj] = w_hh[(int64_t)(seg * HE + jj) * HE + q + r * KQ];
    }
    for (int idx = tid; idx < (L - len) * 4 * HE; idx += blockDim.x) {
        const int t = len + idx / (4 * HE), jj = idx % (4 * HE);
        delta[(((int64_t)b * L + t) * D + dir) * 4 * HE + jj] = 0.f;
    }
    float dc = 0.f;
    if (is_unit) dh_s[tid] = d_h_final[(int64_t)b * HE + tid];
    // saved activations of the next step to process are fetched while the current one computes
    float pf[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto prefetch = [&](int s) {
        const int t = dir ? (len - 1 - s) : s;
        const int64_t row = ((int64_t)b * L + t) * D + dir;
        const float *g = gates + row * 4 * HE;
        pf[0] = g[tid]; pf[1] = g[HE + tid]; pf[2] = g[2 * HE + tid]; pf[3] = g[3 * HE + tid];
        pf[4] = cells[row * HE + tid];
        pf[5] = 0.f;
        if (s > 0) {
            const int tp = dir ? (t + 1) : (t - 1);
            pf[5] = cells[(((int64_t)b * L + tp) * D + dir) * HE + tid];
        }
        pf[6] = d_out[((int64_t)b * L + t) * HE + tid];
    };
    if (is_unit && len > 0) prefetch(len - 1);
    lds_barrier();
    for (int s = len - 1; s >= 0; --s) {
        const int t = dir ? (len - 1 - s) : s;
        const int64_t row = ((int64_t)b * L + t) * D + dir;
        if (is_unit) {
            const float dh = dh_s[tid] + pf[6];
            const float ig = pf[0], fg = pf[1], gg = pf[2], og = pf[3];
            const float c = pf[4];
            const float c_prev = pf[5];
            if (s > 0) prefetch(s - 1);
            const float tc = tanhf_(c);
            const float dct = dc + dh * og * (1.f - tc * tc);
            const float di = dct * gg * ig * (1.f - ig);
            const float df = dct * c_prev * fg * (1.f - fg);
            const float dg = dct * ig * (1.f - gg * gg);
            const float d_o = dh * tc * og * (1.f - og);
            dc = dct * fg;
            delta_s[tid] = di; delta_s[HE + tid] = df; delta_s[2 * HE + tid] = dg; delta_s[3 * HE + tid] = d_o;
            float *dg_out = delta + row * 4 * HE;
            dg_out[tid] = di; dg_out[HE + tid] = df; dg_out[2 * HE + tid] = dg; dg_out[3 * HE + tid] = d_o;
        }
        lds_barrier();
        if (active) {
            float part[R];
            dots_lds<HE, R>(wt, delta_s + seg * HE, part);
#pragma unroll
            for (int r = 0; r < R; ++r) part_s[seg * HE + q + r * KQ] = part[r];
        }
        lds_barrier();
        if (is_unit) dh_s[tid] = (part_s[tid] + part_s[HE + tid]) + (part_s[2 * HE + tid] + part_s[3 * HE + tid]);
        lds_barrier();
    }
}

template <int HE>
static int launch_fwd(int B, int L, int D, const float *gx, const int32_t *lengths, const float *wf, const float *bf,
                      const float *wr, const float *br, float *out, float *hfin, float *gates, float *cells,
                      float *hprev, const float *w_image, hipStream_t stream) {
    const int nt = EncShape<HE>::kThreads;
    // algorithmic work: the recurrent product h.W_hh^T per (row, step, direction); padded steps counted
    ProbeScope probe(P_ENCODER_FWD, stream, 2.0 * B * L * D * 4 * HE * HE);
    hipLaunchKernelGGL(encoder_lstm_fwd_kernel<HE>, dim3(B, D), dim3(nt), (size_t)L * HE * sizeof(float), stream, L, D, gx, lengths, wf, bf, wr, br,
                       out, hfin, gates, cells, hprev, w_image);
    GSCAN_LAUNCHED("encoder_lstm_fwd_kernel");
    return 0;
}
template <int HE>
static int launch_bwd(int B, int L, int D, const int32_t *lengths, const float *wf, const float *wr,
                      const float *gates, const float *cells, const float *d_out, const float *d_hfin, float *delta,
                      hipStream_t stream) {
    const int nt = EncShape<HE>::kThreads;
    ProbeScope probe(P_ENCODER_BWD, stream, 2.0 * B * L * D * 4 * HE * HE);
    hipLaunchKernelGGL(encoder_lstm_bwd_kernel<HE>, dim3(B, D), dim3(nt), 0, stream, L, D, lengths, wf, wr, gates,
                       cells, d_out, d_hfin, delta);
    GSCAN_LAUNCHED("encoder_lstm_bwd_kernel");
    return 0;
}

#define GSCAN_HIDDEN_SIZES(X) X(20) X(32) X(64) X(100) X(128)

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
                         float *gates, float *cells, float *hprev, const float *w_image, hipStream_t stream) {
    GSCAN_CHECK(B > 0 && L > 0 && (D == 1 || D == 2), "encoder lstm: bad dims B=%d L=%d D=%d", B, L, D);
    GSCAN_CHECK(D == 1 || (w_hh_r && b_hh_r), "encoder lstm: reverse weights missing");
    GSCAN_CHECK(w_image, "encoder lstm: weight image missing");
    switch (He) {
#define X(n) case n: return launch_fwd<n>(B, L, D, gx, lengths, w_hh_f, b_hh_f, w_hh_r, b_hh_r, out, h_final, gates, cells, hprev, w_image, stream);
        GSCAN_HIDDEN_SIZES(X)
#undef X
        default: break;
    }
    GSCAN_CHECK(false, "encoder_hidden_size %d has no compiled kernel (supported: 20 32 64 100 128)", He);
}

int encoder_lstm_backward(int B, int L, int He, int D, const int32_t *lengths, const float *w_hh_f,
                          const float *w_hh_r, const float *gates, const float *cells, const float *d_out,
                          const float *d_h_final, float *delta, hipStream_t stream) {
    GSCAN_CHECK(B > 0 && L > 0 && (D == 1 || D == 2), "encoder lstm bwd: bad dims B=%d L=%d D=%d", B, L, D);
    switch (He) {
#define X(n) case n: return launch_bwd<n>(B, L, D, lengths, w_hh_f, w_hh_r, gates, cells, d_out, d_h_final, delta, stream);
        GSCAN_HIDDEN_SIZES(X)
#undef X
        default: break;
    }
    GSCAN_CHECK(false, "encoder_hidden_size %d has no compiled kernel (supported: 20 32 64 100 128)", He);
}

}  // namespace gscan
