// Small memory-bound kernels of the training step: embedding gather / gradient scatter, the step
// prologue (weight images, embeddings, zeroing), fused Adam and the Philox dropout-mask generator.
// All are one-pass, coalesced along the innermost (feature) dimension.
#include "step.h"
#include "prologue.h"

namespace gscan {

// ------------------------------------------------------------------------------------------
// out[row, 0:D] (row stride ldo) = table[tok[row], :] * mask[row, :]
// nn.Embedding + nn.Dropout of seq2seq_model.py:58-59 and :383-384.
// ------------------------------------------------------------------------------------------
__global__ void embed_kernel(const int64_t *__restrict__ tok, const float *__restrict__ table, int vocab,
                             const float *__restrict__ mask, int rows, int D, float *__restrict__ out, int64_t ldo) {
    const int64_t total = (int64_t)rows * D;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(idx % D);
        const int64_t row = idx / D;
        int64_t t = tok[row];
        float v = (t >= 0 && t < vocab) ? table[t * D + d] : 0.f;
        if (mask) v *= mask[idx];
        out[row * ldo + d] = v;
    }
}

int embed_rows(const int64_t *tok, const float *table, int vocab, const float *mask, int rows, int D, float *out,
               int64_t ldo, hipStream_t stream) {
    const int64_t total = (int64_t)rows * D;
    hipLaunchKernelGGL(embed_kernel, dim3((int)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0, stream, tok,
                       table, vocab, mask, rows, D, out, ldo);
    GSCAN_LAUNCHED("embed_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------------
// dtable[v, :] += sum over rows with tok[row] == v, v != pad of g[row, :] * mask[row, :]
// (the padding row of nn.Embedding(padding_idx=...) never receives gradient).
// Vocabularies are tiny (6..21 entries) and thousands of rows carry the same token, so atomics on a shared table —
// global or LDS — serialise (round 1: 18 us for 5 120 x 100).  A thread owns ONE column and a row slot: 256 threads
// = (256 / DP) row slots x DP columns, DP = the next power of two >= D; it walks its rows with its own [vocab] sums in
// LDS (priv[v][thread]: nobody else touches them, plain read-modify-write), the slots of a column are added at the
// end and one global atomic per touched (v, column) leaves the workgroup.
// ------------------------------------------------------------------------------------------
constexpr int kEmbedRows = 32;
constexpr int kEmbedThreads = 256;
// Columns c0 .. c0 + D - 1 of a table that is DT columns wide (wider tables than a workgroup has threads are walked in
// column blocks by the host: round 5, any embedding width).
__global__ __launch_bounds__(kEmbedThreads) void embed_grad_kernel(const int64_t *__restrict__ tok, const float *__restrict__ g, int64_t ldg,
                                  const float *__restrict__ mask, int rows, int D, int DP, int v0, int vocab, int pad,
                                  float *__restrict__ dtable, float *__restrict__ part, int vocab_all,     // tokens v0 .. v0 + vocab - 1
                                  int c0, int DT) {
    TraceScope trace_scope(TK_EMBED_GRAD);
    extern __shared__ float priv[];                       // [vocab][kEmbedThreads]
    const int tid = threadIdx.x, d = tid & (DP - 1), slot = tid / DP, slots = kEmbedThreads / DP;
    for (int v = 0; v < vocab; ++v) priv[v * kEmbedThreads + tid] = 0.f;
    const int r0 = blockIdx.x * kEmbedRows, r1 = min(rows, r0 + kEmbedRows);
    if (d < D) {
        // eight rows per pass: token, gradient and mask loads are all unconditional (a row past the end repeats the
        // last one with weight zero), so one pass costs one memory latency, not a chain of them
        constexpr int U = 8;
        for (int r = r0 + slot; r < r1; r += U * slots) {
            float x[U], m[U];
            int64_t t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = min(r + u * slots, r1 - 1);
                t[u] = tok[rr];
                x[u] = g[(int64_t)rr * ldg + c0 + d];
                m[u] = mask ? mask[(int64_t)rr * DT + c0 + d] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t tv = t[u] - v0;
                const bool live = r + u * slots < r1 && tv >= 0 && tv < vocab && t[u] != pad;
                if (live) priv[(int)tv * kEmbedThreads + tid] += x[u] * m[u];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < vocab * D; i += kEmbedThreads) {
        const int v = i / D, c = i - v * D;
        float sum = 0.f;
        for (int s = 0; s < slots; ++s) sum += priv[v * kEmbedThreads + s * DP + c];
        if (part) part[((int64_t)blockIdx.x * vocab_all + v0 + v) * DT + c0 + c] = sum;
        else if (sum != 0.f) atomicAdd(&dtable[(int64_t)(v0 + v) * DT + c0 + c], sum);
    }
}

// Fixed-order form (the deterministic mode of the training step): the workgroups of embed_grad_kernel write their sums
// to part[workgroup][vocab][D] instead of adding them to the table with atomics, and one thread per table element adds
// the workgroups' sums in workgroup order.
__global__ void embed_grad_reduce_kernel(const float *__restrict__ part, int chunks, int n, float *__restrict__ dtable) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dtable[i] += ordered_chunk_sum(part, chunks, n, i);
}

size_t embed_grad_partial_floats(int rows, int D, int vocab) { return (size_t)cdiv(rows, kEmbedRows) * vocab * D; }

int embed_grad(const int64_t *tok, const float *g, int64_t ldg, const float *mask, int rows, int D, int vocab,
               int pad, float *dtable, hipStream_t stream, float *part) {
    GSCAN_CHECK(D >= 1, "embed_grad: embedding dimension %d", D);
    constexpr int kChunk = 64;                             // vocabulary entries per launch: 64 KB of per-thread sums
    for (int c0 = 0; c0 < D; c0 += kEmbedThreads) {        // column blocks of a workgroup's width (one block up to 256 columns)
        const int Dt = std::min(kEmbedThreads, D - c0);
        int DP = 1;
        while (DP < Dt) DP <<= 1;
        for (int v0 = 0; v0 < vocab; v0 += kChunk) {
            const int n = std::min(kChunk, vocab - v0);
            hipLaunchKernelGGL(embed_grad_kernel, dim3(cdiv(rows, kEmbedRows)), dim3(kEmbedThreads),
                               (size_t)n * kEmbedThreads * sizeof(float), stream, tok, g, ldg, mask, rows, Dt, DP, v0, n, pad,
                               dtable, part, vocab, c0, D);
            GSCAN_LAUNCHED("embed_grad_kernel");
        }
    }
    if (part) {
        hipLaunchKernelGGL(embed_grad_reduce_kernel, dim3(cdiv(vocab * D, 64)), dim3(64), 0, stream, part,
                           cdiv(rows, kEmbedRows), vocab * D, dtable);
        GSCAN_LAUNCHED("embed_grad_reduce_kernel");
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Fused Adam over the flat parameter buffer (torch.optim.Adam semantics, train.py:68,111).
// ------------------------------------------------------------------------------------------
__global__ void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, size_t n, float step_size, float beta1, float beta2, float eps,
                            float inv_sqrt_bc2, const float *__restrict__ grad_scale,
                            const float *__restrict__ dev_scalars, int zero_grad) {
    TraceScope trace_scope(TK_ADAM);
    const float gs = grad_scale ? ((zero_grad & 2) ? 1.f / grad_scale[0] : grad_scale[0]) : 1.f;
    if (dev_scalars) { step_size = dev_scalars[0]; inv_sqrt_bc2 = dev_scalars[1]; }   // graph replay: per-step values
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        if (zero_grad & 1) g[i] = 0.f;             // optimizer.zero_grad() of train.py:113 folded in
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] -= step_size * (mi / denom);
    }
}

void adam_scalars(float lr, float beta1, float beta2, float lr_decay, float lr_decay_steps, int64_t step,
                  float *step_size, float *inv_sqrt_bc2) {
    const double lr_t = (double)lr * pow((double)lr_decay, (double)(step - 1) / (double)lr_decay_steps);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    *step_size = (float)(lr_t / bc1);
    *inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}

int adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
              float beta2, float eps, float lr_decay, float lr_decay_steps, int64_t step, const float *grad_scale,
              const float *dev_scalars, int zero_grad, hipStream_t stream) {
    GSCAN_CHECK(step >= 1 || dev_scalars, "adam: step is 1-based (got %lld)", (long long)step);
    float step_size = 0.f, inv_sqrt_bc2 = 0.f;
    if (!dev_scalars) adam_scalars(lr, beta1, beta2, lr_decay, lr_decay_steps, step, &step_size, &inv_sqrt_bc2);
    hipLaunchKernelGGL(adam_kernel, dim3((int)std::min<size_t>(cdiv(n, 256), 2048)), dim3(256), 0, stream, param, grad,
                       exp_avg, exp_avg_sq, n, step_size, beta1, beta2, eps, inv_sqrt_bc2, grad_scale, dev_scalars,
                       zero_grad);
    GSCAN_LAUNCHED("adam_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Philox-4x32-10 counter RNG -> scaled Bernoulli keep mask.  Counter = element index / 4,
// key = seed, stream id in the high counter words; four outputs per counter.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0,
                                             uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// Up to three consecutive segments (the CNN / encoder / decoder masks of one step) with their own keep
// probabilities in one launch; the stream id may come from device memory (graph replay).
// Round 6: a counter is (quad index WITHIN its segment, segment id in bits 24.. of counter word 1) — as
// drop_quad (dropout.h) does for the masks drawn inside the kernels — not the quad index in the concatenated buffer: the
// segments are grow-only capacities (model.py:_draw_masks), and with one running index the encoder and decoder masks of a
// given (seed, step) moved whenever an earlier segment's capacity had grown, i.e. with the history of batch shapes.
struct MaskSegments { size_t end[3]; float p[3]; };

__device__ __forceinline__ size_t mask_quads(const MaskSegments &seg) {
    return (seg.end[0] + 3) / 4 + (seg.end[1] - seg.end[0] + 3) / 4 + (seg.end[2] - seg.end[1] + 3) / 4;
}
// the four mask values of quad q (numbered segment after segment) of the launch
__device__ __forceinline__ void mask_quad(float *__restrict__ out, const MaskSegments &seg, size_t q, uint64_t seed,
                                          uint64_t stream_id) {
    const size_t q0 = (seg.end[0] + 3) / 4, q1 = q0 + (seg.end[1] - seg.end[0] + 3) / 4;
    const int sid = q < q0 ? 0 : (q < q1 ? 1 : 2);
    const size_t ql = q - (sid == 0 ? 0 : (sid == 1 ? q0 : q1));
    const size_t begin = sid == 0 ? 0 : seg.end[sid - 1], end = seg.end[sid];
    const float p = seg.p[sid];
    uint32_t c0 = (uint32_t)ql, c1 = (uint32_t)(ql >> 32) | ((uint32_t)sid << 24), c2 = (uint32_t)stream_id,
             c3 = (uint32_t)(stream_id >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const uint32_t rnd[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t i = begin + ql * 4 + j;
        if (i < end) {
            const float u = (float)(rnd[j] >> 8) * (1.0f / 16777216.0f);   // [0,1)
            out[i] = (u >= p) ? 1.0f / (1.0f - p) : 0.f;
        }
    }
}

__global__ void dropout_mask_kernel(float *__restrict__ out, MaskSegments seg, uint64_t seed, uint64_t stream_id,
                                    const uint64_t *__restrict__ dev_stream_id) {
    TraceScope trace_scope(TK_DROPOUT);
    const size_t nquad = mask_quads(seg);
    if (dev_stream_id) stream_id = dev_stream_id[0];
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (size_t)gridDim.x * blockDim.x)
        mask_quad(out, seg, q, seed, stream_id);
}

int dropout_masks(float *out, const size_t (&n)[3], const float (&p)[3], uint64_t seed, uint64_t stream_id,
                  const uint64_t *dev_stream_id, hipStream_t stream) {
    MaskSegments seg;
    size_t acc = 0;
    for (int i = 0; i < 3; ++i) {
        GSCAN_CHECK(p[i] >= 0.f && p[i] < 1.f, "dropout: p=%g out of [0,1)", p[i]);
        acc += n[i];
        seg.end[i] = acc;
        seg.p[i] = p[i];
    }
    if (acc == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((int)std::min<size_t>(cdiv((acc + 3) / 4 + 2, 256), 2048)), dim3(256), 0,
                       stream, out, seg, seed, stream_id, dev_stream_id);
    GSCAN_LAUNCHED("dropout_mask_kernel");
    return 0;
}

// optimizer.step() + zero_grad() AND the dropout masks of the NEXT step in one launch (the masks depend on nothing but
// a counter): the first adam_blocks workgroups are the optimiser, the rest the Philox generator — one launch and one
// kernel boundary fewer at the head of every training step.  The backward pass that read the previous masks has
// finished when this kernel runs (same stream), so the new masks go into the same buffer.
__global__ void adam_masks_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                  float *__restrict__ v, size_t n, float step_size, float beta1, float beta2, float eps,
                                  float inv_sqrt_bc2, const float *__restrict__ grad_scale, int zero_grad, int adam_blocks,
                                  float *__restrict__ mask_out, MaskSegments seg, uint64_t seed, uint64_t stream_id) {
    TraceScope trace_scope(TK_ADAM);
    if ((int)blockIdx.x < adam_blocks) {
        const float gs = grad_scale ? ((zero_grad & 2) ? 1.f / grad_scale[0] : grad_scale[0]) : 1.f;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)adam_blocks * blockDim.x) {
            const float gi = g[i] * gs;
            if (zero_grad & 1) g[i] = 0.f;
            const float mi = beta1 * m[i] + (1.f - beta1) * gi;
            const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
            m[i] = mi;
            v[i] = vi;
            const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
            p[i] -= step_size * (mi / denom);
        }
        return;
    }
    const size_t nquad = mask_quads(seg), blocks = gridDim.x - adam_blocks;
    for (size_t q = (size_t)(blockIdx.x - adam_blocks) * blockDim.x + threadIdx.x; q < nquad; q += blocks * blockDim.x)
        mask_quad(mask_out, seg, q, seed, stream_id);
}

int adam_step_masks(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                    float beta2, float eps, float lr_decay, float lr_decay_steps, int64_t step, const float *grad_scale,
                    int zero_grad, float *mask_out, const size_t (&nm)[3], const float (&pm)[3], uint64_t seed,
                    uint64_t stream_id, hipStream_t stream) {
    GSCAN_CHECK(step >= 1, "adam: step is 1-based (got %lld)", (long long)step);
    float step_size = 0.f, inv_sqrt_bc2 = 0.f;
    adam_scalars(lr, beta1, beta2, lr_decay, lr_decay_steps, step, &step_size, &inv_sqrt_bc2);
    MaskSegments seg;
    size_t acc = 0;
    for (int i = 0; i < 3; ++i) {
        GSCAN_CHECK(pm[i] >= 0.f && pm[i] < 1.f, "dropout: p=%g out of [0,1)", pm[i]);
        acc += nm[i];
        seg.end[i] = acc;
        seg.p[i] = pm[i];
    }
    GSCAN_CHECK(acc == 0 || mask_out, "adam_step_masks: NULL mask buffer");
    const int adam_blocks = (int)std::min<size_t>(cdiv(n, 256), 2048);
    const int mask_blocks = acc ? (int)std::min<size_t>(cdiv((acc + 3) / 4 + 2, 256), 2048) : 0;
    hipLaunchKernelGGL(adam_masks_kernel, dim3(adam_blocks + mask_blocks), dim3(256), 0, stream, param, grad, exp_avg,
                       exp_avg_sq, n, step_size, beta1, beta2, eps, inv_sqrt_bc2, grad_scale, zero_grad, adam_blocks, mask_out,
                       seg, seed, stream_id);
    GSCAN_LAUNCHED("adam_masks_kernel");
    return 0;
}

// The masks a step with dropout drawn in its kernels uses (dropout.h), written to memory: one thread per Philox counter of
// each segment.  cnn [B, M, 3Co], enc [rows_e, E], dec [rows_d, H]; a NULL pointer skips the segment.
__global__ void dropout_kernel_layout_kernel(float *__restrict__ cnn, float *__restrict__ enc, float *__restrict__ dec,
                                             DropSpec dc, DropSpec de, DropSpec dd, int B, int M, int Co, int64_t rows_e,
                                             int E, int64_t rows_d, int H) {
    const int ochunks = (Co + 63) >> 6, npairs = M * 3 * ochunks, ngroups = (npairs + 3) >> 2, F = 3 * Co;
    const int64_t n_c = cnn ? (int64_t)B * ngroups * 64 : 0, n_e = enc ? (rows_e + 3) / 4 * E : 0, n_d = dec ? (rows_d + 3) / 4 * H : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_c + n_e + n_d; i += (int64_t)gridDim.x * blockDim.x) {
        float keep[4];
        if (i < n_c) {
            const int lane = (int)(i & 63);
            const int64_t bg = i >> 6;
            const int b = (int)(bg / ngroups), grp = (int)(bg - (int64_t)b * ngroups);
            drop_quad(dc, kDropSegCnn, (uint64_t)i, keep);
            for (int j = 0; j < 4; ++j) {
                const int pair = 4 * grp + j;
                if (pair >= npairs) continue;
                const int q = pair / (3 * ochunks), rem = pair - q * 3 * ochunks, conv = rem / ochunks, oc = rem - conv * ochunks;
                const int o = oc * 64 + lane;
                if (o < Co) cnn[((int64_t)b * M + q) * F + conv * Co + o] = dc.on ? keep[j] : 1.f;
            }
        } else if (i < n_c + n_e) {
            const int64_t k = i - n_c, rq = k / E;
            const int d = (int)(k - rq * E);
            drop_quad(de, kDropSegEnc, (uint64_t)k, keep);
            for (int j = 0; j < 4; ++j)
                if (4 * rq + j < rows_e) enc[(4 * rq + j) * E + d] = de.on ? keep[j] : 1.f;
        } else {
            const int64_t k = i - n_c - n_e, rq = k / H;
            const int d = (int)(k - rq * H);
            drop_quad(dd, kDropSegDec, (uint64_t)k, keep);
            for (int j = 0; j < 4; ++j)
                if (4 * rq + j < rows_d) dec[(4 * rq + j) * H + d] = dd.on ? keep[j] : 1.f;
        }
    }
}

int dropout_masks_kernel_layout(float *cnn, float *enc, float *dec, int B, int M, int Co, int L, int E, int T, int H,
                                float p_cnn, float p_enc, float p_dec, uint64_t seed, uint64_t stream_id, hipStream_t stream) {
    for (float p : {p_cnn, p_enc, p_dec}) GSCAN_CHECK(p >= 0.f && p < 1.f, "dropout: p=%g out of [0,1)", p);
    GSCAN_CHECK(B > 0 && M > 0 && Co > 0 && L > 0 && E > 0 && T > 0 && H > 0, "dropout_masks_kernel_layout: bad dimensions");
    const int64_t total = (int64_t)B * drop_cnn_groups(M, Co) * 64 + ((int64_t)B * L + 3) / 4 * E + ((int64_t)B * T + 3) / 4 * H;
    hipLaunchKernelGGL(dropout_kernel_layout_kernel, dim3((int)std::min<int64_t>(cdiv(total, 256), 2048)), dim3(256), 0, stream,
                       cnn, enc, dec, drop_spec(true, seed, stream_id, p_cnn), drop_spec(true, seed, stream_id, p_enc),
                       drop_spec(true, seed, stream_id, p_dec), B, M, Co, (int64_t)B * L, E, (int64_t)B * T, H);
    GSCAN_LAUNCHED("dropout_kernel_layout_kernel");
    return 0;
}

int dropout_mask(float *out, size_t n, float p, uint64_t seed, uint64_t stream_id, hipStream_t stream) {
    const size_t ns[3] = {n, 0, 0};
    const float ps[3] = {p, 0.f, 0.f};
    return dropout_masks(out, ns, ps, seed, stream_id, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// Step prologue: everything that only rearranges parameters or gathers embeddings, in ONE launch.
//   seg 0  bsum[4H]            = dec_b_ih + dec_b_hh
//   seg 1  head_wc[V,4H]       = W_h2o . W_o2h, columns reordered from W_o2h's [e|h|ctx_t|ctx_v] to S order [e|ctx_t|ctx_v|h]:
//                                the output head as one matrix (decoder.hip)
//   seg 2  wih_stack[D*4He,E]  = [W_ih_fwd ; W_ih_rev]   (one GEMM then gives dXe for both directions), and its
//          per-direction transpose wih_t[D][E][4He] (lstm_encoder.hip's input projection)
//   seg 3  dwc[V,4H]           = 0   (gradient scratch of head_wc, filled by a split-K GEMM in backward)
//   seg 4  xe[B*L,E]           = dropout(Emb_enc[commands])      seq2seq_model.py:58-59
//   seg 5  S[:, 0:H]           = dropout(Emb_dec[targets])       seq2seq_model.py:383-384
//   seg 6  wcat5[5H,3H]        = [W_ih_dec ; (0 | W_q2k[:, H:2H] | 0)]: one product then carries delta AND dzq back to
//                                [e | ctx_text | ctx_vis]
//   seg 7  zero_extra          = 0   (accumulation targets: encoder direction sums enc_out / hN, split-K dxe)
//   seg 8  register images of the decoder's recurrent weights (step.h)
//   seg 9  register image of the encoder's recurrent weights: [dir][r][k][thread] = W_hh_dir[thread + r*NT][k]
//   seg 10 [tap][ch][o] image of the three convolution kernels (conv.hip)
//   seg 11-13 composite weights W_ih[:, ctx] . W_key (visual, textual; rows unit-major: row 4 unit + gate) and
//          W_q2k[:, ctx_text] . W_key_text
// ------------------------------------------------------------------------------------------

template <bool DRAWN>
__global__ void prologue_kernel(PrologueArgs a) {
    TraceScope trace_scope(TK_PROLOGUE);
    const int64_t total = a.end[13];
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x)
        prologue_element<20, DRAWN>(a, idx);
}

int step_prologue(const PrologueArgs &args, hipStream_t stream) {
    const int64_t total = args.end[13];
    if (total == 0) return 0;
    // two passes per thread at most at the benchmark shape (the kernel is chains of dependent loads, not bandwidth;
    // 2048 / 4096 / 8192 / 16384 workgroups: 0.521 / 0.517 / 0.523 / 0.524 ms per step)
    static const int cap = [] { const char *e = getenv("GSCAN_PROLOGUE_BLOCKS"); return e ? atoi(e) : 4096; }();
    const dim3 grid((int)std::min<int64_t>(cdiv(total, 256), cap));
    if (args.drop_enc.on || args.drop_dec.on) hipLaunchKernelGGL(prologue_kernel<true>, grid, dim3(256), 0, stream, args);
    else hipLaunchKernelGGL(prologue_kernel<false>, grid, dim3(256), 0, stream, args);
    GSCAN_LAUNCHED("prologue_kernel");
    return 0;
}

// The output head's weight gradients from d Wc = dlogits^T . S ([V, 4H], columns in S order [e | ctx_t | ctx_v | h]; the
// head is applied as the ONE matrix Wc = W_h2o . W_o2h, decoder.hip): by the chain rule through that product
//   g_w_o2h[j, src(col)] += sum_v W_h2o[v, j] . dWc[v, col]        (a thread per element, V terms)
//   g_w_h2o[v, j]        += sum_col dWc[v, col] . W_o2h[j, src(col)] (a wave per element, 4H terms)
// with src(col) W_o2h's own column order [e | h | ctx_t | ctx_v]; and, in the workgroups behind those, the energy-vector
// gradients: the decoder's reverse kernel leaves per-ROW sums [B, H] for the textual and the visual energy vector,
// added up over the batch here in a fixed order.
__device__ __forceinline__ int o2h_column(int col, int H) {          // S-order column -> column of W_o2h
    const int seg = col / H, k = col - seg * H;
    return (seg == 0 ? 0 : seg == 1 ? 2 * H : seg == 2 ? 3 * H : H) + k;
}
__global__ void head_grad_finish_kernel(const float *__restrict__ dwc, const float *__restrict__ w_h2o,
                                        const float *__restrict__ w_o2h, float *__restrict__ g_w_o2h,
                                        float *__restrict__ g_w_h2o, int H, int V, int n_o2h_blocks, int n_h2o_blocks,
                                        const float *__restrict__ dv_t_rows, const float *__restrict__ dv_v_rows, int B,
                                        float *__restrict__ g_v_t, float *__restrict__ g_v_v) {
    TraceScope trace_scope(TK_UNPERMUTE);
    const int blk = blockIdx.x;
    if (blk < n_o2h_blocks) {
        const int n = H * 4 * H;
        for (int i = blk * blockDim.x + threadIdx.x; i < n; i += n_o2h_blocks * blockDim.x) {
            const int j = i / (4 * H), col = i - j * 4 * H;
            float acc = 0.f;
            for (int v = 0; v < V; ++v) acc = fmaf(w_h2o[v * H + j], dwc[v * 4 * H + col], acc);
            g_w_o2h[(int64_t)j * 4 * H + o2h_column(col, H)] += acc;
        }
        return;
    }
    if (blk < n_o2h_blocks + n_h2o_blocks) {
        const int out = (blk - n_o2h_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (out >= V * H) return;
        const int v = out / H, j = out - v * H;
        float acc = 0.f;
        for (int col = lane; col < 4 * H; col += 64) acc = fmaf(dwc[v * 4 * H + col], w_o2h[(int64_t)j * 4 * H + o2h_column(col, H)], acc);
        acc = wave_sum(acc);
        if (lane == 0) g_w_h2o[out] += acc;
        return;
    }
    // a workgroup takes 32 columns of [dv_text | dv_vis]; a thread = (column, one of 8 slices of the batch rows): up to
    // 32 rows per pass, all loads of a pass in flight, then the 8 slice sums are added in a fixed order
    __shared__ float part[8][32];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5, k = (blk - n_o2h_blocks - n_h2o_blocks) * 32 + c;
    const bool live = k < 2 * H;
    const float *src = (k < H ? dv_t_rows : dv_v_rows) + (live ? (k < H ? k : k - H) : 0);
    const int per = (B + 7) / 8, r0 = slice * per, r1 = min(B, r0 + per);
    float acc = 0.f;
    for (int base = r0; base < r1; base += 32) {
        float x[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) x[u] = (live && base + u < r1) ? src[(int64_t)(base + u) * H] : 0.f;
#pragma unroll
        for (int u = 16; u > 0; u >>= 1)
#pragma unroll
            for (int v = 0; v < u; ++v) x[v] += x[v + u];
        acc += x[0];
    }
    part[slice][c] = acc;
    __syncthreads();
    if (slice == 0 && live) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += part[q][c];
        (k < H ? g_v_t : g_v_v)[k < H ? k : k - H] += sum;
    }
}

int head_grad_finish(const float *dwc, const float *w_h2o, const float *w_o2h, float *g_w_o2h, float *g_w_h2o, int H, int V,
                     hipStream_t stream, const float *dv_t_rows, const float *dv_v_rows, int B, float *g_v_t, float *g_v_v) {
    const int n_o2h = cdiv(H * 4 * H, 256), n_h2o = cdiv(V * H, 4), nsum = dv_t_rows ? cdiv(2 * H, 32) : 0;
    hipLaunchKernelGGL(head_grad_finish_kernel, dim3(n_o2h + n_h2o + nsum), dim3(256), 0, stream, dwc, w_h2o, w_o2h, g_w_o2h,
                       g_w_h2o, H, V, n_o2h, n_h2o, dv_t_rows, dv_v_rows, B, g_v_t, g_v_v);
    GSCAN_LAUNCHED("head_grad_finish_kernel");
    return 0;
}

GSCAN_TRACE_TU(elementwise)

}  // namespace gscan
