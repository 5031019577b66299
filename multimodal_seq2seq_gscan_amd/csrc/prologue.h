// One element of the step prologue's index space (the segments are listed in elementwise.hip): shared by the stand-alone
// prologue kernel (elementwise.hip) and by the launch that runs the prologue and the world encoder together (conv.hip).
// U = products of a composite-weight dot fetched per pass (2 U loads in flight): 20 where the registers allow it.
#pragma once
#include "step.h"

namespace gscan {

// DRAWN: the embedding dropout is drawn here (a.drop_enc / a.drop_dec, dropout.h) — a separate instantiation, so that the
// launches with masks in memory keep the code (and the registers) they had.
template <int U, bool DRAWN = false>
__device__ __forceinline__ void prologue_element(const PrologueArgs &a, int64_t idx) {
    const int H = a.H;
    if (idx < a.end[0]) {
        a.bsum[idx] = a.b_ih[idx] + a.b_hh[idx];
    } else if (idx < a.end[1]) {
        // the output head as ONE matrix (seq2seq_model.py:421-424 applies two bias-free Linears back to back):
        // head_wc[v, col] = sum_j W_h2o[v, j] . W_o2h[j, src(col)], columns in S order [e | ctx_t | ctx_v | h]
        // (W_o2h's own order is [e | h | ctx_t | ctx_v]).  Twenty products fetched per pass, as the composites below.
        const int64_t i = idx - a.end[0];
        const int v = (int)(i / (4 * H)), col = (int)(i % (4 * H));
        const int seg = col / H, k = col % H;
        const int src = (seg == 0 ? 0 : seg == 1 ? 2 * H : seg == 2 ? 3 * H : H) + k;
        const float *hrow = a.w_h2o + (int64_t)v * H, *ocol = a.w_o2h + src;
        float acc0 = 0.f, acc1 = 0.f;
        int j = 0;
        for (; j + U - 1 < H; j += U) {
            float x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { x[u] = hrow[j + u]; y[u] = ocol[(int64_t)(j + u) * 4 * H]; }
#pragma unroll
            for (int u = 0; u < U; u += 2) { acc0 = fmaf(x[u], y[u], acc0); acc1 = fmaf(x[u + 1], y[u + 1], acc1); }
        }
        for (; j < H; ++j) acc0 = fmaf(hrow[j], ocol[(int64_t)j * 4 * H], acc0);
        a.head_wc[i] = acc0 + acc1;
    } else if (idx < a.end[2]) {
        const int64_t i = idx - a.end[1];
        const int64_t per = (int64_t)4 * a.He * a.E, nw = (int64_t)a.D * per;
        // wih_t: the same weights column-major per direction with the bias sum as one more column,
        // [dir][E + 1][4He]: the A operand of the first encoder layer's own input projection (lstm_encoder.hip;
        // its B operand carries a row of ones)
        if (i < nw) {
            const float v = (i < per) ? a.w_ih_f[i] : a.w_ih_r[i - per];
            a.wih_stack[i] = v;
            const int dir = i >= per, r = (int)((i - dir * per) / a.E), e = (int)((i - dir * per) % a.E);
            a.wih_t[((int64_t)dir * (a.E + 1) + e) * 4 * a.He + r] = v;
        } else {
            const int dir = (int)((i - nw) / (4 * a.He)), r = (int)((i - nw) % (4 * a.He));
            a.wih_t[((int64_t)dir * (a.E + 1) + a.E) * 4 * a.He + r] =
                dir ? a.enc_b_ih_r[r] + a.enc_b_hh_r[r] : a.enc_b_ih_f[r] + a.enc_b_hh_f[r];
        }
    } else if (idx < a.end[3]) {
        a.dwc[idx - a.end[2]] = 0.f;
    } else if (idx < a.end[4]) {
        const int64_t i = idx - a.end[3];
        if (DRAWN && a.drop_enc.on) {    // (four rows, column): one Philox call, four gathers (dropout.h)
            const int64_t rq = i / a.E;
            const int d = (int)(i % a.E);
            const DropSpec ds = drop_spec_here(a.drop_enc);
            const uint32_t bits = drop_quad_bits(ds, kDropSegEnc, (uint64_t)i);
#pragma unroll 1                         // one row at a time: this launch runs on 63 VGPRs (conv.hip)
            for (int j = 0; j < 4; ++j) {
                const int64_t row = 4 * rq + j;
                if (row < a.BL) {
                    const int64_t t = a.commands[row];
                    const float keep = ((bits >> j) & 1u) ? ds.scale : 0.f;
                    a.xe[row * a.E + d] = (t >= 0 && t < a.Vi) ? a.enc_emb[t * a.E + d] * keep : 0.f;
                    a.mask_enc_out[row * a.E + d] = keep;        // for the embedding gradient of the backward pass
                }
            }
        } else {
            const int64_t row = i / a.E;
            const int d = (int)(i % a.E);
            const int64_t t = a.commands[row];
            float v = (t >= 0 && t < a.Vi) ? a.enc_emb[t * a.E + d] : 0.f;
            if (a.mask_enc) v *= a.mask_enc[i];
            a.xe[i] = v;
        }
    } else if (idx < a.end[5]) {
        const int64_t i = idx - a.end[4];
        if (DRAWN && a.drop_dec.on) {
            const int64_t rq = i / H;
            const int d = (int)(i % H);
            const DropSpec ds = drop_spec_here(a.drop_dec);
            const uint32_t bits = drop_quad_bits(ds, kDropSegDec, (uint64_t)i);
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const int64_t row = 4 * rq + j;
                if (row < a.BT) {
                    const int64_t t = a.targets[row];
                    const float keep = ((bits >> j) & 1u) ? ds.scale : 0.f;
                    a.S[row * 4 * H + d] = (t >= 0 && t < a.V) ? a.dec_emb[t * H + d] * keep : 0.f;
                    a.mask_dec_out[row * H + d] = keep;
                }
            }
        } else {
            const int64_t row = i / H;
            const int d = (int)(i % H);
            const int64_t t = a.targets[row];
            float v = (t >= 0 && t < a.V) ? a.dec_emb[t * H + d] : 0.f;
            if (a.mask_dec) v *= a.mask_dec[i];
            a.S[row * 4 * H + d] = v;
        }
    } else if (idx < a.end[6]) {
        const int64_t i = idx - a.end[5];
        const int row = (int)(i / (3 * H)), col = (int)(i % (3 * H));
        float v = 0.f;
        if (row < 4 * H) v = a.w_ih_dec[i];
        else if (a.cond && col >= H && col < 2 * H) v = a.w_q2k[(int64_t)(row - 4 * H) * 2 * H + col];
        a.wcat5[i] = v;
    } else if (idx < a.end[7]) {
        a.zero_extra[idx - a.end[6]] = 0.f;
    } else if (idx < a.end[8]) {
        decoder_image_element(a.img, (int)(idx - a.end[7]));
    } else if (idx < a.end[9]) {
        const int i = (int)(idx - a.end[8]), He = a.He, nt = 4 * He / a.enc_rows;
        const int j = i % nt, k = (i / nt) % He, r = (i / (nt * He)) % a.enc_rows, dir = i / (4 * He * He);
        a.enc_image[i] = (dir ? a.enc_w_hh_r : a.enc_w_hh_f)[(int64_t)(j + r * nt) * He + k];
    } else if (idx < a.end[10]) {
        const int i = (int)(idx - a.end[9]);
        a.conv_img[i] = conv_image_element(a.conv_w[0], a.conv_w[1], a.conv_w[2], a.cC, a.cCo, a.cK3, i);
    } else {
        // composite weights: out[r, c] = sum_h A[r, col0 + h] * Wk[h, c].  Consecutive threads take consecutive c
        // (coalesced Wk rows, A broadcast); eight products are fetched before they are added, so the loop is
        // not a chain of load latencies
        const float *__restrict__ A, *__restrict__ Wk;
        float *out;
        int lda, col0, N, i;
        if (idx < a.end[11]) { i = (int)(idx - a.end[10]); A = a.w_ih_dec; lda = 3 * H; col0 = 2 * H; Wk = a.w_key_vis; N = a.F; out = a.w_sk; }
        else if (idx < a.end[12]) { i = (int)(idx - a.end[11]); A = a.w_ih_dec; lda = 3 * H; col0 = H; Wk = a.w_key_txt; N = a.He; out = a.w_ck; }
        else { i = (int)(idx - a.end[12]); A = a.w_q2k; lda = 2 * H; col0 = H; Wk = a.w_key_txt; N = a.He; out = a.w_2kk; }
        const int r = i / N, c = i - r * N;
        // the gate images U = memory . (W_ih[:, ctx] . W_key)^T come out UNIT-major (column 4 unit + gate): the
        // decoder's column sums read the four gates of a unit as one 16-byte word
        const int src = (idx < a.end[12]) ? (r & 3) * H + (r >> 2) : r;
        const float *arow = A + (int64_t)src * lda + col0;
        const float *wcol = Wk + c;
        // twenty products fetched per pass (forty loads in flight): at eight the 100-deep dot was thirteen dependent
        // round trips to L2 and the longest chain of the whole launch
        float acc0 = 0.f, acc1 = 0.f;
        int h = 0;
        
        for (; h + U - 1 < H; h += U) {
            float x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { x[u] = arow[h + u]; y[u] = wcol[(int64_t)(h + u) * N]; }
#pragma unroll
            for (int u = 0; u < U; u += 2) { acc0 = fmaf(x[u], y[u], acc0); acc1 = fmaf(x[u + 1], y[u + 1], acc1); }
        }
        for (; h + 3 < H; h += 4) {
            float x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { x[u] = arow[h + u]; y[u] = wcol[(int64_t)(h + u) * N]; }
            acc0 = fmaf(x[0], y[0], acc0); acc1 = fmaf(x[1], y[1], acc1);
            acc0 = fmaf(x[2], y[2], acc0); acc1 = fmaf(x[3], y[3], acc1);
        }
        for (; h < H; ++h) acc0 = fmaf(arow[h], wcol[(int64_t)h * N], acc0);
        out[i] = acc0 + acc1;
    }
}

}  // namespace gscan
