// Internal C++ interface between the kernels (one .hip per family) and the step sequencer.
#pragma once
#include "common.h"
#include "gscan_hip.h"
#include "dropout.h"

namespace gscan {

// gemm.hip
int gemm_f32(int M, int N, int K, float alpha, const float *a, int64_t sam, int64_t sak, const float *b,
             int64_t sbk, int64_t sbn, float beta, float *c, int64_t ldc, const float *bias, int act,
             const float *mask, int split_k, hipStream_t stream);

int gemm_f32_ex(int M, int N, int K, float alpha, const float *a, int64_t sam, int64_t sak, const float *b,
                int64_t sbk, int64_t sbn, float beta, float *c, int64_t ldc, const float *bias, int act,
                const float *mask, int split_k, float *asum, float *scratch, size_t scratch_floats, hipStream_t stream);

constexpr int kMaxGroup = 12;
struct GemmProblem {
    int M, N, K;
    float alpha, beta;
    const float *a; int64_t sam, sak;     // A(m,k) = a[m*sam + k*sak]
    const float *b; int64_t sbk, sbn;     // B(k,n) = b[k*sbk + n*sbn]
    float *c; int64_t ldc;
    const float *bias; int act;           // act: 0 none, 1 relu, 2 tanh, 3 relu-backward (zero where gate == 0)
    const float *mask, *gate;             // elementwise, same indexing as C
    int k_chunk, atomic;                  // K range per k-slice; split-K accumulates with atomics
    float *asum1, *asum2;                 // optional: += sum_k A(m,k) (bias gradients)
    int tiles_n, tiles_mn, nsplit;        // grid bookkeeping (the first workgroup id is in GemmGroup::tile_begin)
    uint32_t inv_mn, inv_in;              // floor(2^32 / d) + 1 for d = tiles_mn and for the inner tile count (tiles_n,
                                          // or tiles_m when N tiles run slowest): x / d = umulhi(x, inv) for x * d < 2^32
    int flags;                            // log2(floats per global load) of A | of B << 2 | 16: N tiles run slowest
    int tiles_m;
    int nf;                               // wide / macro tiles: 16-column MFMA fragments per workgroup tile (BN = 16 nf)
    float *slab;                          // macro tiles, atomic == 2: partial tiles [tile][slice][wave][4][4][4][64] (gemm_mt.hip)
    int split_ok;                         // the caller allowed split-K (beta = 1, no epilogue): launch() chooses the slices;
                                          // 2: another product of the launch adds into the same C (atomics, never slabs)
};
// tile_begin / xcd_per lead the kernel arguments as one contiguous header: a workgroup finds its problem with ONE
// batch of scalar loads and fetches that problem's descriptor with a second one (the scan used to walk the
// descriptors: six dependent scalar-load round trips, ~1500 cycles, before the first global load could be issued).
struct GemmGroup { int count; int tile_begin[kMaxGroup]; int xcd_per[kMaxGroup]; GemmProblem p[kMaxGroup]; };

int gemm_macro_tile_mode();      // gemm.hip: 1 = every launch on the macro tiles (deterministic mode), 0 never, -1 by rule
// gemm_mt.hip: floats of one split-K partial tile (slab); a scratch region holds kGemmSlabs of them (GemmBatch::launch
// never plans more slices than fit)
size_t gemm_slab_floats();
constexpr int kGemmSlabs = 768;

// Builder for one grouped launch of independent products.
class GemmBatch {
public:
    void add(int M, int N, int K, const float *a, int64_t sam, int64_t sak, const float *b, int64_t sbk, int64_t sbn,
             float *c, int64_t ldc, float beta = 0.f, const float *bias = nullptr, int act = 0,
             const float *mask = nullptr, int split_k = 1, float *asum1 = nullptr, float *asum2 = nullptr,
             const float *gate = nullptr, float alpha = 1.f);
    // The product added last is NOT part of the algorithmic flop count of SURVEY.md 8(d) (composite weights, U
    // images computed in place of per-step context products that the decoder kernel is charged with): it still
    // counts as executed work.
    void overhead() { alg_flops_ -= last_flops_; last_flops_ = 0.0; }
    // Algorithmic flops of reference products that the launch obtains by cheaper algebra: counted as
    // algorithmic, not as executed.
    void credit(double alg_flops) { alg_flops_ += alg_flops; }
    // Scratch for the split-K slabs of the macro-tile kernel (gemm_mt.hip): with it split products are added in a fixed
    // order (bitwise reproducible); without it they fall back to float atomics.  One region per launch IN FLIGHT:
    // launches that may overlap on different streams must not share one.  Handing a launch scratch puts it on the macro
    // tiles whatever its size; the training step does so in deterministic mode only (step.hip).
    void scratch(float *ptr, size_t floats) { scratch_ = ptr; scratch_floats_ = floats; }
    // This launch on the macro tiles whatever the rule says (split products without scratch: float atomics).
    void prefer_macro_tiles() { force_mt_ = true; }
    int launch(hipStream_t stream);
private:
    GemmGroup grp_{};
    int tiles_ = 0;
    double flops_ = 0.0, alg_flops_ = 0.0, last_flops_ = 0.0;
    bool bad_ = false, force_mt_ = false;
    float *scratch_ = nullptr;
    size_t scratch_floats_ = 0;
    int launch_macro_tiles(hipStream_t stream);
};

// decoder.hip
constexpr int kDecThreads = 512;   // 8 waves: 2 per SIMD, 256-VGPR budget
constexpr int kDecPairs = 256;     // lane pairs; a weight row lives in one pair
struct DecoderGeometry { int rows, slots, k0; int64_t image_floats; };
DecoderGeometry decoder_geometry(int H, bool cond);
// Register images of the decoder weights (written once per step, by the prologue kernel):
//   fwd/bwd image[((f/4)*512 + tid)*4 + f%4], f = slot*K0 + i = element i of the half-row that thread tid keeps in slot `slot`
//     (task r = slot*256 + decoder_pair_of(tid), position kk = decoder_half_of(tid)*K0 + i along the dot; gate rows
//     of the forward image unit-major: decoder_gate_row); forward: row q of block sg of
//     [W_hh (4 blocks) | W_query_text | W_q2k[:, :H] or W_query_vis | W_query_vis]; backward: column q of block sg;
// Lane pairs of the decoder kernels: lane s and lane 7-s of every group of eight lanes (one DPP row_half_mirror apart)
// hold the two halves of a weight row.  pair = 4 (tid / 8) + min(s, 7 - s), half = s / 4.
__host__ __device__ inline int decoder_pair_of(int tid) { const int s = tid & 7; return 4 * (tid >> 3) + (s < 4 ? s : 7 - s); }
__host__ __device__ inline int decoder_half_of(int tid) { return (tid >> 2) & 1; }
// Forward image only: the gate rows are handed out UNIT-major so that the four gates of a hidden unit meet in one
// quad of lanes (decoder.hip, gate_lane): task r < 4H of slot 0 (r < 256) is W_hh row (gate r % 4, unit r / 4); of
// slot 1 it is (gate 3 - r % 4, unit 64 + (r - 256) / 4) — the upper quad of a group of eight sees its pairs mirrored.
__host__ __device__ inline int decoder_gate_row(int r, int H) {
    const int slot = r / kDecPairs, pr = r % kDecPairs, unit = 64 * slot + pr / 4, gate = slot ? 3 - pr % 4 : pr % 4;
    return gate * H + unit;
}
struct DecoderImageArgs {
    const float *w_hh, *w_qt, *w_qv, *w_q2k, *w_o2h;
    float *fwd_image, *bwd_image;
    int H, cond, slots, k0;
};
#ifdef __HIPCC__
// element e of the concatenation [fwd image | bwd image]
__device__ __forceinline__ void decoder_image_element(const DecoderImageArgs &a, int e) {
    const int H = a.H, total = a.slots * a.k0 * kDecThreads;
    // every image is stored in 16-byte groups: element i of thread tid sits at ((i / 4) * 512 + tid) * 4 + i % 4
    const bool bwd = e >= total;
    const int x = bwd ? e - total : e;
    const int tid = (x >> 2) % kDecThreads, f = 4 * (x / (4 * kDecThreads)) + (x & 3), i = f % a.k0, s = f / a.k0;
    const int r = s * kDecPairs + decoder_pair_of(tid), kk = decoder_half_of(tid) * a.k0 + i;   // kk: position along the dot
    float v = 0.f;
    const int rows = (a.cond ? 7 : 6) * H;
    if (r < rows && kk < H) {
        const int sg = r / H, q = r % H;                                               // block, index inside it
        if (!bwd) {          // row q of block sg, element kk
            if (sg < 4) v = a.w_hh[(int64_t)decoder_gate_row(r, H) * H + kk];
            else if (sg == 4) v = a.w_qt[(int64_t)q * H + kk];
            else if (sg == 5) v = a.cond ? a.w_q2k[(int64_t)q * 2 * H + kk] : a.w_qv[(int64_t)q * H + kk];
            else v = a.w_qv[(int64_t)q * H + kk];
        } else {             // column q of block sg, element (row) kk
            if (sg < 4) v = a.w_hh[(int64_t)(sg * H + kk) * H + q];
            else if (sg == 4) v = a.w_qt[(int64_t)kk * H + q];
            else if (sg == 5) v = a.cond ? a.w_q2k[(int64_t)kk * 2 * H + q] : a.w_qv[(int64_t)kk * H + q];
            else v = a.w_qv[(int64_t)kk * H + q];
        }
    }
    (bwd ? a.bwd_image : a.fwd_image)[x] = v;
}
#endif

// elementwise.hip
int embed_rows(const int64_t *tok, const float *table, int vocab, const float *mask, int rows, int D, float *out,
               int64_t ldo, hipStream_t stream);
int embed_grad(const int64_t *tok, const float *g, int64_t ldg, const float *mask, int rows, int D, int vocab,
               int pad, float *dtable, hipStream_t stream, float *part = nullptr);   // part: embed_grad_partial_floats, fixed-order sums
size_t embed_grad_partial_floats(int rows, int D, int vocab);
struct PrologueArgs {
    const float *b_ih, *b_hh, *w_o2h, *w_h2o, *w_ih_f, *w_ih_r, *enc_emb, *dec_emb, *mask_enc, *mask_dec;
    const int64_t *commands, *targets;
    float *bsum, *head_wc, *wih_stack, *wih_t, *dwc, *xe, *S, *wcat5, *zero_extra;
    const float *w_ih_dec, *w_q2k;
    int cond;
    int64_t zero_extra_count;
    int H, He, E, D, BL, BT, Vi, V;
    // drop_enc.on / drop_dec.on: the embedding dropout is drawn in the gather (dropout.h); segments 4 / 5 then have one
    // element per (FOUR rows, column), mask_enc / mask_dec are not read and the keep values go to mask_enc_out /
    // mask_dec_out ([BL,E], [BT,H]: the embedding gradients of the backward pass read them there)
    DropSpec drop_enc, drop_dec;
    float *mask_enc_out, *mask_dec_out;
    int64_t end[14];
    DecoderImageArgs img;
    // seg 9: register image of the encoder's recurrent weights, [dir][r][k][thread] (lstm_encoder.hip)
    const float *enc_w_hh_f, *enc_w_hh_r;
    const float *enc_b_ih_f, *enc_b_hh_f, *enc_b_ih_r, *enc_b_hh_r;      // first layer: folded into wih_t (seg 2)
    float *enc_image;
    int enc_rows;                          // weight rows per thread
    // seg 10: [tap][ch][o] image of the three convolution kernels (conv.hip)
    const float *conv_w[3];
    float *conv_img;
    int cC, cCo, cK3;
    // seg 11-13: composite weights, so that the gate images U = memory . (.)^T of the attention memories need no
    // second level of products: w_sk [4H, F] = W_ih[:, 2H:3H] . W_key_vis, w_ck [4H, He] = W_ih[:, H:2H] . W_key_text,
    // w_2kk [H, He] = W_q2k[:, H:2H] . W_key_text
    const float *w_key_vis, *w_key_txt;
    float *w_sk, *w_ck, *w_2kk;
    int F;
};
int step_prologue(const PrologueArgs &args, hipStream_t stream);
// The head's weight gradients from d Wc = dlogits^T . S ([V, 4H], S order): g_w_o2h += W_h2o^T . dWc (columns back in
// the reference's order), g_w_h2o += dWc . W_o2h^T; and, in further workgroups of the same launch, the energy-vector
// gradients: the per-row sums of the decoder's reverse kernel added up over the batch in a fixed order.
int head_grad_finish(const float *dwc, const float *w_h2o, const float *w_o2h, float *g_w_o2h, float *g_w_h2o, int H, int V,
                     hipStream_t stream, const float *dv_t_rows = nullptr, const float *dv_v_rows = nullptr, int B = 0,
                     float *g_v_t = nullptr, float *g_v_v = nullptr);
int adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
              float beta2, float eps, float lr_decay, float lr_decay_steps, int64_t step, const float *grad_scale,
              const float *dev_scalars, int zero_grad, hipStream_t stream);   // zero_grad bit 1: grad_scale divides
int adam_step_masks(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                    float beta2, float eps, float lr_decay, float lr_decay_steps, int64_t step, const float *grad_scale,
                    int zero_grad, float *mask_out, const size_t (&nm)[3], const float (&pm)[3], uint64_t seed,
                    uint64_t stream_id, hipStream_t stream);
void adam_scalars(float lr, float beta1, float beta2, float lr_decay, float lr_decay_steps, int64_t step,
                  float *step_size, float *inv_sqrt_bc2);
int dropout_mask(float *out, size_t n, float p, uint64_t seed, uint64_t stream_id, hipStream_t stream);
int dropout_masks_kernel_layout(float *cnn, float *enc, float *dec, int B, int M, int Co, int L, int E, int T, int H,
                                float p_cnn, float p_enc, float p_dec, uint64_t seed, uint64_t stream_id, hipStream_t stream);
int dropout_masks(float *out, const size_t (&n)[3], const float (&p)[3], uint64_t seed, uint64_t stream_id,
                  const uint64_t *dev_stream_id, hipStream_t stream);
// in-kernel timeline (common.h): one setter per translation unit with kernels
int trace_set_gemm(unsigned long long *buf);
int trace_set_gemm_mt(unsigned long long *buf);
int trace_set_gemm_ws(unsigned long long *buf);
int trace_set_elementwise(unsigned long long *buf);
int trace_set_loss(unsigned long long *buf);
int trace_set_lstm_encoder(unsigned long long *buf);
int trace_set_decoder(unsigned long long *buf);
int trace_set_decoder_p1(unsigned long long *buf);
int trace_set_decoder_p2(unsigned long long *buf);
int trace_set_decoder_p3(unsigned long long *buf);
int decoder_run_part0(bool backward, int B, int H, bool cond, const struct DecoderArgs &a, hipStream_t stream);
int trace_set_decoder_any(unsigned long long *buf);
int trace_set_attention_grad(unsigned long long *buf);

// conv.hip: the world encoder (cnn_model.py:22-36) and its weight gradients, input-sparse
#ifdef __HIPCC__
__device__ __forceinline__ int conv_ksize(int conv, int K3) { return conv == 0 ? 1 : (conv == 1 ? 5 : K3); }
__device__ __forceinline__ int conv_tap0(int conv, int K3) { return conv == 0 ? 0 : (conv == 1 ? 1 : 26); }
#endif
// [tap][ch][o] image of the three kernels (taps of conv_1, conv_2, conv_3 one after the other), rows padded with
// zeros to a multiple of 32 floats so that a row starts on a 128-byte line: element i of (1 + 25 + K3^2) * C * CoP.
// Written by the step prologue, read by the forward kernel as coalesced rows.
__host__ __device__ static inline int conv_row_floats(int Co) { return (Co + 31) / 32 * 32; }
static inline int64_t conv_image_floats(int C, int Co, int K3) { return (int64_t)(26 + K3 * K3) * C * conv_row_floats(Co); }
#ifdef __HIPCC__
__device__ __forceinline__ float conv_image_element(const float *w1, const float *w2, const float *w3, int C, int Co,
                                                    int K3, int i) {
    const int CoP = conv_row_floats(Co);
    const int o = i % CoP, ch = (i / CoP) % C, tg = i / (CoP * C);
    if (o >= Co) return 0.f;
    const int conv = tg < 1 ? 0 : (tg < 26 ? 1 : 2), k = conv_ksize(conv, K3), t = tg - conv_tap0(conv, K3);
    const float *w = conv == 0 ? w1 : (conv == 1 ? w2 : w3);
    return w[(o * C + ch) * k * k + t];                      // t = kh * k + kw, the reference's own tap order
}
#endif
int conv_weight_image(const float *const (&w)[3], int C, int Co, int K3, float *img, hipStream_t stream);
int world_conv_forward(const void *world, int world_is_u8, const float *img, const float *const (&b)[3],
                       const float *mask, int B, int G, int C, int Co, int K3, float *feat, hipStream_t stream,
                       const DropSpec *drop = nullptr);
// the step prologue and the world encoder in one launch (conv.hip); -1: the shape does not fit, nothing was launched
constexpr int kFusedMaxFlags = 512;
int prologue_world_forward(const PrologueArgs &pa, const void *world, int world_is_u8, const float *const (&b)[3],
                           const float *mask, int B, int G, int C, int Co, int K3, float *feat, uint32_t *flags,
                           hipStream_t stream, const DropSpec *drop = nullptr);
size_t world_conv_backward_scratch_floats(int B, int G, int C);
int world_conv_lists(const void *world, int world_is_u8, int B, int G, int C, float *scratch, hipStream_t stream);
size_t world_conv_bias_partial_floats(int B, int G, int Co);
int world_conv_backward(const float *dfeat, int B, int G, int C, int Co, int K3, float *scratch,
                        float *const (&gw)[3], float *const (&gb)[3], hipStream_t stream, float *bias_part = nullptr);
int trace_set_conv(unsigned long long *buf);

// loss.hip
int step_losses(const float *logp, const int64_t *targets, const float *aux, const int64_t *pos, int B, int T, int V,
                int M, int pad, float *stats, float *dlogp, float *daux, hipStream_t stream);
int loss_seeds(const float *stats, float w, int auxiliary, float *seeds, hipStream_t stream);
int sequence_nll(const float *logp, const int64_t *targets, int B, int T, int V, int pad, float *loss_sum,
                 float *count, float *dlogp, hipStream_t stream);
int position_nll(const float *aux, const int64_t *pos, int B, int M, float *loss_sum, float *daux,
                 hipStream_t stream);
int sequence_metrics(const float *logp, const int64_t *targets, int B, int T, int V, int pad, float *out3,
                     hipStream_t stream);

// Hidden sizes with compiled kernels (the recurrent weights live in registers, so the size is a template parameter):
// EVERY multiple of 4 — decoder / keys kernels up to 100 (five column quads per unit must fit 128 threads, 7 H^2
// weights the register file), encoder up to 128.
// The decoder kernels of all sizes are ONE source compiled as four translation units (-DGSCAN_DEC_PART=0..3, build.py:
// 80 s of a 95 s build were this one file on one core): part k instantiates the sizes of list k and defines
// decoder_run_part<k>; part 0 also holds the host functions they share.
#ifdef GSCAN_DEC_HIDDEN_ONLY      // development builds (assembly inspection): one decoder hidden size only, one part
#define GSCAN_DEC_HIDDEN_PART0(X) X(GSCAN_DEC_HIDDEN_ONLY)
#define GSCAN_DEC_HIDDEN_PART1(X)
#define GSCAN_DEC_HIDDEN_PART2(X)
#define GSCAN_DEC_HIDDEN_PART3(X)
#else
#define GSCAN_DEC_HIDDEN_PART0(X) X(4) X(20) X(36) X(52) X(68) X(84) X(100)
#define GSCAN_DEC_HIDDEN_PART1(X) X(8) X(24) X(40) X(56) X(72) X(88)
#define GSCAN_DEC_HIDDEN_PART2(X) X(12) X(28) X(44) X(60) X(76) X(92)
#define GSCAN_DEC_HIDDEN_PART3(X) X(16) X(32) X(48) X(64) X(80) X(96)
#endif
#define GSCAN_DEC_HIDDEN_SIZES(X) GSCAN_DEC_HIDDEN_PART0(X) GSCAN_DEC_HIDDEN_PART1(X) GSCAN_DEC_HIDDEN_PART2(X) GSCAN_DEC_HIDDEN_PART3(X)
#define GSCAN_DEC_HIDDEN_LIST "multiples of 4 from 4 to 100"
#define GSCAN_ENC_HIDDEN_SIZES(X) X(4) X(8) X(12) X(16) X(20) X(24) X(28) X(32) X(36) X(40) X(44) X(48) X(52) X(56) X(60) X(64) X(68) X(72) X(76) X(80) X(84) X(88) X(92) X(96) X(100) X(104) X(108) X(112) X(116) X(120) X(124) X(128)
#define GSCAN_ENC_HIDDEN_LIST "multiples of 4 from 4 to 128"

// lstm_encoder.hip
bool hidden_size_supported(int h);
// x != NULL: the kernel projects its own input x [B,L,E] through W_ih (+ b_ih) instead of reading gx (first layer)
// w_ih_t: the input weights column-major per direction plus the bias sums, [dir][E + 1][4He] (prologue seg 2): what
// the kernel multiplies by
struct EncInput { const float *x; int E; const float *w_ih_f, *b_ih_f, *w_ih_r, *b_ih_r; const float *w_ih_t; };
int encoder_lstm_forward(int B, int L, int He, int D, const float *gx, const int32_t *lengths, const float *w_hh_f,
                         const float *b_hh_f, const float *w_hh_r, const float *b_hh_r, float *out, float *h_final,
                         float *gates, float *cells, float *hprev, const float *w_image, hipStream_t stream,
                         float *hcat = nullptr, const float *hcat_mask = nullptr, const EncInput *input = nullptr);
bool encoder_fast_supported(int He, int L, int E);   // false: the streaming any-size kernels run (no weight images)
int encoder_rows_per_thread(int He);   // rows of W_hh a thread of the forward kernel keeps (layout of its image)
int encoder_weight_image(const float *w_hh_f, const float *w_hh_r, int He, int D, float *image, hipStream_t stream);
int encoder_lstm_backward(int B, int L, int He, int D, const int32_t *lengths, const float *w_hh_f,
                          const float *w_hh_r, const float *gates, const float *cells, const float *d_out,
                          const float *d_h_final, float *delta, hipStream_t stream, int d_out_row = 0,
                          int d_out_dir = 0, const float *d_out_mask = nullptr);

// decoder.hip (geometry and weight images: declared above, next to the prologue that writes the images)
constexpr int kHeadChunk = 32;     // target steps per pass of the fused output head (two 16-row MFMA tiles)

struct DecoderArgs {
    int T, L, M;                       // target steps, command memories, grid memories (G*G)
    const int32_t *cmd_lengths;        // [B]
    const float *pk_t, *u_t, *u2_t;    // [B,L,H] [B,L,4H] [B,L,H]
    const float *pk_v, *u_v;           // [B,M,H] [B,M,4H]
    const float *ge;                   // [B,T,4H] embedding part of the gates + both biases
    const float *w_image;              // register image of the recurrent weights (decoder_image_element)
    const float *b_q2k, *v_t, *v_v;
    // the same weights in the reference's own layouts, for the any-shape kernels (decoder_any.hip): lstm.weight_hh
    // [4H,H], lstm.weight_ih [4H,3H], the attentions' query layers [H,H], queries_to_keys.weight [H,2H] or NULL
    const float *any_w_hh, *any_w_ih, *any_w_qt, *any_w_qv, *any_w_q2k;
    int any_use_u;                      // set by decoder_run_any
    int any_lds_floats;                 // set by decoder_run_any: LDS floats the launch may use (any_residency's limit; 0: no residency)
    float *any_wcat, *any_wcat_stream;  // [6H, H] = [W_hh ; W_q2k[:, :H] or W_query_vis ; W_query_text], row-major and trip-major
                                        // (anyshape.h): written by the forward launch of decoder_run_any into the resident kernels' two
                                        // image slots (unused on this path), read by the reverse / the forward streaming kernel
    float *hprev;                      // [B,T,H]  hprev[b,0] = h0 (= c0 unless c0 is given) on entry; kernel fills t+1
    const float *c0;                   // [B,H] initial cell state, or NULL for c0 = h0 (seq2seq_model.py:494-504)
    float *h_last;                     // [B,H] h after the last step, or NULL
    float *s;                          // [B,T,4H] = [e | ctx_text | ctx_vis | h_t]; kernel fills 3 parts
    float *cells, *gates;              // [B,T,H] [B,T,4H]
    float *alpha_c, *alpha_s;          // [B,T,L] [B,T,M]
    float *q2, *qt, *qv;               // [B,T,H] conditional query, projected text / visual query
    float *att_sum;                    // [B,M] sum_t alpha_s (auxiliary head input)
    // fused output head (seq2seq_model.py:421-424, model.py:203): forward epilogue / backward prologue
    int V;
    const float *head_wc;              // [V,4H] the head as ONE matrix: W_h2o . W_o2h, columns in S order (step prologue)
    float *logits;                     // [B,T,V]
    float *logp_saved, *logp_out;      // [B,T,V] workspace copy and the caller's output
    float *aux_saved, *aux_out;        // [B,M] log_softmax(att_sum), or NULL without the auxiliary task
    // backward only
    const float *dlogp, *daux, *seeds; // incoming gradients ([B,T,V], [B,M] or NULL) and optional device scales [2]
    float *dlogits;                    // [B,T,V] saved for the head's weight gradients
    float *ds;                         // [B,T,4H] head gradient wrt [e | ctx_text | ctx_vis | h_t] (written first)
    // fused training loss (model.py:147-164): forward leaves per-row partial sums, backward starts from them
    const int64_t *targets, *positions;   // [B,T]; [B] or NULL
    int pad_tgt, B;
    float *row_stats;                  // [B,4] = [sum NLL, live tokens, aux NLL, 1] of the row, or NULL
    int nll_mode;                      // backward: seed from row_stats / targets instead of dlogp / daux / seeds
                                       // (1: mean loss as the reference's, 2: sum loss for the data-parallel step)
    float w_aux;                       // weight of the auxiliary loss (train.py:105-107)
    float *stats_out, *seeds_out;      // [4] batch sums, [3] = [1/tokens, w/rows, loss]; written by workgroup 0
    float *delta, *dqt, *dqv;          // [B,T,5H] = [gate deltas (4H) | dzq (H)], [B,T,H], [B,T,H]
    float *dpk_t, *dpk_v;              // [B,L,H] [B,M,H]  score-path key gradients
    float *dv_t, *dv_v;                // [B,H] energy-vector gradients of every row (summed by head_grad_finish's launch)
    float *dh0;                        // [B,H] gradient wrt the bridge pre-activation
    float *stamps;                     // diagnostics: [2][16] per-phase cycle sums of workgroup 0, or NULL
    // greedy decoding (forward kernel, GREEDY instantiation; selected by tokens_out != NULL): T = step limit,
    // ge = [V,4H] table Emb . W_ih[:, :H]^T + biases, hprev = h0 [B,H]
    const float *dec_emb;              // [V,H]
    int sos, eos;
    int64_t *tokens_out;               // [B,T] tokens produced (the <EOS> included)
    int32_t *steps_out;                // [B] steps taken
};
bool decoder_hidden_supported(int h);
size_t decoder_lds_bytes(int H, int L, int M, int V, bool cond, bool backward);
int decoder_run(bool backward, int B, int H, bool cond, const DecoderArgs &a, hipStream_t stream);
// decoder_any.hip: the same launches for any hidden size / number of memories (weights streamed, memories in global)
int decoder_run_any(bool backward, int B, int H, bool cond, const DecoderArgs &a, hipStream_t stream);
// does decoder.hip have kernels for this shape (a compiled hidden size, <= 64 memories per attention, LDS fits)?
bool decoder_fast_supported(int H, int L, int M, int V, bool cond);
bool decoder_any_uses_gate_images(int H, int L, int M);   // decoder_any.hip: the streaming kernels read U images too

// attention_grad.hip: value path of both attentions + key layers + bridge, one workgroup per batch row
struct KeysBackwardArgs {
    int T, L, M, He, F;
    const float *alpha_c, *alpha_s, *ds;   // [B,T,L] [B,T,M] [B,T,4H] (columns H..3H = d ctx_text | d ctx_vis)
    float *dpk_t, *dpk_v;                  // [B,L,H] [B,M,H]  in: score path, out: score + value path
    const float *dh0;                      // [B,H]
    const float *w_kt, *w_kv, *w_b;        // [H,He] [H,F] [H,He]
    const float *feat, *mask;              // [B,M,F]; mask may be NULL
    float mask_scale;                      // mask == NULL and != 0: dropout was drawn in the world encoder (dropout.h): the
                                           // gradient passes where feat != 0, times 1 / (1 - p)
    float *denc, *dhN, *dfeat;             // [B,L,He] [B,He] [B,M,F]
    int value_path_only;                   // != 0: stage 1 only (dPK totals); the key layers and the bridge follow as a GEMM launch
};
int keys_backward(int B, int H, const KeysBackwardArgs &a, hipStream_t stream);

// attention_grad.hip: the decoder's pre-activation gradients summed over TIME per attention memory,
//   G_text[b,l,:] = sum_t alpha_text[b,t,l] * [delta | dzq][b,t,:]      G_vis[b,m,:] = sum_t alpha_vis[b,t,m] * delta[b,t,:]
// (a context is alpha . PK, so everything the LSTM input and the conditional query hand back to the contexts reaches PK
// and the context columns of W_ih / W_q2k through these sums: step.hip, attention_time_reduced)
struct AlphaReduceArgs {
    int T, L, M, wt, wv, ldx;              // columns of G_text (5H conditional, else 4H) / of G_vis (4H); row stride of x
    const float *alpha_c, *alpha_s, *x;    // [B,T,L] [B,T,M] [B,T,ldx]
    float *g_t, *g_v;                      // [B,L,wt] [B,M,wv]
};
int alpha_reduce(int B, const AlphaReduceArgs &a, hipStream_t stream);

// comm.hip: RCCL all-reduce on the caller's stream (run-time binding)
int comm_available();                  // 0: RCCL is loadable in this process (no device call, nothing collective)
int comm_unique_id(void *id_host);
int comm_init(void **comm, int nranks, int rank, const void *id_host);
int comm_allreduce_f32(void *comm, float *buf, size_t n, hipStream_t stream);
int comm_destroy(void *comm);
int comm_count(void *comm, int *nranks);

// probe.hip
enum ProbeId { P_DECODER_FWD = 0, P_DECODER_BWD, P_ENCODER_FWD, P_ENCODER_BWD, P_GEMM, P_CONV_FWD, P_CONV_BWD, P_KEYS_BWD, P_COUNT };
struct ProbeScope {
    // flops: executed by the launch (2 M N K, padding excluded); alg_flops: its share of SURVEY.md 8(d)'s
    // algorithmic count (negative: same as flops)
    ProbeScope(int id, hipStream_t st, double flops, double alg_flops = -1.0);
    ~ProbeScope();
    int id_;
    hipStream_t st_;
};
int probe_enable(int on);
bool probe_stamps_enabled();
int early_gradients_wait(hipStream_t stream);      // step.hip: the backward pass's first leaf stream is done
int set_early_allreduce(void *comm, float *buf, size_t n);
int probe_reset();
int probe_read(const char *name, double *total_ms, double *flops, double *alg_flops, int64_t *launches);

// step.hip
struct WorkspaceSlot { const char *name; int64_t offset, count; };
struct Workspace {
    int64_t feat, pkv, uv, xe, gx, enc_out, hN, enc_gates, enc_cells, enc_hprev, pkt, ut, u2t, bsum, hprev, S,
        ge, cells, gates, alpha_c, alpha_s, q2, qt, qv, att_sum, logits, logp_saved, aux_saved, row_stats, dlogits,
        dS, datt, delta, dzq, dqt, dqv, dpk_t, dpk_v, dv_t, dv_v, dh0, denc, dhN, enc_delta, dxe, dfeat, stamps,
        dwc, wih_stack, wih_t, w_sk, w_ck, w_2kk, dec_w_fwd, dec_w_bwd, enc_w_image, conv_img, conv_flags, conv_lists, wcat5,
        deep_gates, deep_cells, deep_hprev, deep_y, deep_dy, deep_delta, deep_image,   // encoder layers below the last
        ge_table, head_wc,        // [V,4H] tables: greedy decoding's embedded gates; the composite head
        gemm_slabs_side, gemm_slabs_main, conv_bias_part, embed_part_dec, embed_part_enc,
        drawn_mask_enc, drawn_mask_dec,    // [B,L,E] [B,T,H]: embedding dropout drawn in the gathers, kept for the backward pass
        g_t, g_v;                          // [B,L,5H] [B,M,4H]: time-reduced gate gradients per memory (long targets only)
    WorkspaceSlot slot[96];
    int nslots;
    int64_t total_floats;
};
int check_dims(const gscan_dims &d);
int workspace_layout(const gscan_dims &d, Workspace *ws);
int step_forward(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const gscan_masks &mk, float *w,
                 float *logp, float *aux_logp, hipStream_t st, const float *given_feat = nullptr,
                 const float *given_enc_out = nullptr, const float *given_hN = nullptr);
struct NllSeed { float w_aux; bool sum; float *stats_out, *seeds_out; };   // backward of the training loss itself
int step_encode(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const gscan_masks &mk, float *w,
                hipStream_t st);
int step_decode_one(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const int64_t *tokens,
                    const float *h_in, const float *c_in, float *w, float *logits, float *h_out, float *c_out,
                    float *alpha_text, float *alpha_vis, hipStream_t st);
int step_greedy(const gscan_dims &d, int max_steps, const gscan_params &p, const gscan_batch &bt, float *w, int sos,
                int eos, int64_t *tokens, int32_t *steps, float *alpha_text, float *alpha_vis, float *att_sum,
                hipStream_t st);
int step_backward(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const gscan_masks &mk, float *w,
                  const float *dlogp, const float *daux, const float *seeds, const NllSeed *nll, const gscan_params &g,
                  hipStream_t st);
// forward + training loss + backward as one sequence
int step_train_nll(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const gscan_masks &mk, float *w,
                   float *logp, float *aux_logp, const NllSeed &nll, const gscan_params &g, hipStream_t st);

}  // namespace gscan
