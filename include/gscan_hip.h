/*
 * gscan_hip.h — C ABI of libgscan_hip.so, the MI355X (gfx950) implementation of the
 * training hot path of LauraRuis/multimodal_seq2seq_gSCAN.
 *
 * The reference has no native boundary: its hot path is the Python class surface
 * seq2seq/model.py:206-219 (Model.forward), :147-160 (get_loss), :162-170 (auxiliary head)
 * and the optimiser lines of seq2seq/train.py:110-113.  Each entry point below replaces the
 * arithmetic behind one of those call sites; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.  Plain pointers and sizes only: every pointer is a
 * DEVICE pointer unless the parameter name ends in _host; `stream` is a hipStream_t passed
 * as void*.  Every function returns 0 on success, non-zero on failure
 * (gscan_last_error() gives the text).  Nothing here allocates device memory or
 * synchronises the stream, so a caller may capture any sequence of calls in a hipGraph.
 *
 * Layouts are the reference's own: parameters are the tensors of Model.state_dict()
 * (row-major [out, in], LSTM gate order i,f,g,o); commands [B,L] and targets [B,T] are
 * int64 token ids; world is float32 — or uint8, the form the batcher ships — [B,G,G,C] indexed
 * [row][col][channel]; log-probabilities are float32 [B,T,V].
 */
#ifndef GSCAN_HIP_H
#define GSCAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSCAN_ABI_VERSION 14
#define GSCAN_MAX_ENC_LAYERS 4

/* Problem dimensions (names follow the reference's flags, seq2seq/__main__.py:21-102).
 * Accepted (gscan_workspace_bytes returns 0 and gscan_last_error says why otherwise): H 1..1024, He 1..2048, E 1..1024,
 * L and G*G up to 4096, K3 odd, up to GSCAN_MAX_ENC_LAYERS encoder layers, B*T*4H < 2^31.  Shapes outside what the
 * register/LDS-resident kernels are compiled for (H a multiple of 4 up to 100, He a multiple of 4 up to 128, at most 64
 * memories per attention, a target vocabulary V of at most 16 — one matrix-core tile of logits —, and a row's memories
 * within 160 KB of LDS) run on streaming kernels with the same results, several times slower;
 * gscan_decoder_kernel_family() says which family a shape takes.
 * Environment, read once per process: GSCAN_DETERMINISTIC=1 makes every sum that crosses workgroups fixed-order (bitwise
 * reproducible training steps, +6 % time; it changes the workspace size: set it before gscan_workspace_bytes). */
typedef struct gscan_dims {
    int32_t B;             /* rows in this batch                                   */
    int32_t L;             /* padded command length                                */
    int32_t T;             /* padded target length                                 */
    int32_t G;             /* grid side                                            */
    int32_t C;             /* num_cnn_channels                                     */
    int32_t Co;            /* cnn_hidden_num_channels                              */
    int32_t K3;            /* cnn_kernel_size (conv_3); conv_1 is 1, conv_2 is 5   */
    int32_t E;             /* embedding_dimension                                  */
    int32_t He;            /* encoder_hidden_size                                  */
    int32_t H;             /* decoder_hidden_size                                  */
    int32_t Vi;            /* input_vocabulary_size                                */
    int32_t V;             /* target_vocabulary_size                               */
    int32_t conditional;   /* conditional_attention                                */
    int32_t auxiliary;     /* auxiliary_task                                       */
    int32_t bidirectional; /* encoder_bidirectional                                */
    int32_t pad_in;        /* input_padding_idx                                    */
    int32_t pad_tgt;       /* target_pad_idx                                       */
    int32_t enc_layers;    /* num_encoder_layers (0 or 1: one layer; at most GSCAN_MAX_ENC_LAYERS) */
} gscan_dims;

/* One pointer per reference parameter (named_parameters() order, seq2seq/model.py:47-87).
 * The same struct addresses the gradients.  Entries that do not exist for a configuration
 * (w_q2k/b_q2k without conditional attention, *_rev when unidirectional) are NULL. */
typedef struct gscan_params {
    float *conv1_w, *conv1_b, *conv2_w, *conv2_b, *conv3_w, *conv3_b;
    float *vis_key_w, *vis_query_w, *vis_energy_w;
    float *enc_emb;
    float *enc_w_ih, *enc_w_hh, *enc_b_ih, *enc_b_hh;
    float *enc_w_ih_rev, *enc_w_hh_rev, *enc_b_ih_rev, *enc_b_hh_rev;
    float *bridge_w, *bridge_b;
    float *txt_key_w, *txt_query_w, *txt_energy_w;
    float *q2k_w, *q2k_b;
    float *dec_emb;
    float *dec_w_ih, *dec_w_hh, *dec_b_ih, *dec_b_hh;
    float *out2hid_w, *hid2out_w;
    /* encoder layers 1.. (nn.LSTM(num_layers=n), seq2seq_model.py:44-45): per layer weight_ih [4He, D*He],
     * weight_hh [4He, He], bias_ih, bias_hh, then the same four of the reverse direction; NULL where absent.
     * enc_w_* above are layer 0.  (In named_parameters() order these sit right behind layer 0; they are at the
     * end of this struct so that one-layer callers keep their field offsets.) */
    float *enc_deep[GSCAN_MAX_ENC_LAYERS - 1][8];
} gscan_params;

/* The reference's batch tuple (seq2seq/gSCAN_dataset.py:229-231) as device arrays. */
typedef struct gscan_batch {
    const int64_t *commands;     /* [B,L]                                           */
    const int32_t *cmd_lengths;  /* [B]  number of real tokens per command          */
    const float   *world;        /* [B,G,G,C] float32, or NULL when world_u8 is given */
    const int64_t *targets;      /* [B,T]                                           */
    const int64_t *target_positions; /* [B] flat grid cell of the target object (gSCAN_dataset.py:216-219),
                                      * or NULL; only the auxiliary loss of gscan_backward_nll reads it */
    const uint8_t *world_u8;     /* [B,G,G,C] the same tensor as bytes (Grid.encode writes uint8, minigrid.py:384): when
                                  * not NULL it is read INSTEAD of `world` and widened inside the kernels — 1 byte per
                                  * element crosses PCIe and HBM instead of 4 */
} gscan_batch;

/* Dropout.  Two forms:
 *  - in_kernel != 0 (ABI 13, the production mode: SURVEY.md 7 hard part 3): the kernels that apply dropout draw it
 *    themselves from a counter-based generator (Philox-4x32-10, key = seed, counter = [element | segment | stream_id]):
 *    no mask ever exists in memory.  p_cnn / p_enc / p_dec are the three drop probabilities (0 = none), the pointers
 *    cnn / enc / dec are ignored.  forward and backward of one step must be given the same seed and stream_id;
 *    gscan_dropout_masks_kernel_layout() writes the masks such a step uses (tests, host-mask parity mode).
 *  - in_kernel == 0: scaled masks in memory (0 or 1/(1-p)), or NULL for "no dropout" (eval mode / p = 0):
 *    cnn [B,G*G,3*Co], enc [B,L,E], dec [B,T,H] in batch-row order.  The caller draws them with gscan_dropout_mask() or
 *    supplies masks of its own (host-mask parity mode).
 * enc_deep (only with more than one encoder layer) is always a mask in memory. */
typedef struct gscan_masks {
    const float *cnn, *enc, *dec;
    const float *enc_deep;   /* [enc_layers-1, B, L, D*He]: inputs of encoder layers 1.. (nn.LSTM's inter-layer dropout) */
    int32_t in_kernel;
    float p_cnn, p_enc, p_dec;
    uint64_t seed, stream_id;
} gscan_masks;

int         gscan_abi_version(void);
const char *gscan_last_error(void);

/* Bytes of device scratch gscan_forward/gscan_backward need for `dims`.  The same buffer
 * must be passed to the backward call that follows a forward call (it holds the saved
 * activations).  Returns 0 if the dimensions are unsupported (see gscan_last_error). */
size_t gscan_workspace_bytes(const gscan_dims *dims);

/* Which decoder kernels `dims` runs on (ABI 14): 1 = the register/LDS-resident kernels (csrc/decoder.hip), 0 = the
 * streaming kernels (csrc/decoder_any.hip: any shape, several times slower), negative = unsupported dimensions (see
 * gscan_last_error).  bench.py prints it as config.decoder_kernels; the conditions are listed above gscan_dims. */
int gscan_decoder_kernel_family(const gscan_dims *dims);

/* Byte offset and float count of one named activation inside the workspace (for tests and
 * debugging: "feat", "enc_out", "pkt", "S", "gates", "delta", ...; see csrc/step.hip). */
int gscan_workspace_find(const gscan_dims *dims, const char *name, size_t *offset_bytes, size_t *count);

/* Model.forward (seq2seq/model.py:206-219): CNN + BiLSTM encoders, joint-attention LSTM
 * decoder over all T steps, output head, log_softmax.  Writes logp [B,T,V] and, when
 * dims->auxiliary, aux_logp [B,G*G] (= log_softmax of the summed visual attention,
 * model.py:166-170).  aux_logp may be NULL otherwise. */
int gscan_forward(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                  const gscan_masks *masks, void *workspace, float *logp, float *aux_logp, void *stream);

/* Reverse of gscan_forward: given d(loss)/d(logp) [B,T,V] and (optionally, may be NULL)
 * d(loss)/d(aux_logp) [B,G*G], ADDS d(loss)/d(parameter) into `grads` (the caller zeroes
 * them, as optimizer.zero_grad() does in seq2seq/train.py:113).  Replaces the autograd
 * replay behind loss.backward() (train.py:110). */
int gscan_backward(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                   const gscan_masks *masks, void *workspace, const float *dlogp, const float *daux_logp,
                   const gscan_params *grads, void *stream);

/* ---- greedy decoding (seq2seq/predict.py:82-115): encode once, then one decoder step per call ---- */

/* Model.encode_input (model.py:172-180) plus what predict.py computes once per example before its decoding loop
 * (:87-96): both encoders, the projected keys of both attentions and the bridge tanh(W h_N + b).  dims->T must be
 * 1; `workspace` (gscan_workspace_bytes of the same dims) then holds, under the names gscan_workspace_find knows,
 * "feat" [B,G*G,3Co], "enc_out" [B,L,He], "hN" [B,He], "pkv" [B,G*G,H], "pkt" [B,L,H] and "hprev" [B,H] = the
 * initial decoder state (h0 = c0).  batch->targets / target_positions are not read; masks is normally NULL. */
int gscan_encode(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                 const gscan_masks *masks, void *workspace, void *stream);

/* Model.decode_input = BahdanauAttentionDecoderRNN.forward_step (seq2seq_model.py:359-431) for all B rows, eval
 * mode, on the memories the last gscan_encode left in `workspace`:  tokens [B] -> logits [B,V] (un-normalised, as
 * the reference returns them), the new state (h_out, c_out [B,H]; may alias h_in / c_in) and the two attention
 * distributions alpha_text [B,L], alpha_vis [B,G*G].  Only batch->cmd_lengths is read from `batch`. */
int gscan_decode_step(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                      const int64_t *tokens, const float *h_in, const float *c_in, void *workspace, float *logits,
                      float *h_out, float *c_out, float *alpha_text, float *alpha_vis, void *stream);

/* Model.decode_input_batched (model.py:190-204), the second half of the reference's forward(): teacher-forced decoding
 * from encodings HANDED IN — encoded_situations [B,G*G,3Co] (ConvolutionalNet output), encoder_outputs [B,L,He]
 * (batch-major; the reference holds them time-major), hidden_states [B,He] — through the bridge, both key layers,
 * the decoder over all T steps, the output head and log_softmax.  Eval semantics (no dropout), inference only: no
 * backward call may follow.  batch->commands / world are not read.  Writes logp [B,T,V] and att_sum [B,G*G] (the
 * visual attention summed over the T steps, seq2seq_model.py:490).  `workspace` as for gscan_forward. */
int gscan_decode_batched(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                         const float *encoded_situations, const float *encoder_outputs, const float *hidden_states,
                         void *workspace, float *logp, float *att_sum, void *stream);

/* predict.py:82-115 for all B rows in ONE call: gscan_encode, then the persistent decoder kernel with the argmax fed
 * back in-kernel (eval mode) — every row decodes from <SOS> until it emits <EOS> or max_steps steps have run
 * (max_steps = max_decoding_steps + 1: the reference's `while token != eos and i <= max_decoding_steps`).
 * dims->T must be 1 (workspace of gscan_workspace_bytes(dims)).  Outputs: tokens [B,max_steps] (the tokens produced,
 * the final <EOS> included; entries behind a row's own steps are not written), steps [B], the attention rows of
 * every step alpha_text [B,max_steps,L] and alpha_vis [B,max_steps,G*G], and att_sum [B,G*G] = the row's visual
 * attention summed over its steps (the auxiliary head's input, predict.py:118-120).  No host synchronisation. */
int gscan_greedy_decode(const gscan_dims *dims, int max_steps, const gscan_params *params, const gscan_batch *batch,
                        void *workspace, int sos_idx, int eos_idx, int64_t *tokens, int32_t *steps, float *alpha_text,
                        float *alpha_vis, float *att_sum, void *stream);

/* Model.get_loss (seq2seq/model.py:147-160): NLL of targets shifted left by one with a PAD
 * appended, over positions whose shifted target != pad.  Writes loss_sum[0] = sum of -logp
 * and count[0] = number of such positions (as float); the reference's loss is
 * loss_sum/count.  dlogp (may be NULL) receives -1 at the picked entries and 0 elsewhere,
 * i.e. d(loss_sum)/d(logp). */
int gscan_sequence_nll(const float *logp, const int64_t *targets, int B, int T, int V, int pad,
                       float *loss_sum, float *count, float *dlogp, void *stream);

/* Model.get_auxiliary_loss (seq2seq/model.py:162-164): loss_sum[0] = sum_b -aux_logp[b,pos[b]];
 * daux (may be NULL) = d(loss_sum)/d(aux_logp). */
int gscan_position_nll(const float *aux_logp, const int64_t *positions, int B, int M,
                       float *loss_sum, float *daux, void *stream);

/* Model.get_metrics (seq2seq/model.py:117-137): out[0] = correct tokens, out[1] = non-pad
 * tokens, out[2] = rows with every non-pad token correct. */
int gscan_sequence_metrics(const float *logp, const int64_t *targets, int B, int T, int V, int pad,
                           float *out3, void *stream);

/* torch.optim.Adam step + LambdaLR factor of seq2seq/train.py:67-70,111-112 over one flat
 * buffer of n floats: lr_t = lr * lr_decay^((step-1)/lr_decay_steps), step is 1-based.
 * The gradient is multiplied by grad_scale[0] (device pointer, may be NULL = 1) before
 * use — the data-parallel step passes 1/global_token_count there without a host sync. */
int gscan_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                    float lr, float beta1, float beta2, float eps, float lr_decay, float lr_decay_steps,
                    int64_t step, const float *grad_scale, void *stream);

/* optimizer.step() followed by optimizer.zero_grad() (train.py:110-113) in one launch: as gscan_adam_step, and
 * the gradient buffer is cleared as it is consumed. */
int gscan_adam_step_zero_grad(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                              float beta1, float beta2, float eps, float lr_decay, float lr_decay_steps,
                              int64_t step, const float *grad_scale, void *stream);

/* gscan_adam_step_zero_grad on gradients of a SUM loss: every gradient is divided by count[0] (device scalar: the
 * all-reduced number of live target tokens) before the update. */
int gscan_adam_step_mean(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                         float beta1, float beta2, float eps, float lr_decay, float lr_decay_steps, int64_t step,
                         const float *count, void *stream);

/* optimizer.step() + optimizer.zero_grad() (train.py:111-113) AND the three dropout masks of the NEXT training step
 * (gscan_dropout_masks: contiguous cnn | enc | dec, Philox stream `stream_id`) in ONE launch: the masks depend on
 * nothing but a counter, so a training loop that knows the next batch's shape saves a launch at the head of every step.
 * count != NULL: gradients of a sum loss, divided by count[0] (gscan_adam_step_mean); NULL: gscan_adam_step_zero_grad.
 * The mask buffer may be the one the step that just finished used (its backward pass is complete on this stream). */
int gscan_adam_step_masks(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                          float beta2, float eps, float lr_decay, float lr_decay_steps, int64_t step, const float *count,
                          float *mask_out, size_t n_cnn, size_t n_enc, size_t n_dec, float p_cnn, float p_enc, float p_dec,
                          uint64_t seed, uint64_t stream_id, void *stream);

/* ---- the loop body's remaining pieces as launches of their own ---- */

/* Both losses in one launch: stats[4] = [sum NLL, tokens, sum aux NLL, rows] and the unit seeds
 * dlogp = d(sum NLL)/d(logp), daux = d(sum aux NLL)/d(aux_logp).  aux_logp/positions/daux may be NULL. */
int gscan_step_losses(const float *logp, const int64_t *targets, const float *aux_logp, const int64_t *positions, int B,
                      int T, int V, int M, int pad, float *stats, float *dlogp, float *daux, void *stream);

/* seeds[0] = 1/stats[1], seeds[1] = w/stats[3] (0 without the auxiliary task), seeds[2] = the reference's loss
 * (train.py:102-107).  Call after the data-parallel all-reduce of stats. */
int gscan_loss_seeds(const float *stats, float weight_target_loss, int auxiliary, float *seeds, void *stream);

/* gscan_backward with dlogp multiplied by seeds[0] and daux_logp by seeds[1] (device scalars, may be NULL). */
int gscan_backward_seeded(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                          const gscan_masks *masks, void *workspace, const float *dlogp, const float *daux_logp,
                          const float *seeds, const gscan_params *grads, void *stream);

/* loss.backward() of the reference's training loss itself (train.py:102-110), with nothing launched between
 * gscan_forward and this call:  loss = get_loss(logp, targets) [+ weight_target_loss * get_auxiliary_loss(aux_logp,
 * target_positions) when dims->auxiliary]  (model.py:147-164).  gscan_forward leaves per-row partial sums of both
 * losses in the workspace; the backward kernels start from them.  stats[4] = [sum NLL, live tokens, sum aux NLL,
 * rows] and seeds[3] = [seed of the sequence loss, seed of the auxiliary loss, loss] are written for the caller
 * (device, must not be NULL).
 * sum_reduction = 0: the reference's loss as is (mean over live tokens + w * mean over rows; seeds 1/tokens, w/rows).
 * sum_reduction = 1: gradients of  sum NLL + w * sum aux NLL  (seeds 1, w) — the data-parallel form: every rank adds
 * its sums, ONE all-reduce carries gradients and `stats` together, and the optimiser divides by the global token
 * count (gscan_adam_step_mean).  With the auxiliary loss the two terms need different divisors; a data-parallel
 * step then all-reduces `stats` first (gscan_step_losses, gscan_loss_seeds, gscan_backward_seeded). */
int gscan_backward_nll(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                       const gscan_masks *masks, void *workspace, float weight_target_loss, int sum_reduction,
                       float *stats, float *seeds, const gscan_params *grads, void *stream);

/* One training iteration's forward pass, loss and backward pass (seq2seq/train.py:96-110: model(...), get_loss,
 * loss.backward()) as ONE call: what gscan_forward followed by gscan_backward_nll compute.  Arguments as in those two
 * calls; `logp` / `aux_logp` receive the forward pass's outputs. */
int gscan_train_step_nll(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                         const gscan_masks *masks, void *workspace, float *logp, float *aux_logp, float weight_target_loss,
                         int sum_reduction, float *stats, float *seeds, const gscan_params *grads, void *stream);

/* The three dropout masks of one step (contiguous: cnn | enc | dec) in one launch; the Philox stream id is
 * dev_stream_id[0] when that device pointer is not NULL. */
int gscan_dropout_masks(float *out, size_t n_cnn, size_t n_enc, size_t n_dec, float p_cnn, float p_enc, float p_dec,
                        uint64_t seed, uint64_t stream_id, const uint64_t *dev_stream_id, void *stream);

/* Counter-based (Philox-4x32-10) scaled dropout mask: out[i] = keep ? 1/(1-p) : 0. */
int gscan_dropout_mask(float *out, size_t n, float p, uint64_t seed, uint64_t stream_id, void *stream);

/* The masks that a step with gscan_masks::in_kernel != 0 and the same (seed, stream_id, p) draws inside its kernels,
 * written to memory: cnn [B,G*G,3*Co], enc [B,L,E], dec [B,T,H] (a NULL pointer skips that mask; p = 0 gives ones).
 * Feeding them back through the pointer form of gscan_masks reproduces that step bit for bit: tests, and the way to
 * hand the masks of a production step to an external checker (the oracle). */
int gscan_dropout_masks_kernel_layout(const gscan_dims *dims, float *cnn, float *enc, float *dec, float p_cnn, float p_enc,
                                      float p_dec, uint64_t seed, uint64_t stream_id, void *stream);

/* ---- data-parallel gradient exchange (NOT in the reference, which is single-process: seq2seq/train.py:24,65;
 * SURVEY.md 8(b) `flat_allreduce`, 8(e)).  One RCCL communicator per process (one process per GPU); the all-reduce
 * is enqueued on the CALLER'S stream, between gscan_backward_nll and gscan_adam_step_mean, so the step needs no
 * hop to a communication stream.  RCCL is resolved at run time from the librccl.so.1 the process already holds
 * (PyTorch-ROCm's); these four calls fail with a message when there is none, nothing else in the library needs it.
 *   every rank: gscan_comm_available() == 0 ?  -> agree on the answer over the side channel FIRST: a rank that cannot
 *                                            load RCCL must not leave the others blocked inside gscan_comm_init
 *   rank 0:     gscan_comm_unique_id(id)  -> ship the GSCAN_COMM_ID_BYTES bytes to every rank (any side channel:
 *                                            the torch.distributed store, MPI, a file)
 *   every rank: hipSetDevice(local GPU); gscan_comm_init(&comm, nranks, rank, id)   (collective: blocks until all join)
 *   every step: gscan_allreduce_f32(comm, flat_gradients, n, stream)               (in place, sum)
 *   at exit:    gscan_comm_destroy(comm)                                                                            */
#define GSCAN_COMM_ID_BYTES 128
/* 0 when RCCL can be loaded in this process (dlopen + symbols only: no device call, nothing collective) */
int gscan_comm_available(void);
int gscan_comm_unique_id(void *id_host);
int gscan_comm_init(void **comm, int nranks, int rank, const void *id_host);
int gscan_allreduce_f32(void *comm, float *buf, size_t n, void *stream);
int gscan_comm_destroy(void *comm);
/* ranks of the communicator as RCCL reports them (ncclCommCount): what bench.py prints as rccl_nranks */
int gscan_comm_count(void *comm, int *nranks);
/* Two-bucket gradient exchange (ABI 14).  The backward pass finishes its gradients in two groups: EARLY — everything its
 * first leaf stream produces: the decoder's, both attentions', the bridge's and the conditional query's parameters —
 * about 45 us before LATE: the convolution kernels (second leaf stream) and the command encoder (embedding + LSTM, the
 * caller's chain).  gscan_early_gradients_wait makes `stream` wait for the EARLY group of the most recent
 * gscan_backward* / gscan_train_step_nll call of this process (an event recorded on the leaf stream; no host
 * synchronisation), so that a communication stream can all-reduce that part of the flat gradient while the tail of the
 * backward pass still runs.  Returns non-zero when no backward call has been issued yet. */
int gscan_early_gradients_wait(void *stream);
/* The same overlap without a stream or an event of the caller's: register a communicator and a range of the flat
 * gradient buffer (the EARLY group: from the bridge's weight to the end of the buffer, statistics included), and every
 * following backward pass all-reduces that range ITSELF, on its first leaf stream, right behind the last kernel that
 * writes into it; the join in front of the optimiser that the pass ends with anyway covers it.  The caller then
 * all-reduces only the rest on its own stream.  comm = NULL clears the registration (do so before gscan_comm_destroy).
 * Use a communicator of its own for this (not the one the late group travels on). */
int gscan_comm_set_early_allreduce(void *comm, float *buf, size_t n);

/* Per-kernel-family timing for roofline reports: when enabled, every launch of a family
 * ("decoder_forward", "decoder_backward", "encoder_forward", "encoder_backward", "gemm", "conv_forward",
 * "conv_backward", "keys_backward") is bracketed by HIP events on its launch stream.  gscan_probe_read synchronises
 * on those events and returns the summed duration, the summed EXECUTED flops (2 M N K of every product launched,
 * tile padding excluded; 0 for the input-sparse convolution kernels, whose work depends on the data), the summed
 * ALGORITHMIC flops (the launch's share of SURVEY.md 8(d)'s count: valid convolution taps only, composite-weight
 * and U-image products not counted; DESIGN.md states the per-launch formulas) and the launch count since the last
 * reset.  Do not enable during graph capture. */
int gscan_probe_enable(int on);
/* In-kernel timeline (diagnostic): with a device buffer of GSCAN_TRACE_WORDS uint64 set (zeroed by the caller), the
 * first workgroup of every kernel launch appends (kernel id, grid size, 100 MHz device clock) to the start list and
 * the last workgroup appends the same to the end list on exit: words [0],[1] = list lengths, then two lists of
 * GSCAN_TRACE_RECORDS records of 3 words.  No profiler is involved, so the timeline of an undisturbed step can be
 * read (tools/device_timeline.py).  NULL switches tracing off.  Synchronises the device.  The stamps are compiled in
 * only with -DGSCAN_TRACE (they cost 1.4 % of a step when merely present); the shipped build returns an error for a
 * non-NULL buffer. */
#define GSCAN_TRACE_RECORDS 256
#define GSCAN_TRACE_WORDS (2 + 2 * 3 * GSCAN_TRACE_RECORDS)
int gscan_trace_set(unsigned long long *device_buffer);
int gscan_probe_reset(void);
int gscan_probe_read(const char *name, double *total_ms, double *executed_flops, double *algorithmic_flops,
                     int64_t *launches);

/* ---- building blocks, exported so that each kernel can be parity-tested on its own ---- */

/* C[M,N] = act(alpha * A.B + beta * C + bias[n]) * mask[m,n]
 * A(m,k) at a[m*sam + k*sak], B(k,n) at b[k*sbk + n*sbn], C and mask row-major with ldc.
 * act: 0 none, 1 relu, 2 tanh.  bias/mask may be NULL.  split_k > 1 accumulates partial
 * products with float atomics and requires beta == 1, act == 0, bias == mask == NULL. */
int gscan_gemm_f32(int M, int N, int K, float alpha, const float *a, int64_t sam, int64_t sak,
                   const float *b, int64_t sbk, int64_t sbn, float beta, float *c, int64_t ldc,
                   const float *bias, int act, const float *mask, int split_k, void *stream);
/* The same product as the training step issues it: asum (optional, [M]) += sum_k A(m,k) (the bias gradient that
 * rides along a weight gradient), and `scratch` (optional, scratch_floats floats of device memory that no other
 * launch in flight uses) for the partial tiles of a split-K product, which are then added in a FIXED order by a
 * second launch: bitwise reproducible, where gscan_gemm_f32's float atomics are not (csrc/gemm_mt.hip). */
int gscan_gemm_f32_scratch(int M, int N, int K, float alpha, const float *a, int64_t sam, int64_t sak,
                           const float *b, int64_t sbk, int64_t sbn, float beta, float *c, int64_t ldc,
                           const float *bias, int act, const float *mask, int split_k, float *asum,
                           float *scratch, size_t scratch_floats, void *stream);

/* ConvolutionalNet.forward (seq2seq/cnn_model.py:22-36) on its own: the three same-padded convolutions (kernels
 * 1, 5, K3; kh walks grid columns and kw grid rows because the reference convolves the transposed image), bias,
 * ReLU and dropout mask, computed from the NON-ZEROS of the world tensor (csrc/conv.hip; exact for any input).
 * world: [B,G,G,C] float32, or uint8 when world_is_u8 != 0.  conv_w[i] [Co,C,k_i,k_i], conv_b[i] [Co];
 * mask [B,G*G,3Co] or NULL; feat [B,G*G,3Co].  image_scratch: (1 + 25 + K3*K3) * C * roundup(Co, 32) floats. */
int gscan_world_encoder_forward(const void *world, int world_is_u8, const float *const conv_w[3],
                                const float *const conv_b[3], int B, int G, int C, int Co, int K3, const float *mask,
                                float *image_scratch, float *feat, void *stream);
/* Its weight gradients: given d(loss)/d(conv output before ReLU/dropout) `dfeat` [B,G*G,3Co], ADDS the kernel and
 * bias gradients into grad_w[i] / grad_b[i].  list_scratch: gscan_world_encoder_backward_scratch_floats(B, G, C)
 * floats (the per-channel lists of non-zeros). */
size_t gscan_world_encoder_backward_scratch_floats(int B, int G, int C);
int gscan_world_encoder_backward(const void *world, int world_is_u8, const float *dfeat, int B, int G, int C, int Co,
                                 int K3, float *list_scratch, float *const grad_w[3], float *const grad_b[3],
                                 void *stream);

/* Masked per-row LSTM over the command (seq2seq/seq2seq_model.py:62-88).
 * gx [B,L,D,4He] = W_ih x + b_ih (D directions); out [B,L,He] = sum of directions,
 * zero at t >= len; h_final [B,He].  gates/cells/hprev [B,L,D,*] are saved for backward.
 * w_image_scratch: D*4He*He floats of device scratch (the kernel reads W_hh through a register image). */
int gscan_encoder_lstm_forward(int B, int L, int He, int D, const float *gx, const int32_t *lengths,
                               const float *w_hh_fwd, const float *b_hh_fwd, const float *w_hh_rev,
                               const float *b_hh_rev, float *out, float *h_final, float *gates,
                               float *cells, float *hprev, float *w_image_scratch, void *stream);
int gscan_encoder_lstm_backward(int B, int L, int He, int D, const int32_t *lengths, const float *w_hh_fwd,
                                const float *w_hh_rev, const float *gates, const float *cells,
                                const float *d_out, const float *d_h_final, float *delta, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSCAN_HIP_H */
