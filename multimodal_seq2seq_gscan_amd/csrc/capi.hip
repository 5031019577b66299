// extern "C" surface of libgscan_hip.so (declared in include/gscan_hip.h).
#include <string.h>

#include "step.h"

namespace gscan {
static thread_local char g_error[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace gscan

using namespace gscan;

#define ARG(cond, ...) do { if (!(cond)) { set_error(__VA_ARGS__); return 1; } } while (0)

extern "C" {

int gscan_abi_version(void) { return GSCAN_ABI_VERSION; }
const char *gscan_last_error(void) { return g_error; }

size_t gscan_workspace_bytes(const gscan_dims *dims) {
    if (!dims) { set_error("workspace_bytes: dims is NULL"); return 0; }
    if (check_dims(*dims)) return 0;
    Workspace ws;
    workspace_layout(*dims, &ws);
    return (size_t)ws.total_floats * sizeof(float);
}

int gscan_decoder_kernel_family(const gscan_dims *dims) {
    if (!dims) { set_error("decoder_kernel_family: dims is NULL"); return -1; }
    if (check_dims(*dims)) return -1;
    return decoder_fast_supported(dims->H, dims->L, dims->G * dims->G, dims->V, dims->conditional != 0) ? 1 : 0;
}

int gscan_workspace_find(const gscan_dims *dims, const char *name, size_t *offset_bytes, size_t *count) {
    ARG(dims && name && offset_bytes && count, "workspace_find: NULL argument");
    if (int rc = check_dims(*dims)) return rc;
    Workspace ws;
    workspace_layout(*dims, &ws);
    for (int i = 0; i < ws.nslots; ++i)
        if (strcmp(ws.slot[i].name, name) == 0) {
            *offset_bytes = (size_t)ws.slot[i].offset * sizeof(float);
            *count = (size_t)ws.slot[i].count;
            return 0;
        }
    set_error("workspace_find: no slot named '%s'", name);
    return 1;
}

int gscan_forward(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                  const gscan_masks *masks, void *workspace, float *logp, float *aux_logp, void *stream) {
    ARG(dims && params && batch && workspace, "forward: NULL argument");
    ARG(batch->commands && batch->cmd_lengths && (batch->world || batch->world_u8) && batch->targets, "forward: NULL batch array");
    gscan_masks none{nullptr, nullptr, nullptr};
    return step_forward(*dims, *params, *batch, masks ? *masks : none, (float *)workspace, logp, aux_logp,
                        (hipStream_t)stream);
}

int gscan_backward(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                   const gscan_masks *masks, void *workspace, const float *dlogp, const float *daux_logp,
                   const gscan_params *grads, void *stream) {
    ARG(dims && params && batch && workspace && grads, "backward: NULL argument");
    gscan_masks none{nullptr, nullptr, nullptr};
    ARG(dlogp, "backward: dlogp is NULL");
    return step_backward(*dims, *params, *batch, masks ? *masks : none, (float *)workspace, dlogp, daux_logp, nullptr,
                         nullptr, *grads, (hipStream_t)stream);
}

int gscan_backward_seeded(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                          const gscan_masks *masks, void *workspace, const float *dlogp, const float *daux_logp,
                          const float *seeds, const gscan_params *grads, void *stream) {
    ARG(dims && params && batch && workspace && grads, "backward: NULL argument");
    gscan_masks none{nullptr, nullptr, nullptr};
    ARG(dlogp, "backward: dlogp is NULL");
    return step_backward(*dims, *params, *batch, masks ? *masks : none, (float *)workspace, dlogp, daux_logp, seeds,
                         nullptr, *grads, (hipStream_t)stream);
}

int gscan_backward_nll(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                       const gscan_masks *masks, void *workspace, float weight_target_loss, int sum_reduction,
                       float *stats, float *seeds, const gscan_params *grads, void *stream) {
    ARG(dims && params && batch && workspace && grads && stats && seeds, "backward_nll: NULL argument");
    ARG(!dims->auxiliary || batch->target_positions, "backward_nll: auxiliary task set but target_positions is NULL");
    gscan_masks none{nullptr, nullptr, nullptr};
    const NllSeed nll{weight_target_loss, sum_reduction != 0, stats, seeds};
    return step_backward(*dims, *params, *batch, masks ? *masks : none, (float *)workspace, nullptr, nullptr, nullptr,
                         &nll, *grads, (hipStream_t)stream);
}

int gscan_train_step_nll(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                         const gscan_masks *masks, void *workspace, float *logp, float *aux_logp, float weight_target_loss,
                         int sum_reduction, float *stats, float *seeds, const gscan_params *grads, void *stream) {
    ARG(dims && params && batch && workspace && grads && stats && seeds && logp, "train_step_nll: NULL argument");
    ARG(batch->commands && batch->cmd_lengths && (batch->world || batch->world_u8) && batch->targets,
        "train_step_nll: NULL batch array");
    ARG(!dims->auxiliary || batch->target_positions, "train_step_nll: auxiliary task set but target_positions is NULL");
    gscan_masks none{nullptr, nullptr, nullptr};
    const NllSeed nll{weight_target_loss, sum_reduction != 0, stats, seeds};
    return step_train_nll(*dims, *params, *batch, masks ? *masks : none, (float *)workspace, logp, aux_logp, nll, *grads,
                          (hipStream_t)stream);
}

int gscan_encode(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                 const gscan_masks *masks, void *workspace, void *stream) {
    ARG(dims && params && batch && workspace, "encode: NULL argument");
    gscan_masks none{nullptr, nullptr, nullptr};
    return step_encode(*dims, *params, *batch, masks ? *masks : none, (float *)workspace, (hipStream_t)stream);
}

int gscan_decode_step(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                      const int64_t *tokens, const float *h_in, const float *c_in, void *workspace, float *logits,
                      float *h_out, float *c_out, float *alpha_text, float *alpha_vis, void *stream) {
    ARG(dims && params && batch && batch->cmd_lengths && tokens && h_in && c_in && workspace && logits && h_out &&
            c_out && alpha_text && alpha_vis,
        "decode_step: NULL argument");
    return step_decode_one(*dims, *params, *batch, tokens, h_in, c_in, (float *)workspace, logits, h_out, c_out,
                           alpha_text, alpha_vis, (hipStream_t)stream);
}

int gscan_decode_batched(const gscan_dims *dims, const gscan_params *params, const gscan_batch *batch,
                         const float *encoded_situations, const float *encoder_outputs, const float *hidden_states,
                         void *workspace, float *logp, float *att_sum, void *stream) {
    ARG(dims && params && batch && workspace && logp && att_sum, "decode_batched: NULL argument");
    ARG(batch->cmd_lengths && batch->targets && encoded_situations && encoder_outputs && hidden_states,
        "decode_batched: NULL array");
    gscan_dims d = *dims;
    d.auxiliary = 0;                        // the summed visual attention itself is returned, not its log_softmax
    gscan_masks none{};
    if (int rc = step_forward(d, *params, *batch, none, (float *)workspace, logp, nullptr, (hipStream_t)stream,
                              encoded_situations, encoder_outputs, hidden_states)) return rc;
    size_t off = 0, cnt = 0;
    if (int rc = gscan_workspace_find(&d, "att_sum", &off, &cnt)) return rc;
    GSCAN_HIP(hipMemcpyAsync(att_sum, (const char *)workspace + off, cnt * sizeof(float), hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
    return 0;
}

int gscan_greedy_decode(const gscan_dims *dims, int max_steps, const gscan_params *params, const gscan_batch *batch,
                        void *workspace, int sos_idx, int eos_idx, int64_t *tokens, int32_t *steps, float *alpha_text,
                        float *alpha_vis, float *att_sum, void *stream) {
    ARG(dims && params && batch && workspace, "greedy_decode: NULL argument");
    ARG(batch->commands && batch->cmd_lengths && (batch->world || batch->world_u8), "greedy_decode: NULL batch array");
    return step_greedy(*dims, max_steps, *params, *batch, (float *)workspace, sos_idx, eos_idx, tokens, steps,
                       alpha_text, alpha_vis, att_sum, (hipStream_t)stream);
}

int gscan_step_losses(const float *logp, const int64_t *targets, const float *aux_logp, const int64_t *positions, int B,
                      int T, int V, int M, int pad, float *stats, float *dlogp, float *daux, void *stream) {
    ARG(logp && targets && stats && dlogp && B > 0 && T > 0 && V > 0, "step_losses: bad argument");
    ARG(!aux_logp || (positions && daux && M > 0), "step_losses: auxiliary arrays incomplete");
    return step_losses(logp, targets, aux_logp, positions, B, T, V, M, pad, stats, dlogp, daux, (hipStream_t)stream);
}

int gscan_loss_seeds(const float *stats, float weight_target_loss, int auxiliary, float *seeds, void *stream) {
    ARG(stats && seeds, "loss_seeds: NULL argument");
    return loss_seeds(stats, weight_target_loss, auxiliary, seeds, (hipStream_t)stream);
}

int gscan_dropout_masks(float *out, size_t n_cnn, size_t n_enc, size_t n_dec, float p_cnn, float p_enc, float p_dec,
                        uint64_t seed, uint64_t stream_id, const uint64_t *dev_stream_id, void *stream) {
    ARG(out, "dropout_masks: NULL output");
    const size_t n[3] = {n_cnn, n_enc, n_dec};
    const float p[3] = {p_cnn, p_enc, p_dec};
    return dropout_masks(out, n, p, seed, stream_id, dev_stream_id, (hipStream_t)stream);
}

int gscan_trace_set(unsigned long long *device_buffer) {
    static_assert(GSCAN_TRACE_RECORDS == kTraceRecords, "header and kernels disagree on the trace buffer layout");
#ifndef GSCAN_TRACE
    GSCAN_CHECK(device_buffer == nullptr, "trace_set: this build has no in-kernel stamps (compile every source with "
                                          "-DGSCAN_TRACE: tools/device_timeline.py does)");
#endif
    GSCAN_HIP(hipDeviceSynchronize());
    int rc = trace_set_gemm(device_buffer) | trace_set_gemm_mt(device_buffer) | trace_set_gemm_ws(device_buffer) | trace_set_elementwise(device_buffer) | trace_set_loss(device_buffer) |
             trace_set_lstm_encoder(device_buffer) | trace_set_decoder(device_buffer) | trace_set_decoder_p1(device_buffer) | trace_set_decoder_p2(device_buffer) |
             trace_set_decoder_p3(device_buffer) | trace_set_decoder_any(device_buffer) |
             trace_set_attention_grad(device_buffer) | trace_set_conv(device_buffer);
    GSCAN_CHECK(rc == 0, "trace_set: hipMemcpyToSymbol failed");
    GSCAN_HIP(hipDeviceSynchronize());
    return 0;
}

int gscan_sequence_nll(const float *logp, const int64_t *targets, int B, int T, int V, int pad, float *loss_sum,
                       float *count, float *dlogp, void *stream) {
    ARG(logp && targets && loss_sum && count && B > 0 && T > 0 && V > 0, "sequence_nll: bad argument");
    return sequence_nll(logp, targets, B, T, V, pad, loss_sum, count, dlogp, (hipStream_t)stream);
}

int gscan_position_nll(const float *aux_logp, const int64_t *positions, int B, int M, float *loss_sum, float *daux,
                       void *stream) {
    ARG(aux_logp && positions && loss_sum && B > 0 && M > 0, "position_nll: bad argument");
    return position_nll(aux_logp, positions, B, M, loss_sum, daux, (hipStream_t)stream);
}

int gscan_sequence_metrics(const float *logp, const int64_t *targets, int B, int T, int V, int pad, float *out3,
                           void *stream) {
    ARG(logp && targets && out3 && B > 0 && T > 0 && V > 0, "sequence_metrics: bad argument");
    return sequence_metrics(logp, targets, B, T, V, pad, out3, (hipStream_t)stream);
}

int gscan_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                    float beta1, float beta2, float eps, float lr_decay, float lr_decay_steps, int64_t step,
                    const float *grad_scale, void *stream) {
    ARG(param && grad && exp_avg && exp_avg_sq && n > 0, "adam_step: bad argument");
    return adam_step(param, const_cast<float *>(grad), exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, lr_decay,
                     lr_decay_steps, step, grad_scale, nullptr, 0, (hipStream_t)stream);
}

int gscan_adam_step_zero_grad(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                              float beta1, float beta2, float eps, float lr_decay, float lr_decay_steps,
                              int64_t step, const float *grad_scale, void *stream) {
    ARG(param && grad && exp_avg && exp_avg_sq && n > 0, "adam_step_zero_grad: bad argument");
    return adam_step(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, lr_decay, lr_decay_steps, step,
                     grad_scale, nullptr, 1, (hipStream_t)stream);
}

int gscan_adam_step_mean(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                         float beta1, float beta2, float eps, float lr_decay, float lr_decay_steps, int64_t step,
                         const float *count, void *stream) {
    ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && count, "adam_step_mean: bad argument");
    return adam_step(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, lr_decay, lr_decay_steps, step,
                     count, nullptr, 1 | 2, (hipStream_t)stream);
}

int gscan_adam_step_masks(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                          float beta2, float eps, float lr_decay, float lr_decay_steps, int64_t step, const float *count,
                          float *mask_out, size_t n_cnn, size_t n_enc, size_t n_dec, float p_cnn, float p_enc, float p_dec,
                          uint64_t seed, uint64_t stream_id, void *stream) {
    ARG(param && grad && exp_avg && exp_avg_sq && n > 0, "adam_step_masks: bad argument");
    const size_t nm[3] = {n_cnn, n_enc, n_dec};
    const float pm[3] = {p_cnn, p_enc, p_dec};
    return adam_step_masks(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, lr_decay, lr_decay_steps, step, count,
                           count ? (1 | 2) : 1, mask_out, nm, pm, seed, stream_id, (hipStream_t)stream);
}

int gscan_dropout_mask(float *out, size_t n, float p, uint64_t seed, uint64_t stream_id, void *stream) {
    ARG(out || n == 0, "dropout_mask: NULL output");
    return dropout_mask(out, n, p, seed, stream_id, (hipStream_t)stream);
}

int gscan_dropout_masks_kernel_layout(const gscan_dims *dims, float *cnn, float *enc, float *dec, float p_cnn, float p_enc,
                                      float p_dec, uint64_t seed, uint64_t stream_id, void *stream) {
    ARG(dims, "dropout_masks_kernel_layout: NULL dims");
    return dropout_masks_kernel_layout(cnn, enc, dec, dims->B, dims->G * dims->G, dims->Co, dims->L, dims->E, dims->T, dims->H,
                                       p_cnn, p_enc, p_dec, seed, stream_id, (hipStream_t)stream);
}

int gscan_comm_available(void) { return comm_available(); }
int gscan_comm_unique_id(void *id_host) { return comm_unique_id(id_host); }
int gscan_comm_init(void **comm, int nranks, int rank, const void *id_host) { return comm_init(comm, nranks, rank, id_host); }
int gscan_allreduce_f32(void *comm, float *buf, size_t n, void *stream) {
    return comm_allreduce_f32(comm, buf, n, (hipStream_t)stream);
}
int gscan_comm_destroy(void *comm) { return comm_destroy(comm); }
int gscan_comm_count(void *comm, int *nranks) { return comm_count(comm, nranks); }
int gscan_early_gradients_wait(void *stream) { return early_gradients_wait((hipStream_t)stream); }
int gscan_comm_set_early_allreduce(void *comm, float *buf, size_t n) { return set_early_allreduce(comm, buf, n); }

int gscan_probe_enable(int on) { return probe_enable(on); }
int gscan_probe_reset(void) { return probe_reset(); }
int gscan_probe_read(const char *name, double *total_ms, double *executed_flops, double *algorithmic_flops,
                     int64_t *launches) {
    ARG(name && total_ms && executed_flops && algorithmic_flops && launches, "probe_read: NULL argument");
    return probe_read(name, total_ms, executed_flops, algorithmic_flops, launches);
}

int gscan_gemm_f32(int M, int N, int K, float alpha, const float *a, int64_t sam, int64_t sak, const float *b,
                   int64_t sbk, int64_t sbn, float beta, float *c, int64_t ldc, const float *bias, int act,
                   const float *mask, int split_k, void *stream) {
    return gemm_f32(M, N, K, alpha, a, sam, sak, b, sbk, sbn, beta, c, ldc, bias, act, mask, split_k,
                    (hipStream_t)stream);
}

int gscan_gemm_f32_scratch(int M, int N, int K, float alpha, const float *a, int64_t sam, int64_t sak, const float *b,
                           int64_t sbk, int64_t sbn, float beta, float *c, int64_t ldc, const float *bias, int act,
                           const float *mask, int split_k, float *asum, float *scratch, size_t scratch_floats,
                           void *stream) {
    return gemm_f32_ex(M, N, K, alpha, a, sam, sak, b, sbk, sbn, beta, c, ldc, bias, act, mask, split_k, asum, scratch,
                       scratch_floats, (hipStream_t)stream);
}

int gscan_world_encoder_forward(const void *world, int world_is_u8, const float *const conv_w[3],
                                const float *const conv_b[3], int B, int G, int C, int Co, int K3, const float *mask,
                                float *image_scratch, float *feat, void *stream) {
    ARG(world && conv_w && conv_b && image_scratch && feat, "world_encoder_forward: NULL argument");
    ARG(B > 0 && G > 0 && C > 0 && Co > 0 && K3 > 0 && (K3 & 1), "world_encoder_forward: bad dims");
    const float *const cw[3] = {conv_w[0], conv_w[1], conv_w[2]};
    const float *const cb[3] = {conv_b[0], conv_b[1], conv_b[2]};
    if (int rc = conv_weight_image(cw, C, Co, K3, image_scratch, (hipStream_t)stream)) return rc;
    return world_conv_forward(world, world_is_u8, image_scratch, cb, mask, B, G, C, Co, K3, feat, (hipStream_t)stream);
}

size_t gscan_world_encoder_backward_scratch_floats(int B, int G, int C) {
    return (B > 0 && G > 0 && C > 0) ? world_conv_backward_scratch_floats(B, G, C) : 0;
}

int gscan_world_encoder_backward(const void *world, int world_is_u8, const float *dfeat, int B, int G, int C, int Co,
                                 int K3, float *list_scratch, float *const grad_w[3], float *const grad_b[3], void *stream) {
    ARG(world && dfeat && list_scratch && grad_w && grad_b, "world_encoder_backward: NULL argument");
    ARG(B > 0 && G > 0 && C > 0 && Co > 0 && K3 > 0 && (K3 & 1), "world_encoder_backward: bad dims");
    float *const gw[3] = {grad_w[0], grad_w[1], grad_w[2]};
    float *const gb[3] = {grad_b[0], grad_b[1], grad_b[2]};
    if (int rc = world_conv_lists(world, world_is_u8, B, G, C, list_scratch, (hipStream_t)stream)) return rc;
    return world_conv_backward(dfeat, B, G, C, Co, K3, list_scratch, gw, gb, (hipStream_t)stream);
}

int gscan_encoder_lstm_forward(int B, int L, int He, int D, const float *gx, const int32_t *lengths,
                               const float *w_hh_fwd, const float *b_hh_fwd, const float *w_hh_rev,
                               const float *b_hh_rev, float *out, float *h_final, float *gates, float *cells,
                               float *hprev, float *w_image_scratch, void *stream) {
    ARG(gx && lengths && w_hh_fwd && b_hh_fwd && out && h_final && gates && cells && hprev && w_image_scratch,
        "encoder_lstm_forward: NULL argument");
    ARG(B > 0 && L > 0 && He > 0 && (D == 1 || D == 2), "encoder_lstm_forward: bad dims");
    // the kernel accumulates the direction sums into out / h_final and reads its weights from a register image
    GSCAN_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * L * He, (hipStream_t)stream));
    GSCAN_HIP(hipMemsetAsync(h_final, 0, sizeof(float) * (size_t)B * He, (hipStream_t)stream));
    if (encoder_fast_supported(He, L, 0))
        if (int rc = encoder_weight_image(w_hh_fwd, w_hh_rev, He, D, w_image_scratch, (hipStream_t)stream)) return rc;
    return encoder_lstm_forward(B, L, He, D, gx, lengths, w_hh_fwd, b_hh_fwd, w_hh_rev, b_hh_rev, out, h_final, gates,
                                cells, hprev, w_image_scratch, (hipStream_t)stream);
}

int gscan_encoder_lstm_backward(int B, int L, int He, int D, const int32_t *lengths, const float *w_hh_fwd,
                                const float *w_hh_rev, const float *gates, const float *cells, const float *d_out,
                                const float *d_h_final, float *delta, void *stream) {
    ARG(lengths && w_hh_fwd && gates && cells && d_out && d_h_final && delta, "encoder_lstm_backward: NULL argument");
    return encoder_lstm_backward(B, L, He, D, lengths, w_hh_fwd, w_hh_rev, gates, cells, d_out, d_h_final, delta,
                                 (hipStream_t)stream);
}

}  // extern "C"
