// The masked NLL reductions of Model.get_loss / get_auxiliary_loss / get_metrics (seq2seq/model.py:117-164) as
// stand-alone kernels (autograd and data-parallel paths; the single-process training step computes its loss inside
// the decoder kernels).  log_softmax itself lives in the decoder kernels' epilogue / prologue.
#include "step.h"

namespace gscan {


// ------------------------------------------------------------------------------------------
// Model.get_loss: target of position (b,t) is targets[b,t+1] (PAD for t = T-1); positions whose
// target == pad are ignored (nn.NLLLoss(ignore_index=pad), model.py:100,108-115,147-160).
// Single workgroup: B*T is a few thousand; the reduction order is fixed (deterministic).
// ------------------------------------------------------------------------------------------
__global__ void sequence_nll_kernel(const float *__restrict__ logp, const int64_t *__restrict__ targets, int B, int T,
                                    int V, int pad, float *loss_sum, float *count, float *__restrict__ dlogp) {
    __shared__ float s_sum[16], s_cnt[16];
    float acc = 0.f, cnt = 0.f;
    const int n = B * T;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int t = i % T;
        const int64_t tgt = (t + 1 < T) ? targets[i + 1] : (int64_t)0;   // model.py:112 appends literal 0
        const bool live = tgt != pad && tgt >= 0 && tgt < V;
        if (dlogp)
            for (int j = 0; j < V; ++j) dlogp[(int64_t)i * V + j] = (live && j == tgt) ? -1.f : 0.f;
        if (live) { acc -= logp[(int64_t)i * V + tgt]; cnt += 1.f; }
    }
    acc = wave_sum(acc);
    cnt = wave_sum(cnt);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[w] = acc; s_cnt[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += s_sum[i]; c += s_cnt[i]; }
        loss_sum[0] = a;
        count[0] = c;
    }
}

int sequence_nll(const float *logp, const int64_t *targets, int B, int T, int V, int pad, float *loss_sum,
                 float *count, float *dlogp, hipStream_t stream) {
    hipLaunchKernelGGL(sequence_nll_kernel, dim3(1), dim3(1024), 0, stream, logp, targets, B, T, V, pad, loss_sum,
                       count, dlogp);
    GSCAN_LAUNCHED("sequence_nll_kernel");
    return 0;
}

// Model.get_auxiliary_loss (model.py:162-164): sum_b -aux_logp[b, pos[b]]
__global__ void position_nll_kernel(const float *__restrict__ aux, const int64_t *__restrict__ pos, int B, int M,
                                    float *loss_sum, float *__restrict__ daux) {
    __shared__ float s_sum[16];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int64_t p = pos[b];
        const bool ok = p >= 0 && p < M;
        if (daux)
            for (int j = 0; j < M; ++j) daux[(int64_t)b * M + j] = (ok && j == p) ? -1.f : 0.f;
        if (ok) acc -= aux[(int64_t)b * M + p];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) a += s_sum[i];
        loss_sum[0] = a;
    }
}

int position_nll(const float *aux, const int64_t *pos, int B, int M, float *loss_sum, float *daux,
                 hipStream_t stream) {
    hipLaunchKernelGGL(position_nll_kernel, dim3(1), dim3(256), 0, stream, aux, pos, B, M, loss_sum, daux);
    GSCAN_LAUNCHED("position_nll_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Both losses of the training step in one launch (train.py:102-107): stats = [sum NLL, tokens, sum aux NLL, rows],
// unit seeds d(sum NLL)/d(logp) and d(sum aux NLL)/d(aux_logp).  Single workgroup, fixed reduction order.
// ------------------------------------------------------------------------------------------
__global__ void step_losses_kernel(const float *__restrict__ logp, const int64_t *__restrict__ targets,
                                   const float *__restrict__ aux, const int64_t *__restrict__ pos, int B, int T, int V,
                                   int M, int pad, float *stats, float *__restrict__ dlogp, float *__restrict__ daux) {
    TraceScope trace_scope(TK_LOSS);
    __shared__ float s_sum[16], s_cnt[16], s_aux[16];
    float acc = 0.f, cnt = 0.f, acc_aux = 0.f;
    const int n = B * T;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int t = i % T;
        const int64_t tgt = (t + 1 < T) ? targets[i + 1] : (int64_t)0;
        const bool live = tgt != pad && tgt >= 0 && tgt < V;
        for (int j = 0; j < V; ++j) dlogp[(int64_t)i * V + j] = (live && j == tgt) ? -1.f : 0.f;
        if (live) { acc -= logp[(int64_t)i * V + tgt]; cnt += 1.f; }
    }
    if (aux) {
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            const int64_t p = pos[b];
            const bool ok = p >= 0 && p < M;
            for (int j = 0; j < M; ++j) daux[(int64_t)b * M + j] = (ok && j == p) ? -1.f : 0.f;
            if (ok) acc_aux -= aux[(int64_t)b * M + p];
        }
    }
    acc = wave_sum(acc); cnt = wave_sum(cnt); acc_aux = wave_sum(acc_aux);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[w] = acc; s_cnt[w] = cnt; s_aux[w] = acc_aux; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f, x = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += s_sum[i]; c += s_cnt[i]; x += s_aux[i]; }
        stats[0] = a; stats[1] = c; stats[2] = x; stats[3] = (float)B;
    }
}

int step_losses(const float *logp, const int64_t *targets, const float *aux, const int64_t *pos, int B, int T, int V,
                int M, int pad, float *stats, float *dlogp, float *daux, hipStream_t stream) {
    hipLaunchKernelGGL(step_losses_kernel, dim3(1), dim3(1024), 0, stream, logp, targets, aux, pos, B, T, V, M, pad,
                       stats, dlogp, daux);
    GSCAN_LAUNCHED("step_losses_kernel");
    return 0;
}

// seeds[0] = 1/tokens, seeds[1] = w/rows, seeds[2] = loss = sum NLL / tokens (+ w * sum aux NLL / rows)
__global__ void loss_seeds_kernel(const float *stats, float w, int auxiliary, float *seeds) {
    TraceScope trace_scope(TK_LOSS);
    if (threadIdx.x == 0) {
        const float s0 = 1.f / stats[1];
        const float s1 = auxiliary ? w / stats[3] : 0.f;
        seeds[0] = s0;
        seeds[1] = s1;
        seeds[2] = stats[0] * s0 + (auxiliary ? stats[2] * s1 : 0.f);
    }
}
int loss_seeds(const float *stats, float w, int auxiliary, float *seeds, hipStream_t stream) {
    hipLaunchKernelGGL(loss_seeds_kernel, dim3(1), dim3(64), 0, stream, stats, w, auxiliary, seeds);
    GSCAN_LAUNCHED("loss_seeds_kernel");
    return 0;
}

// Model.get_metrics (model.py:117-137): argmax accuracy and exact match under the pad mask.
// One lane per batch row; out[0] += correct, out[1] += live, out[2] += exact rows.
__global__ void sequence_metrics_kernel(const float *__restrict__ logp, const int64_t *__restrict__ targets, int B,
                                        int T, int V, int pad, float *out3) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    float correct = 0.f, live = 0.f, exact = 0.f;
    if (b < B) {
        for (int t = 0; t < T; ++t) {
            const int64_t tgt = (t + 1 < T) ? targets[(int64_t)b * T + t + 1] : (int64_t)0;
            if (tgt == pad) continue;
            const float *row = logp + ((int64_t)b * T + t) * V;
            int best = 0;
            for (int j = 1; j < V; ++j) if (row[j] > row[best]) best = j;   // first maximum, as torch.max
            live += 1.f;
            correct += (best == tgt) ? 1.f : 0.f;
        }
        exact = (correct == live) ? 1.f : 0.f;
    }
    correct = wave_sum(correct);
    live = wave_sum(live);
    exact = wave_sum(exact);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out3[0], correct);
        atomicAdd(&out3[1], live);
        atomicAdd(&out3[2], exact);
    }
}

int sequence_metrics(const float *logp, const int64_t *targets, int B, int T, int V, int pad, float *out3,
                     hipStream_t stream) {
    GSCAN_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(float), stream));
    hipLaunchKernelGGL(sequence_metrics_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, logp, targets, B, T, V, pad,
                       out3);
    GSCAN_LAUNCHED("sequence_metrics_kernel");
    return 0;
}

GSCAN_TRACE_TU(loss)

}  // namespace gscan
