// Backward of the two attention memories once the decoder's reverse recurrence has produced the per-step
// gradients, ONE WORKGROUP PER BATCH ROW, all products on the matrix cores:
//   value path  dPK[m,:] += sum_t alpha[t,m] * dctx[t,:]            (context = alpha . PK, seq2seq_model.py:138-139)
//   key layers  d enc_out = dPK_text . W_key_text                   (seq2seq_model.py:468-469)
//               d feat    = (dPK_vis . W_key_vis) * dropout mask, zero where ReLU was inactive
//                                                                    (seq2seq_model.py:466-467, cnn_model.py:33-35)
//   bridge      d h_N     = d h0 . W_bridge                          (model.py:195)
// A row's memories are small (L + G*G <= 128 keys of H floats): its dPK stays in LDS between the two stages, so the
// chain "value path -> key layers" is one launch instead of a batched reduction plus a grouped GEMM with an HBM
// round trip in between.  The totals dPK_text / dPK_vis are also written out: the key-layer weight gradients
// are dense products over them (leaves of the step's schedule).
#include "step.h"

namespace gscan {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kKbThreads = 512, kKbWaves = kKbThreads / 64, kKbSteps = 32, kKbMaxTiles = 8;
static_assert(kKbThreads == 16 * kKbSteps, "staging maps 16 lanes to a step");

struct KeysLds { int dpk, al, dc, dh, total; };
__host__ __device__ inline KeysLds keys_lds(int H, int L, int M) {
    const int HS = (H % 8 == 4) ? H : H + 4, MT = (M + 15) / 16 + (L + 15) / 16;
    KeysLds o;
    int p = 0;
    o.dpk = p; p += MT * 16 * HS;
    o.al = p;  p += kKbSteps * MT * 16;
    o.dc = p;  p += kKbSteps * 2 * H;
    o.dh = p;  p += (H + 3) / 4 * 4;
    o.total = p;
    return o;
}

template <int H>
__global__ __launch_bounds__(kKbThreads) void keys_backward_kernel(KeysBackwardArgs a) {
    TraceScope trace_scope(TK_KEYS_BWD);
    constexpr int HS = (H % 8 == 4) ? H : H + 4;      // dPK row stride: the 16 rows of an A fragment hit distinct banks
    constexpr int NTH = (H + 15) / 16, KS = H / 4;
    static_assert(NTH <= kKbMaxTiles && H % 4 == 0, "hidden size not supported");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int T = a.T, L = a.L, M = a.M;
    const int MTV = (M + 15) / 16, MTT = (L + 15) / 16, MT = MTV + MTT, AS = MT * 16;
    const KeysLds o = keys_lds(H, L, M);
    float *dpk_s = sm + o.dpk, *al_s = sm + o.al, *dc_s = sm + o.dc, *dh_s = sm + o.dh;

    // ---- stage 1: dPK = (score path, from the decoder kernel) + alpha^T . dctx, tiles of 16 memories x 16 features.
    //      Memory tiles: the first MTV cover the grid cells, the rest the command tokens (both zero-padded to 16).
    const int ntiles = MT * NTH;
    f32x4 acc[kKbMaxTiles];
#pragma unroll
    for (int i = 0; i < kKbMaxTiles; ++i) {
        const int tile = wave + kKbWaves * i;
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tile < ntiles) {
            const int mt = tile / NTH, nt = tile - mt * NTH, k = 16 * nt + fr;
            const bool vis = mt < MTV;
            const int mx = vis ? M : L, m0 = 16 * (vis ? mt : mt - MTV) + 4 * fg;
            const float *src = (vis ? a.dpk_v : a.dpk_t) + (int64_t)b * mx * H;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (m0 + r < mx && k < H) acc[i][r] = src[(m0 + r) * H + k];
        }
    }
    if (tid < H) dh_s[tid] = a.dh0[(int64_t)b * H + tid];
    for (int t0 = 0; t0 < T; t0 += kKbSteps) {
        const int n = min(kKbSteps, T - t0);
        const int64_t bt0 = (int64_t)b * T + t0;
        {   // thread = (step t, 16 lanes across the row): no divisions, every load of a thread in flight at once
            const int t = tid >> 4, q = tid & 15;
            const bool live = t < n;
            float av[kKbMaxTiles];
#pragma unroll
            for (int i = 0; i < kKbMaxTiles; ++i) {                         // alpha, [step][padded memories]
                const int c = q + 16 * i;
                av[i] = 0.f;
                if (live && i < MT) {
                    if (i < MTV) { if (c < M) av[i] = a.alpha_s[(bt0 + t) * M + c]; }
                    else if (c - MTV * 16 < L) av[i] = a.alpha_c[(bt0 + t) * L + c - MTV * 16];
                }
            }
            constexpr int NQ = (2 * H / 4 + 15) / 16;                       // [step][dctx_text | dctx_vis], 16-byte loads
            float4 dv[NQ];
            const float4 *src4 = reinterpret_cast<const float4 *>(a.ds + (bt0 + (live ? t : 0)) * 4 * H + H);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int c4 = q + 16 * i;
                dv[i] = float4{0.f, 0.f, 0.f, 0.f};
                if (live && c4 < 2 * H / 4) dv[i] = src4[c4];
            }
#pragma unroll
            for (int i = 0; i < kKbMaxTiles; ++i)
                if (i < MT) al_s[t * AS + q + 16 * i] = av[i];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int c4 = q + 16 * i;
                if (c4 < 2 * H / 4) *reinterpret_cast<float4 *>(dc_s + t * 2 * H + 4 * c4) = dv[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kKbMaxTiles; ++i) {
            const int tile = wave + kKbWaves * i;
            if (tile < ntiles) {
                const int mt = tile / NTH, nt = tile - mt * NTH;
                const float *ap = al_s + fg * AS + 16 * mt + fr;                               // A(m, t) = alpha[t][m]
                const float *bp = dc_s + fg * 2 * H + (mt < MTV ? H : 0) + min(16 * nt + fr, H - 1);   // B(t, k)
#pragma unroll
                for (int s = 0; s < kKbSteps / 4; ++s)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * AS], bp[4 * s * 2 * H], acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < kKbMaxTiles; ++i) {
        const int tile = wave + kKbWaves * i;
        if (tile < ntiles) {
            const int mt = tile / NTH, nt = tile - mt * NTH, k = 16 * nt + fr;
            const bool vis = mt < MTV;
            const int mx = vis ? M : L, m0 = 16 * (vis ? mt : mt - MTV) + 4 * fg;
            float *dst = (vis ? a.dpk_v : a.dpk_t) + (int64_t)b * mx * H;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k < H) {
                    dpk_s[(16 * mt + 4 * fg + r) * HS + k] = acc[i][r];       // padded memories: exact zeros
                    if (m0 + r < mx) dst[(m0 + r) * H + k] = acc[i][r];
                }
            }
        }
    }
    __syncthreads();

    // ---- stage 2: through the key layers.  A job = one 16-column tile of d feat (F columns, MTV memory tiles) or
    //      of d enc_out (He columns, MTT memory tiles); the job's B fragments (H/4 steps) are read once.
    //      Latency, not arithmetic, bounds this stage (a wave runs two or three jobs back to back and each starts
    //      with loads): the NEXT job's B fragments and the current job's ReLU gates / dropout masks are requested
    //      before the current job's MFMAs, so every load has a whole job to arrive.
    const int NTF = (a.F + 15) / 16, NTE = (a.He + 15) / 16, njobs = NTF + NTE;
    auto load_b = [&](int job, float (&bw)[KS]) {
        const bool vis = job < NTF;
        const int nt = vis ? job : job - NTF, ncols = vis ? a.F : a.He, col = 16 * nt + fr;
        const float *wsrc = vis ? a.w_kv : a.w_kt;
#pragma unroll
        for (int s = 0; s < KS; ++s) bw[s] = (job < njobs && col < ncols) ? wsrc[(int64_t)(4 * s + fg) * ncols + col] : 0.f;
    };
    auto run_job = [&](int job, const float (&bw)[KS]) {
        const bool vis = job < NTF;
        const int nt = vis ? job : job - NTF, ncols = vis ? a.F : a.He, col = 16 * nt + fr;
        const int mt_lo = vis ? 0 : MTV, nmt = vis ? MTV : MTT, mx = vis ? M : L;
        constexpr int kMaxMt = 4;                           // 64 memories per attention at most
        float gate[kMaxMt][4], mk[kMaxMt][4];
#pragma unroll
        for (int i = 0; i < kMaxMt; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * i + 4 * fg + r;
                gate[i][r] = 1.f; mk[i][r] = 1.f;
                if (vis && i < nmt && m < mx && col < ncols) {
                    const int64_t at = ((int64_t)b * mx + m) * ncols + col;
                    gate[i][r] = a.feat[at];
                    if (a.mask) mk[i][r] = a.mask[at];
                }
            }
#pragma unroll
        for (int i = 0; i < kMaxMt; ++i) {
            if (i >= nmt) continue;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            const float *ap = dpk_s + (16 * (mt_lo + i) + fr) * HS + fg;
#pragma unroll
            for (int s = 0; s < KS; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s], bw[s], c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * i + 4 * fg + r;
                if (m < mx && col < ncols) {
                    const int64_t at = ((int64_t)b * mx + m) * ncols + col;
                    // feat = relu(conv) * mask  =>  d conv = (feat != 0) ? d feat * mask : 0
                    if (vis) a.dfeat[at] = (gate[i][r] == 0.f) ? 0.f : c[r] * mk[i][r];
                    else a.denc[at] = c[r];
                }
            }
        }
    };
    {
        float bw0[KS], bw1[KS];
        load_b(wave, bw0);
        for (int job = wave; job < njobs; job += 2 * kKbWaves) {
            load_b(job + kKbWaves, bw1);
            run_job(job, bw0);
            if (job + kKbWaves >= njobs) break;
            load_b(job + 2 * kKbWaves, bw0);
            run_job(job + kKbWaves, bw1);
        }
    }
    // ---- bridge: d h_N[e] = sum_k d h0[k] * W_bridge[k][e]
    if (tid < a.He) {
        float wv[H];
#pragma unroll
        for (int k = 0; k < H; ++k) wv[k] = a.w_b[(int64_t)k * a.He + tid];   // all loads in flight
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            s0 = fmaf(dh_s[k], wv[k], s0);
            s1 = fmaf(dh_s[k + 1], wv[k + 1], s1);
            s2 = fmaf(dh_s[k + 2], wv[k + 2], s2);
            s3 = fmaf(dh_s[k + 3], wv[k + 3], s3);
        }
        a.dhN[(int64_t)b * a.He + tid] = (s0 + s1) + (s2 + s3);
    }
}

template <int H>
static int launch_keys_backward(int B, const KeysBackwardArgs &a, hipStream_t stream) {
    const size_t bytes = (size_t)keys_lds(H, a.L, a.M).total * sizeof(float);
    GSCAN_CHECK(bytes <= 160 * 1024, "keys backward: %zu bytes of LDS needed (L=%d cells=%d)", bytes, a.L, a.M);
    static bool attr_set = false;
    if (!attr_set) {
        GSCAN_HIP(hipFuncSetAttribute((const void *)keys_backward_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((keys_backward_kernel<H>), dim3(B), dim3(kKbThreads), bytes, stream, a);
    GSCAN_LAUNCHED("keys_backward_kernel");
    return 0;
}

int keys_backward(int B, int H, const KeysBackwardArgs &a, hipStream_t stream) {
    GSCAN_CHECK(B > 0 && a.T > 0 && a.L > 0 && a.M > 0 && a.L <= 64 && a.M <= 64 && a.He > 0 && a.F > 0 && a.He <= 512,
                "keys backward: bad dims B=%d T=%d L=%d cells=%d He=%d F=%d", B, a.T, a.L, a.M, a.He, a.F);
    // algorithmic flops: the data-gradient halves of the key layers and of the bridge (SURVEY.md 8d counts backward as
    // 2 x forward MACs: half of it data gradients), 2 * B * (L He H + M F H + He H); the value-path sums
    // dPK += alpha^T . dctx are the data-gradient halves of the context reductions, 2 * B * T * (L + M) * H
    const double alg = 2.0 * B * ((double)a.L * a.He * H + (double)a.M * a.F * H + (double)a.He * H) +
                       2.0 * B * a.T * (double)(a.L + a.M) * H;
    ProbeScope probe(P_KEYS_BWD, stream, alg, alg);
    switch (H) {
#define X(n) case n: return launch_keys_backward<n>(B, a, stream);
        GSCAN_DEC_HIDDEN_SIZES(X)
#undef X
        default: break;
    }
    GSCAN_CHECK(false, "keys backward: decoder_hidden_size %d has no compiled kernel (" GSCAN_DEC_HIDDEN_LIST ")", H);
}

GSCAN_TRACE_TU(attention_grad)

}  // namespace gscan
