// Backward of the two attention memories once the decoder's reverse recurrence has produced the per-step
// gradients, ONE WORKGROUP PER (BATCH ROW, TILE OF 16 MEMORIES), all products on the matrix cores:
//   value path  dPK[m,:] += sum_t alpha[t,m] * dctx[t,:]            (context = alpha . PK, seq2seq_model.py:138-139)
//   key layers  d enc_out = dPK_text . W_key_text                   (seq2seq_model.py:468-469)
//               d feat    = (dPK_vis . W_key_vis) * dropout mask, zero where ReLU was inactive
//                                                                    (seq2seq_model.py:466-467, cnn_model.py:33-35)
//   bridge      d h_N     = d h0 . W_bridge                          (model.py:195; one extra workgroup per row)
// A tile's dPK (16 memories x H floats) stays in LDS between the two stages, so the chain "value path -> key
// layers" is one launch instead of a batched reduction plus a grouped GEMM with an HBM round trip in between.
// The totals dPK_text / dPK_vis are also written out: the key-layer weight gradients are dense products over
// them (leaves of the step's schedule).
// Why tiles and not whole rows: this launch sits on the step's critical chain while the decoder's leaf products
// (1500+ GEMM workgroups of 4 waves x 128 VGPRs) run on a side stream.  A 512-thread, 168-VGPR, 61 KB workgroup
// only fits a CU once three of those have drained from it, and the dispatcher hands every freed slot to the
// other queue first: the row-per-workgroup version of this kernel waited ~30 us for placement (round-2 device
// timelines).  A 256-thread workgroup under 128 VGPRs has the footprint of a GEMM workgroup and interleaves with
// them; it also lifts the 64-memory limit of the row version.
#include "step.h"

namespace gscan {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kKbThreads = 256, kKbWaves = kKbThreads / 64, kKbSteps = 32;

struct KeysLds { int dpk, al, dc, total; };
__host__ __device__ inline KeysLds keys_lds(int H) {
    const int HS = (H % 8 == 4) ? H : H + 4;
    KeysLds o;
    int p = 0;
    o.dpk = p; p += 16 * HS;
    o.al = p;  p += kKbSteps * 16;
    o.dc = p;  p += kKbSteps * H;
    o.total = p;
    return o;
}

template <int H>
__global__ __launch_bounds__(kKbThreads, H <= 128 ? 4 : 2) void keys_backward_kernel(KeysBackwardArgs a) {   // hidden sizes above 128: 64 B-fragment registers per lane, two workgroups per CU
    TraceScope trace_scope(TK_KEYS_BWD);
    constexpr int HS = (H % 8 == 4) ? H : H + 4;      // dPK row stride: the 16 rows of an A fragment hit distinct banks
    constexpr int NTH = (H + 15) / 16, KS = H / 4, Q = H / 4;
    constexpr int kKbMaxTiles = (NTH + kKbWaves - 1) / kKbWaves;     // 16-feature tiles of dPK per wave (2 up to hidden 128, 4 up to 256)
    static_assert(H % 4 == 0 && H <= 256, "hidden size not supported");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int T = a.T, L = a.L, M = a.M;
    const int MTV = (M + 15) / 16, MTT = (L + 15) / 16, MT = MTV + MTT;
    const int mt = blockIdx.y;

    if (mt == MT) {
        if (a.value_path_only) return;
        // ---- bridge: d h_N[e] = sum_k d h0[k] * W_bridge[k][e]
        float *dh_s = sm;
        if (tid < H) dh_s[tid] = a.dh0[(int64_t)b * H + tid];
        __syncthreads();
        for (int e = tid; e < a.He; e += kKbThreads) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 5
            for (int k = 0; k < H; k += 4) {
                s0 = fmaf(dh_s[k], a.w_b[(int64_t)k * a.He + e], s0);
                s1 = fmaf(dh_s[k + 1], a.w_b[(int64_t)(k + 1) * a.He + e], s1);
                s2 = fmaf(dh_s[k + 2], a.w_b[(int64_t)(k + 2) * a.He + e], s2);
                s3 = fmaf(dh_s[k + 3], a.w_b[(int64_t)(k + 3) * a.He + e], s3);
            }
            a.dhN[(int64_t)b * a.He + e] = (s0 + s1) + (s2 + s3);
        }
        return;
    }

    const KeysLds o = keys_lds(H);
    float *dpk_s = sm + o.dpk, *al_s = sm + o.al, *dc_s = sm + o.dc;
    const bool vis = mt < MTV;
    const int mx = vis ? M : L, mbase = 16 * (vis ? mt : mt - MTV);
    float *dpk_g = (vis ? a.dpk_v : a.dpk_t) + (int64_t)b * mx * H;
    const float *alpha = vis ? a.alpha_s : a.alpha_c;
    const int dcol = vis ? 2 * H : H;                 // d ctx_text | d ctx_vis inside a dS row

    // ---- stage 1: dPK = (score path, from the decoder kernel) + alpha^T . dctx, tiles of 16 memories x 16 features
    f32x4 acc[kKbMaxTiles];
#pragma unroll
    for (int i = 0; i < kKbMaxTiles; ++i) {
        const int nt = wave + kKbWaves * i, k = 16 * nt + fr;
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nt < NTH && k < H) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mbase + 4 * fg + r;
                if (m < mx) acc[i][r] = dpk_g[m * H + k];
            }
        }
    }
    for (int t0 = 0; t0 < T; t0 += kKbSteps) {
        const int n = min(kKbSteps, T - t0);
        const int64_t bt0 = (int64_t)b * T + t0;
        {   // every load of a thread in flight at once
            constexpr int NA = kKbSteps * 16 / kKbThreads, ND = (kKbSteps * Q + kKbThreads - 1) / kKbThreads;
            float av[NA];
            float4 dv[ND];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = tid + kKbThreads * i, t = idx >> 4, q = idx & 15;
                av[i] = (t < n && mbase + q < mx) ? alpha[(bt0 + t) * mx + mbase + q] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int idx = tid + kKbThreads * i, t = idx / Q, c4 = idx - t * Q;
                dv[i] = float4{0.f, 0.f, 0.f, 0.f};
                if (t < n) dv[i] = *reinterpret_cast<const float4 *>(a.ds + (bt0 + t) * 4 * H + dcol + 4 * c4);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) al_s[tid + kKbThreads * i] = av[i];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int idx = tid + kKbThreads * i;
                if (idx < kKbSteps * Q) *reinterpret_cast<float4 *>(dc_s + 4 * idx) = dv[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kKbMaxTiles; ++i) {
            const int nt = wave + kKbWaves * i;
            if (nt < NTH) {
                const float *ap = al_s + fg * 16 + fr;                                  // A(m, t) = alpha[t][m]
                const float *bp = dc_s + fg * H + min(16 * nt + fr, H - 1);             // B(t, k)
#pragma unroll
                for (int s = 0; s < kKbSteps / 4; ++s)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * 16], bp[4 * s * H], acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < kKbMaxTiles; ++i) {
        const int nt = wave + kKbWaves * i, k = 16 * nt + fr;
        if (nt < NTH && k < H) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mbase + 4 * fg + r;
                dpk_s[(4 * fg + r) * HS + k] = acc[i][r];                               // padded memories: exact zeros
                if (m < mx) dpk_g[m * H + k] = acc[i][r];
            }
        }
    }
    if (a.value_path_only) return;
    __syncthreads();

    // ---- stage 2: through the key layer.  A job = one 16-column tile of d feat (F columns) or of d enc_out (He
    //      columns).  Each job starts with its loads (B fragments, ReLU gates, dropout masks); the four workgroups
    //      a CU holds cover each other's latency, so there is no software pipeline here (a double-buffered B
    //      fragment pushed the kernel past 128 VGPRs, and with it out of the footprint the header describes).
    const int ncols = vis ? a.F : a.He, njobs = (ncols + 15) / 16;
    const float *wsrc = vis ? a.w_kv : a.w_kt;
    for (int job = wave; job < njobs; job += kKbWaves) {
        const int col = 16 * job + fr;
        const bool col_ok = col < ncols;
        float bw[KS], gate[4], mk[4];
        {
            const float *wp = wsrc + (int64_t)fg * ncols + (col_ok ? col : 0);
#pragma unroll
            for (int s = 0; s < KS; ++s) { bw[s] = *wp; wp += 4 * ncols; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mbase + 4 * fg + r;
            gate[r] = 1.f; mk[r] = (a.mask == nullptr && a.mask_scale != 0.f) ? a.mask_scale : 1.f;   // drawn in the world encoder
            if (vis && m < mx && col_ok) {
                const int64_t at = ((int64_t)b * mx + m) * ncols + col;
                gate[r] = a.feat[at];
                if (a.mask) mk[r] = a.mask[at];
            }
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        const float *ap = dpk_s + fr * HS + fg;
#pragma unroll
        for (int s = 0; s < KS; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s], col_ok ? bw[s] : 0.f, c, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mbase + 4 * fg + r;
            if (m < mx && col_ok) {
                const int64_t at = ((int64_t)b * mx + m) * ncols + col;
                // feat = relu(conv) * mask  =>  d conv = (feat != 0) ? d feat * mask : 0
                if (vis) a.dfeat[at] = (gate[r] == 0.f) ? 0.f : c[r] * mk[r];
                else a.denc[at] = c[r];
            }
        }
    }
}

// The same launch for ANY decoder hidden size (the kernel above is compiled for the sizes whose recurrent weights fit
// the decoder's register file): plain loops, a thread per (memory, feature) of the tile in stage 1 and per (memory,
// output column) in stage 2, the tile's dPK in LDS between the two.  Correctness first (decoder_any.hip's companion).
__global__ __launch_bounds__(kKbThreads) void keys_backward_any_kernel(KeysBackwardArgs a, int H) {
    TraceScope trace_scope(TK_KEYS_BWD);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int T = a.T, L = a.L, M = a.M;
    const int MTV = (M + 15) / 16, MTT = (L + 15) / 16, MT = MTV + MTT;
    const int mt = blockIdx.y;
    if (mt == MT) {              // bridge: d h_N[e] = sum_k d h0[k] * W_bridge[k][e]
        if (a.value_path_only) return;
        float *dh_s = sm;
        for (int k = tid; k < H; k += kKbThreads) dh_s[k] = a.dh0[(int64_t)b * H + k];
        __syncthreads();
        for (int e = tid; e < a.He; e += kKbThreads) {
            float s0 = 0.f;
            for (int k = 0; k < H; ++k) s0 = fmaf(dh_s[k], a.w_b[(int64_t)k * a.He + e], s0);
            a.dhN[(int64_t)b * a.He + e] = s0;
        }
        return;
    }
    float *dpk_s = sm;                                   // [16][H]
    const bool vis = mt < MTV;
    const int mx = vis ? M : L, mbase = 16 * (vis ? mt : mt - MTV);
    float *dpk_g = (vis ? a.dpk_v : a.dpk_t) + (int64_t)b * mx * H;
    const float *alpha = vis ? a.alpha_s : a.alpha_c;
    const int dcol = vis ? 2 * H : H;
    // stage 1: dPK = (score path) + alpha^T . dctx
    for (int idx = tid; idx < 16 * H; idx += kKbThreads) {
        const int q = idx / H, k = idx - q * H, m = mbase + q;
        float acc = 0.f;
        if (m < mx) {
            acc = dpk_g[(int64_t)m * H + k];
            for (int t = 0; t < T; ++t) {
                const int64_t bt = (int64_t)b * T + t;
                acc = fmaf(alpha[bt * mx + m], a.ds[bt * 4 * H + dcol + k], acc);
            }
            dpk_g[(int64_t)m * H + k] = acc;
        }
        dpk_s[idx] = acc;
    }
    if (a.value_path_only) return;
    __syncthreads();
    // stage 2: through the key layer
    const int ncols = vis ? a.F : a.He;
    const float *wsrc = vis ? a.w_kv : a.w_kt;
    for (int idx = tid; idx < 16 * ncols; idx += kKbThreads) {
        const int q = idx / ncols, col = idx - q * ncols, m = mbase + q;
        if (m >= mx) continue;
        float acc = 0.f;
        for (int k = 0; k < H; ++k) acc = fmaf(dpk_s[q * H + k], wsrc[(int64_t)k * ncols + col], acc);
        const int64_t at = ((int64_t)b * mx + m) * ncols + col;
        if (vis) {               // feat = relu(conv) * mask  =>  d conv = (feat != 0) ? d feat * mask : 0
            const float gate = a.feat[at], mk = a.mask ? a.mask[at] : (a.mask_scale != 0.f ? a.mask_scale : 1.f);
            a.dfeat[at] = (gate == 0.f) ? 0.f : acc * mk;
        } else {
            a.denc[at] = acc;
        }
    }
}

// hidden sizes above the resident decoder kernels' 100 that still get the matrix-core kernel (any other size: the plain loops below)
#define GSCAN_KEYS_EXTRA_SIZES(X) X(104) X(108) X(112) X(116) X(120) X(124) X(128) X(144) X(160) X(176) X(192) X(200) X(208) X(224) X(240) X(256)

template <int H>
static int launch_keys_backward(int B, const KeysBackwardArgs &a, hipStream_t stream) {
    const size_t bytes = (size_t)keys_lds(H).total * sizeof(float);
    if (bytes > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&keys_backward_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)bytes));
            attr_set = true;
        }
    }
    const int tiles = (a.M + 15) / 16 + (a.L + 15) / 16;
    hipLaunchKernelGGL((keys_backward_kernel<H>), dim3(B, tiles + 1), dim3(kKbThreads), bytes, stream, a);
    GSCAN_LAUNCHED("keys_backward_kernel");
    return 0;
}

int keys_backward(int B, int H, const KeysBackwardArgs &a, hipStream_t stream) {
    GSCAN_CHECK(B > 0 && a.T > 0 && a.L > 0 && a.M > 0 && a.He > 0 && a.F > 0,
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
        GSCAN_KEYS_EXTRA_SIZES(X)      // round 5: the matrix-core kernel for the usual sizes above 100 too (the decoder streams there)
#undef X
        default: break;
    }
    {   // any other hidden size
        const size_t bytes = (size_t)16 * H * sizeof(float);
        GSCAN_CHECK(H >= 1 && bytes <= 160 * 1024, "keys backward: decoder_hidden_size %d", H);
        static bool attr_set = false;
        if (!attr_set) {
            GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(keys_backward_any_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        const int tiles = (a.M + 15) / 16 + (a.L + 15) / 16;
        hipLaunchKernelGGL(keys_backward_any_kernel, dim3(B, tiles + 1), dim3(kKbThreads), bytes, stream, a, H);
        GSCAN_LAUNCHED("keys_backward_any_kernel");
    }
    return 0;
}

// ---- sums over time per attention memory (long target sequences; see AlphaReduceArgs in step.h) -------------------------
// One workgroup per (batch row, pass of 128 columns, group of up to 4 memory tiles): the workgroups of a row split the COLUMNS
// of x = [delta | dzq], so every gradient row is read once per launch (a workgroup per memory tile read it once per tile:
// 43 -> 16 us at S3).  Chunks of 32 steps go through LDS — alpha of every memory tile of the group as A operands [memory, step],
// the chunk's 128 columns as the B operand [step, column] — a wave owns two 16-column tiles x all memory tiles, 16x16x4 MFMAs.
// Visual tiles skip the columns past 4H (dzq reaches the textual memories only).  Any hidden size (run-time column counts).
constexpr int kArThreads = 256, kArWaves = kArThreads / 64, kArSteps = 32, kArColTiles = 2, kArMemTiles = 4;
constexpr int kArCols = 16 * kArWaves * kArColTiles;       // 128 columns per pass
constexpr int kArStride = kArCols + 16;                    // row stride = 16 mod 32 banks: the four step rows of a B fragment read disjoint banks

__global__ __launch_bounds__(kArThreads, 4) void alpha_reduce_kernel(AlphaReduceArgs a) {
    TraceScope trace_scope(TK_KEYS_BWD);
    __shared__ __attribute__((aligned(16))) float al_s[kArMemTiles * kArSteps * 16];
    __shared__ __attribute__((aligned(16))) float x_s[kArSteps * kArStride];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int T = a.T, MTV = (a.M + 15) / 16, MT = MTV + (a.L + 15) / 16;
    const int g0 = blockIdx.z * kArMemTiles, ng = min(kArMemTiles, MT - g0);      // this workgroup's memory tiles: g0 .. g0 + ng
    const int c0 = blockIdx.y * kArCols;
    const bool any_text = g0 + ng > MTV;
    const int W = any_text ? max(a.wt, a.wv) : a.wv;                               // widest output among this group's tiles
    if (c0 >= W) return;
    const int wc = min(kArCols, W - c0);
    const bool v4 = ((a.ldx | c0) & 3) == 0 && (wc & 3) == 0;

    f32x4 acc[kArMemTiles][kArColTiles];
#pragma unroll
    for (int j = 0; j < kArMemTiles; ++j)
#pragma unroll
        for (int i = 0; i < kArColTiles; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a chunk's loads all in flight at once, and the NEXT chunk's requested before this chunk's MFMAs (the launch is a chain of
    // T / 32 round trips per workgroup otherwise: 46 us at S3 however little it reads)
    constexpr int NA = kArMemTiles * kArSteps * 16 / kArThreads, NX = kArSteps * (kArCols / 4) / kArThreads;
    float av[NA];
    float4 xv[NX];
    auto request = [&](int t0) {
        const int n = min(kArSteps, T - t0);
        const int64_t bt0 = (int64_t)b * T + t0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {     // alpha of the group's tiles: [tile][step][16 memories]; steps past the end and memories past the last are zeros
            static_assert(kArSteps * 16 == 2 * kArThreads, "two passes of the workgroup per tile of alpha");
            const int j = i >> 1, t = (tid >> 4) + (kArThreads / 16) * (i & 1), q = tid & 15;   // element tid + 256 i of [tile][step][memory]
            const int mt = g0 + j;                                                              // (uniform per i: tile, memory count, base pointer)
            const bool vis = mt < MTV;
            const int mx = vis ? a.M : a.L, m = 16 * (vis ? mt : mt - MTV) + q;
            const float *alpha = vis ? a.alpha_s : a.alpha_c;
            av[i] = (j < ng && t < n && m < mx) ? alpha[(bt0 + t) * mx + m] : 0.f;
        }
        if (v4) {                          // rows past the end of the sequence are zeros (0 * garbage)
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int idx = tid + kArThreads * i, tt = idx / (kArCols / 4), c4 = idx - tt * (kArCols / 4);
                xv[i] = float4{0.f, 0.f, 0.f, 0.f};
                if (tt < n && 4 * c4 < wc) xv[i] = *reinterpret_cast<const float4 *>(a.x + (bt0 + tt) * a.ldx + c0 + 4 * c4);
            }
        }
    };
    request(0);
    for (int t0 = 0; t0 < T; t0 += kArSteps) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + kArThreads * i;
            if (idx < ng * kArSteps * 16) al_s[idx] = av[i];
        }
        if (v4) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int idx = tid + kArThreads * i, tt = idx / (kArCols / 4), c4 = idx - tt * (kArCols / 4);
                if (4 * c4 < wc) *reinterpret_cast<float4 *>(x_s + tt * kArStride + 4 * c4) = xv[i];
            }
        } else {                           // odd widths: element by element, no prefetch
            const int n = min(kArSteps, T - t0);
            const int64_t bt0 = (int64_t)b * T + t0;
            for (int idx = tid; idx < kArSteps * wc; idx += kArThreads) {
                const int tt = idx / wc, c = idx - tt * wc;
                x_s[tt * kArStride + c] = tt < n ? a.x[(bt0 + tt) * a.ldx + c0 + c] : 0.f;
            }
        }
        __syncthreads();
        if (t0 + kArSteps < T) request(t0 + kArSteps);
#pragma unroll
        for (int i = 0; i < kArColTiles; ++i) {
            const int nt = wave + kArWaves * i;
            if (16 * nt < wc) {       // columns past wc inside the last tile hold stale LDS: they only reach output columns that are not stored
                const float *bp = x_s + fg * kArStride + 16 * nt + fr;       // B(t, c)
                float bv[kArSteps / 4];
#pragma unroll
                for (int s = 0; s < kArSteps / 4; ++s) bv[s] = bp[4 * s * kArStride];
#pragma unroll
                for (int j = 0; j < kArMemTiles; ++j) {
                    if (j < ng && (g0 + j >= MTV || c0 + 16 * nt < a.wv)) {
                        const float *ap = al_s + j * (kArSteps * 16) + fg * 16 + fr;          // A(m, t) = alpha[t][m]
#pragma unroll
                        for (int s = 0; s < kArSteps / 4; ++s)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * 16], bv[s], acc[j][i], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < kArMemTiles; ++j) {
        if (j >= ng) continue;
        const int mt = g0 + j;
        const bool vis = mt < MTV;
        const int mx = vis ? a.M : a.L, mbase = 16 * (vis ? mt : mt - MTV), Wj = vis ? a.wv : a.wt;
        float *g = (vis ? a.g_v : a.g_t) + (int64_t)b * mx * Wj;
#pragma unroll
        for (int i = 0; i < kArColTiles; ++i) {
            const int c = c0 + 16 * (wave + kArWaves * i) + fr;
            if (c < Wj) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mbase + 4 * fg + r;
                    if (m < mx) g[(int64_t)m * Wj + c] = acc[j][i][r];
                }
            }
        }
    }
}

int alpha_reduce(int B, const AlphaReduceArgs &a, hipStream_t stream) {
    GSCAN_CHECK(B > 0 && a.T > 0 && a.L > 0 && a.M > 0 && a.wt > 0 && a.wv > 0 && a.ldx >= a.wt && a.ldx >= a.wv,
                "alpha reduce: bad dims B=%d T=%d L=%d cells=%d columns=%d/%d stride=%d", B, a.T, a.L, a.M, a.wt, a.wv, a.ldx);
    const int tiles = (a.M + 15) / 16 + (a.L + 15) / 16, groups = (tiles + kArMemTiles - 1) / kArMemTiles;
    const int passes = (std::max(a.wt, a.wv) + kArCols - 1) / kArCols;
    const double flops = 2.0 * B * a.T * ((double)a.L * a.wt + (double)a.M * a.wv);
    ProbeScope probe(P_KEYS_BWD, stream, flops, 0.0);      // executed in place of per-step products the GEMM launches are credited with
    hipLaunchKernelGGL(alpha_reduce_kernel, dim3(B, passes, groups), dim3(kArThreads), 0, stream, a);
    GSCAN_LAUNCHED("alpha_reduce_kernel");
    return 0;
}

GSCAN_TRACE_TU(attention_grad)

}  // namespace gscan
