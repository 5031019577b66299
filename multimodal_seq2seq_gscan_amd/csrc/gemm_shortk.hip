// Single-shot variant of the grouped fp32 GEMM (gemm.hip) for products whose WHOLE K extent fits LDS: K <= 156, both
// operands k-contiguous, no split — every product of the forward pass's one GEMM launch (keys, gate images, bridge,
// embedding part of the gates: K = 100 or 150).
//
// Why.  In gemm.hip a workgroup's life is K / 32 rounds of [issue the next panels' loads, fragment reads + MFMAs, wait,
// stage, barrier]; with K = 150 that is five rounds of ~2 000 cycles of which 512 are MFMAs, behind 3 400 cycles of
// set-up and first panels (tools/gemm_stamps.py) — the matrix cores are busy 30 % of the launch.  Here a workgroup
// issues ALL loads of its 64 x K and 64 x K panels at once (one latency, not five), stages them, passes ONE barrier and
// then runs its 2 x 2 x K / 4 MFMAs per wave back to back with the next 32-deep chunk's fragments read while the
// current chunk multiplies.  Two workgroups fit a CU (2 x 78 KB of LDS): one loads while the other multiplies.
//
// Geometry: tile 64 x 64, 4 waves as 2 x 2, wave tile 32 x 32 = 2 x 2 MFMA tiles (v_mfma_f32_16x16x4_f32).  LDS image of
// a panel: [row][156] floats (156 = -4 mod 32: the b128 fragment reads of 16 rows x 4 lane groups are two lanes per
// bank, the best 64 lanes can do).  Within a full 32-deep chunk lane group g reads k = 8 g .. 8 g + 7 (two b128 reads
// feed eight MFMA steps); the tail of K (< 32) runs in 4-deep steps with k = 4 s + g (scalar reads), zero-filled to a
// multiple of 4.  Loads are VW = 2 or 4 floats wide (what both operands' alignment allows); a thread's loads walk the
// panel's (row, chunk) pairs in row-major order, so rows that are contiguous in memory (row stride = K: the conv features,
// the composite weights) are read as one contiguous block.
// Same GemmProblem / GemmGroup, tile order and epilogues as gemm.hip.
#include "gemm_panel.h"

namespace gscan {

constexpr int SK_LDK = 156, SK_BM = 64, SK_THREADS = 256;
constexpr size_t kShortKLdsBytes = 2 * (size_t)SK_BM * SK_LDK * sizeof(float);
extern __shared__ __attribute__((aligned(16))) float shortk_lds[];

bool gemm_shortk_supports(const GemmProblem &p) {
    const int wa = p.flags & 3, wb = (p.flags >> 2) & 3;
    return p.sak == 1 && p.sbk == 1 && p.atomic == 0 && p.k_chunk >= p.K && p.K <= SK_LDK - 4 && wa >= 1 && wb >= 1 &&
           p.asum1 == nullptr;
}

// What a workgroup keeps of a tile between issuing its loads and finishing its epilogue (all wave-uniform).
struct ShortKTile {
    float *c; const float *bias, *mask, *gate;
    uint32_t ldc;
    float alpha, beta;
    int M, N, K, act, m0, n0;
};
constexpr int SK_NL = 20;                   // 8-byte loads per thread and operand: 64 rows x 80 chunks over 256 threads
using sk_vec = __attribute__((ext_vector_type(2))) float;

// (row, k) of load j of thread tid in a panel of KC two-float chunks per row
__device__ __forceinline__ void sk_where(int tid, int j, int KC, uint32_t inv_kc, int &row, int &k) {
    const uint32_t i = (uint32_t)(tid + SK_THREADS * j);
    row = (int)__umulhi(i, inv_kc);
    k = 2 * ((int)i - row * KC);
}

// Tile u of the launch: its problem, its place in it (XCD-aware order as in gemm.hip), and ALL loads of its two panels
// issued into xa / xb.  Returns false for a padding slot of the XCD order (nothing issued).
__device__ __forceinline__ bool sk_fetch(const int (&tb)[kMaxGroup], const GemmGroup &grp, int u, ShortKTile &t,
                                         sk_vec (&xa)[SK_NL], sk_vec (&xb)[SK_NL]) {
    int pi = 0, first = tb[0];
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if (u >= tb[i]) { pi = i; first = tb[i]; }
    const int per = grp.xcd_per[pi];
    const GemmProblem &g = grp.p[pi];
    int local = u - first;
    const int tiles_mn = g.tiles_mn;
    if (per > 0) {
        const int x = local & 7, j = local >> 3;
        local = x * per + j;
        if (j >= per || local >= tiles_mn) return false;
    }
    const int flags = g.flags, tiles_m = g.tiles_m, tiles_n = g.tiles_n;
    const uint32_t inv_in = g.inv_in;
    const bool n_major = (flags & 16) != 0;
    const int inner = inv_in ? (int)__umulhi((uint32_t)local, inv_in) : local;
    const int by = n_major ? local - inner * tiles_m : inner, bx = n_major ? inner : local - inner * tiles_n;
    t.c = g.c; t.bias = g.bias; t.mask = g.mask; t.gate = g.gate; t.ldc = (uint32_t)g.ldc;
    t.alpha = g.alpha; t.beta = g.beta; t.M = g.M; t.N = g.N; t.K = g.K; t.act = g.act;
    t.m0 = by * SK_BM; t.n0 = bx * BN;
    const int K = t.K, kz = ((K >> 5) << 5) + ((((K & 31) + 3) >> 2) << 2), KC = kz >> 1;
    const uint32_t inv_kc = 0xFFFFFFFFu / (uint32_t)KC + 1u;
    const gfloat *ga = as_global(g.a), *gb = as_global(g.b);
    const uint32_t sam = (uint32_t)g.sam, sbn = (uint32_t)g.sbn;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < SK_NL; ++j) {                 // unconditional: a dead load reads element 0 of the operand
        int row, k;
        sk_where(tid, j, KC, inv_kc, row, k);
        const bool la = row < SK_BM && t.m0 + row < t.M && k < K, lb = row < BN && t.n0 + row < t.N && k < K;
        xa[j] = *reinterpret_cast<const GSCAN_GLOBAL sk_vec *>(ga + (la ? (uint32_t)(t.m0 + row) * sam + (uint32_t)k : 0u));
        xb[j] = *reinterpret_cast<const GSCAN_GLOBAL sk_vec *>(gb + (lb ? (uint32_t)(t.n0 + row) * sbn + (uint32_t)k : 0u));
    }
    return true;
}

// registers -> LDS images [row][SK_LDK]; dead elements (tile edges, k >= K up to the zero-filled extent) become zeros
__device__ __forceinline__ void sk_stage(const ShortKTile &t, const sk_vec (&xa)[SK_NL], const sk_vec (&xb)[SK_NL]) {
    float *lds_a = shortk_lds, *lds_b = shortk_lds + SK_BM * SK_LDK;
    const int K = t.K, kz = ((K >> 5) << 5) + ((((K & 31) + 3) >> 2) << 2), KC = kz >> 1;
    const uint32_t inv_kc = 0xFFFFFFFFu / (uint32_t)KC + 1u;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < SK_NL; ++j) {
        int row, k;
        sk_where(tid, j, KC, inv_kc, row, k);
        if (row < SK_BM) {
            const bool la = t.m0 + row < t.M && k < K, lb = t.n0 + row < t.N && k < K;
            *reinterpret_cast<sk_vec *>(lds_a + row * SK_LDK + k) = la ? xa[j] : sk_vec{0.f, 0.f};
            *reinterpret_cast<sk_vec *>(lds_b + row * SK_LDK + k) = lb ? xb[j] : sk_vec{0.f, 0.f};
        }
    }
}

// MFMAs of the staged tile (full 32-deep chunks with the next chunk's fragments in flight, then the 4-deep tail steps)
// and its epilogue (gemm.hip's, natural tiles on both sides)
__device__ __forceinline__ void sk_multiply(const ShortKTile &t, f32x4 (&acc)[2][2]) {
    const float *lds_a = shortk_lds, *lds_b = shortk_lds + SK_BM * SK_LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int K = t.K, n32 = K >> 5, tail = ((K & 31) + 3) >> 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const float *fa = lds_a + (wm * 32 + fr) * SK_LDK, *fb = lds_b + (wn * 32 + fr) * SK_LDK;
    float4 af[2][2][2], bf[2][2][2];                                 // [buffer][tile][half]
    auto read_chunk = [&](int c, int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 *qa = reinterpret_cast<const float4 *>(fa + 16 * q * SK_LDK + 32 * c + 8 * fg);
            const float4 *qb = reinterpret_cast<const float4 *>(fb + 16 * q * SK_LDK + 32 * c + 8 * fg);
            af[buf][q][0] = qa[0]; af[buf][q][1] = qa[1];
            bf[buf][q][0] = qb[0]; bf[buf][q][1] = qb[1];
        }
    };
    auto mma_chunk = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 &x = af[buf][i][h], &y = bf[buf][j][h];
                        const float a = s == 0 ? x.x : s == 1 ? x.y : s == 2 ? x.z : x.w;
                        const float b = s == 0 ? y.x : s == 1 ? y.y : s == 2 ? y.z : y.w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][j], 0, 0, 0);
                    }
    };
    if (n32 > 0) read_chunk(0, 0);
    for (int c = 0; c + 1 < n32; c += 2) {                           // two chunks per trip: the buffers are compile-time
        read_chunk(c + 1, 1);
        mma_chunk(0);
        if (c + 2 < n32) read_chunk(c + 2, 0);
        mma_chunk(1);
    }
    if (n32 & 1) mma_chunk(0);
    for (int s = 0; s < tail; ++s) {
        const int k = 32 * n32 + 4 * s + fg;
        float a[2], b[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { a[q] = fa[16 * q * SK_LDK + k]; b[q] = fb[16 * q * SK_LDK + k]; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

__device__ __forceinline__ void sk_epilogue(const ShortKTile &t, const f32x4 (&acc)[2][2]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    // C fragment: column = lane & 15, row = (lane >> 4) * 4 + reg
    int coff[2];
    bool cok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        coff[j] = t.n0 + wn * 32 + 16 * j + fr;
        cok[j] = coff[j] < t.N;
    }
    const uint32_t ldc = t.ldc;
    const float alpha = t.alpha;
    gfloat *gc = as_global(t.c);
    const gfloat *gbias = as_global(t.bias), *ggate = as_global(t.gate), *gmask = as_global(t.mask);
    const bool plain = t.beta == 0.f && !t.bias && t.act == 0 && !t.mask;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = t.m0 + wm * 32 + 16 * i + fg * 4 + r;
            if (row >= t.M) continue;
            const uint32_t roff = (uint32_t)row * ldc;
            if (plain) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (cok[j]) gc[roff + coff[j]] = alpha * acc[i][j][r];
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!cok[j]) continue;
                    const uint32_t at = roff + coff[j];
                    float v = alpha * acc[i][j][r];
                    if (t.beta != 0.f) v += t.beta * gc[at];
                    if (gbias) v += gbias[coff[j]];
                    if (t.act == 1) v = fmaxf(v, 0.f);
                    else if (t.act == 2) v = tanhf_(v);
                    else if (t.act == 3 && ggate[at] == 0.f) v = 0.f;
                    if (gmask) v *= gmask[at];
                    gc[at] = v;
                }
            }
        }
}

// PERSISTENT: the launch has at most two workgroups per CU, each walks tiles u = block, block + grid, ...  The loads of
// a workgroup's NEXT tile are issued (into 80 staging registers) before the MFMAs of its current one, so a tile's load
// latency hides behind a tile's multiplication instead of one of the two resident workgroups' turns.
__global__ __launch_bounds__(SK_THREADS) void gemm_shortk_kernel(int tb0, int tb1, int tb2, int tb3, int tb4, int tb5, int tb6,
                                                                 int tb7, int tb8, int tb9, int tb10, int tb11, GemmGroup grp,
                                                                 int total) {
    const int tb[kMaxGroup] = {tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7, tb8, tb9, tb10, tb11};
    static_assert(kMaxGroup == 12, "the preloaded header is twelve scalars");
    TraceScope trace_scope(TK_GEMM);
    sk_vec xa[SK_NL], xb[SK_NL];
    ShortKTile cur, nxt;
    int u = blockIdx.x;
#ifdef GSCAN_GEMM_STAMPS
    unsigned gst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long gst_prev = clock64();
#endif
    bool have = u < total && sk_fetch(tb, grp, u, nxt, xa, xb);
    GST(0)                                            // first tile's descriptor and loads issued
    while (u < total) {
        cur = nxt;
        const bool cur_ok = have;
        if (cur_ok) sk_stage(cur, xa, xb);
        GST(1)                                        // wait for the panels + LDS stores
        // barriers that order LDS traffic only: __syncthreads() would also drain vmcnt, i.e. wait for the previous
        // tile's epilogue stores to be acknowledged and for the loads just issued for the next one
        lds_barrier();
        GST(2)
        u += gridDim.x;
        have = u < total && sk_fetch(tb, grp, u, nxt, xa, xb);
        GST(3)                                        // next tile's descriptor and loads issued
        f32x4 acc[2][2];
        if (cur_ok) sk_multiply(cur, acc);
        GST(4)                                        // fragment reads + MFMAs
        lds_barrier();                                // every wave is done with the LDS images
        GST(5)
        if (cur_ok) sk_epilogue(cur, acc);
        GST(6)
    }
#ifdef GSCAN_GEMM_STAMPS
    if (g_trace_buf && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_trace_buf[1500 + i] += gst_acc[i];
#endif
}

int gemm_shortk_launch(const GemmGroup &grp, int total, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_shortk_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kShortKLdsBytes));
        attr_set = true;
    }
    const int *t = grp.tile_begin;
    static const int cap = [] { const char *e = getenv("GSCAN_SHORTK_WGS"); return e ? atoi(e) : 512; }();   // two per CU
    hipLaunchKernelGGL(gemm_shortk_kernel, dim3(std::min(total, cap)), dim3(SK_THREADS), kShortKLdsBytes, stream, t[0], t[1],
                       t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11], grp, total);
    GSCAN_LAUNCHED("gemm_shortk_kernel");
    return 0;
}

GSCAN_TRACE_TU(gemm_shortk)

}  // namespace gscan
