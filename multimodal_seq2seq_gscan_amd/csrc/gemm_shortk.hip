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

// NL = loads per thread and operand (64 rows x KC chunks over 256 threads)
template <int VW, int NL>
__device__ __forceinline__ void shortk_tile(const GemmProblem &g, int local) {
    using vec = __attribute__((ext_vector_type(VW))) float;
    float *lds_a = shortk_lds, *lds_b = shortk_lds + SK_BM * SK_LDK;
    const int rem = local;                                           // no K split: the tile index inside the problem
    const bool n_major = (g.flags & 16) != 0;
    const int inner = g.inv_in ? (int)__umulhi((uint32_t)rem, g.inv_in) : rem;
    const int by = n_major ? rem - inner * (int)g.tiles_m : inner, bx = n_major ? inner : rem - inner * g.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = by * SK_BM, n0 = bx * BN;
    const int K = g.K, n32 = K >> 5, tail = ((K & 31) + 3) >> 2, kz = 32 * n32 + 4 * tail;
    const int KC = kz / VW;                                          // chunks per row (kz is a multiple of 4)
    const uint32_t inv_kc = 0xFFFFFFFFu / (uint32_t)KC + 1u;         // i / KC = umulhi(i, inv_kc) for i * KC < 2^32

    // ---- all loads of both panels, then the LDS stores ------------------------------------------------------------
    vec xa[NL], xb[NL];
    const gfloat *ga = as_global(g.a), *gb = as_global(g.b);
    const uint32_t sam = (uint32_t)g.sam, sbn = (uint32_t)g.sbn;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const uint32_t i = (uint32_t)(tid + SK_THREADS * j);
        const int row = (int)__umulhi(i, inv_kc), k = VW * ((int)i - row * KC);
        const bool la = row < SK_BM && m0 + row < g.M && k < K, lb = row < BN && n0 + row < g.N && k < K;
        xa[j] = *reinterpret_cast<const GSCAN_GLOBAL vec *>(ga + (la ? (uint32_t)(m0 + row) * sam + (uint32_t)k : 0u));
        xb[j] = *reinterpret_cast<const GSCAN_GLOBAL vec *>(gb + (lb ? (uint32_t)(n0 + row) * sbn + (uint32_t)k : 0u));
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const uint32_t i = (uint32_t)(tid + SK_THREADS * j);
        const int row = (int)__umulhi(i, inv_kc), k = VW * ((int)i - row * KC);
        const bool la = row < SK_BM && m0 + row < g.M && k < K, lb = row < BN && n0 + row < g.N && k < K;
        if (row < SK_BM) {                                           // dead elements (tile edges, k >= K) are stored as zeros
            vec za = xa[j], zb = xb[j];
#pragma unroll
            for (int e = 0; e < VW; ++e) { za[e] = la ? za[e] : 0.f; zb[e] = lb ? zb[e] : 0.f; }
            *reinterpret_cast<vec *>(lds_a + row * SK_LDK + k) = za;
            *reinterpret_cast<vec *>(lds_b + row * SK_LDK + k) = zb;
        }
    }
    __syncthreads();

    // ---- MFMAs: full 32-deep chunks with the next chunk's fragments in flight, then the 4-deep tail steps -----------
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const float *fa = lds_a + (wm * 32 + fr) * SK_LDK, *fb = lds_b + (wn * 32 + fr) * SK_LDK;
    float4 af[2][2][2], bf[2][2][2];                                 // [buffer][tile][half]
    auto read_chunk = [&](int c, int buf) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 *qa = reinterpret_cast<const float4 *>(fa + 16 * t * SK_LDK + 32 * c + 8 * fg);
            const float4 *qb = reinterpret_cast<const float4 *>(fb + 16 * t * SK_LDK + 32 * c + 8 * fg);
            af[buf][t][0] = qa[0]; af[buf][t][1] = qa[1];
            bf[buf][t][0] = qb[0]; bf[buf][t][1] = qb[1];
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
        for (int t = 0; t < 2; ++t) { a[t] = fa[16 * t * SK_LDK + k]; b[t] = fb[16 * t * SK_LDK + k]; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue (gemm.hip's, natural tiles on both sides): C fragment column = lane & 15, row = (lane >> 4) * 4 + reg
    int coff[2];
    bool cok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        coff[j] = n0 + wn * 32 + 16 * j + fr;
        cok[j] = coff[j] < g.N;
    }
    const uint32_t ldc = (uint32_t)g.ldc;
    const float alpha = g.alpha;
    gfloat *gc = as_global(g.c);
    const gfloat *gbias = as_global(g.bias), *ggate = as_global(g.gate), *gmask = as_global(g.mask);
    const bool plain = g.beta == 0.f && !g.bias && g.act == 0 && !g.mask;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 32 + 16 * i + fg * 4 + r;
            if (row >= g.M) continue;
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
                    if (g.beta != 0.f) v += g.beta * gc[at];
                    if (gbias) v += gbias[coff[j]];
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    else if (g.act == 2) v = tanhf_(v);
                    else if (g.act == 3 && ggate[at] == 0.f) v = 0.f;
                    if (gmask) v *= gmask[at];
                    gc[at] = v;
                }
            }
        }
}

__global__ __launch_bounds__(SK_THREADS) void gemm_shortk_kernel(int tb0, int tb1, int tb2, int tb3, int tb4, int tb5, int tb6,
                                                                 int tb7, int tb8, int tb9, int tb10, int tb11, GemmGroup grp) {
    const int tb[kMaxGroup] = {tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7, tb8, tb9, tb10, tb11};
    static_assert(kMaxGroup == 12, "the preloaded header is twelve scalars");
    TraceScope trace_scope(TK_GEMM);
    int pi = 0, first = tb[0];
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if ((int)blockIdx.x >= tb[i]) { pi = i; first = tb[i]; }
    asm volatile("" : "+s"(pi), "+s"(first));
    int per = grp.xcd_per[pi];
    GemmProblem g = grp.p[pi];
    asm volatile("" : "+s"(g.M), "+s"(g.N), "+s"(g.K), "+s"(g.alpha), "+s"(g.beta), "+s"(g.a), "+s"(g.sam), "+s"(g.b),
                      "+s"(g.sbn), "+s"(g.c), "+s"(g.ldc), "+s"(per));
    asm volatile("" : "+s"(g.bias), "+s"(g.act), "+s"(g.mask), "+s"(g.gate), "+s"(g.tiles_n), "+s"(g.tiles_mn), "+s"(g.flags),
                      "+s"(g.inv_in), "+s"(g.tiles_m));
    int local = blockIdx.x - first;
    if (per > 0) {                                   // XCD-aware order, as in gemm.hip
        const int x = local & 7, j = local >> 3;
        local = x * per + j;
        if (j >= per || local >= g.tiles_mn) return;
    }
    const int vw = 1 << min(g.flags & 3, (g.flags >> 2) & 3);       // 2 or 4 floats per load
    const int kz = ((g.K >> 5) << 5) + ((((g.K & 31) + 3) >> 2) << 2);
    if (vw == 4) {
        if (kz <= 112) shortk_tile<4, 7>(g, local);                // 64 x 28 chunks
        else shortk_tile<4, 10>(g, local);                         // 64 x 40
    } else {
        if (kz <= 104) shortk_tile<2, 13>(g, local);               // 64 x 52
        else shortk_tile<2, 20>(g, local);                         // 64 x 80
    }
}

int gemm_shortk_launch(const GemmGroup &grp, int total, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_shortk_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kShortKLdsBytes));
        attr_set = true;
    }
    const int *t = grp.tile_begin;
    hipLaunchKernelGGL(gemm_shortk_kernel, dim3(total), dim3(SK_THREADS), kShortKLdsBytes, stream, t[0], t[1], t[2], t[3], t[4],
                       t[5], t[6], t[7], t[8], t[9], t[10], t[11], grp);
    GSCAN_LAUNCHED("gemm_shortk_kernel");
    return 0;
}

GSCAN_TRACE_TU(gemm_shortk)

}  // namespace gscan
