// Wide-tile variant of the grouped fp32 GEMM (gemm.hip) for launches whose products are TALL: thousands of rows,
// N and K of a few hundred (the forward pass's key / value / U-image / embedding products, the decoder's `dS +=`).
//
// Why a second tile shape.  A CU of this chip receives ~10 bytes per cycle from L2 (measured: 4096^3 on the 64 x 64
// tiles of gemm.hip runs at 8.7 B/cycle/CU, on 32 x 64 tiles at 11.1; the sparse world encoder's row gathers top out
// at the same figure) while its four matrix cores retire 256 flop per cycle.  A (BM x BN) tile re-reads
// 4 (BM + BN) bytes of operand per 2 BM BN flop of a K step, so it needs 2 BM BN / (4 (BM + BN)) >= 25.6 flop/B to be
// bound by the matrix cores: 32 x 64 has 10.7, 64 x 64 has 16, 128 x 80 has 24.6, 128 x 112 has 29.9.  The small tiles
// exist because most launches of the step have too few of the big ones to fill 256 CUs and hide a K round's latency
// with co-resident workgroups; the tall launches have enough.
//
// Geometry.  Workgroup tile 128 x (16 nf) x 32, nf = 1..8 chosen per problem so that N is covered with little padding
// (N = 400: five tiles of nf = 5; N = 100: one tile of nf = 7).  512 threads = eight waves as 4 (along M, 32 rows = 2 MFMA
// row tiles each) x 2 (the first ceil(nf / 2) column tiles, the rest): at most 2 x 4 accumulators of 16 x 16 per wave, under
// 128 VGPRs, so the two workgroups a CU holds put four waves on every SIMD (a 256-thread version of this tile had two:
// its matrix cores idled through every load-issue burst, stage and epilogue).  nf is a run-time value: the accumulators
// and fragments are sized for 8, the B fragments of a half round are read unconditionally (tiles past nf hold stale
// LDS, never multiplied), and each column tile's 8 MFMAs sit behind one wave-uniform branch.
// LDS: two buffers x (A 128 x 36 + B 128 x 36 floats) = 72 KB, two workgroups per CU; the next round's global loads
// are in flight during a round's MFMAs.
// Operand layouts, load widths, split-K, epilogues, bias-gradient row sums and the XCD-aware tile order are those of
// gemm.hip (same GemmProblem, same PanelIter); row-contiguous operands store their k rows permuted (gemm_panel.h).
#include "gemm_panel.h"

namespace gscan {

constexpr int WBM = 128, WBK = 32, WLDK = WBK + 4, WLDR = 128 + 16, WMAXF = 8, WTHREADS = 512, WWF = WMAXF / 2;
constexpr int W_FLOATS = (WBM * WLDK > WBK * WLDR) ? WBM * WLDK : WBK * WLDR;     // one operand panel image
static_assert(W_FLOATS == 4608, "both images of a 128-row panel take 4608 floats");
constexpr size_t kWideLdsBytes = 4 * (size_t)W_FLOATS * sizeof(float);
// [A buffer 0 | A buffer 1 | B buffer 0 | B buffer 1].  Always addressed as wide_lds + offset: a pointer picked from an
// array of buffer pointers loses its address space and every fragment read becomes a flat load (waits on vmcnt too).
extern __shared__ __attribute__((aligned(16))) float wide_lds[];

// One 32-deep round of a wave: 2 x NF accumulators, fragments of a half round (4 steps) in registers at a time.
template <int NF, int KCA, int KCB>
__device__ __forceinline__ void wide_round(const float *la, const float *lb, f32x4 (&acc)[2][WWF]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {           // half round: steps s = 4 h .. 4 h + 3
        float af[2][4], bf[NF][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (KCA) {
                const float4 x = *reinterpret_cast<const float4 *>(la + 16 * t * WLDK + 4 * h);
                af[t][0] = x.x; af[t][1] = x.y; af[t][2] = x.z; af[t][3] = x.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) af[t][s] = la[(4 * (4 * h + s)) * WLDR + 16 * t];
            }
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (KCB) {
                const float4 x = *reinterpret_cast<const float4 *>(lb + 16 * j * WLDK + 4 * h);
                bf[j][0] = x.x; bf[j][1] = x.y; bf[j][2] = x.z; bf[j][3] = x.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) bf[j][s] = lb[(4 * (4 * h + s)) * WLDR + 16 * j];
            }
        }
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][s], bf[j][s], acc[t][j], 0, 0, 0);
    }
}

template <int KCA, int KCB, int VWA, int VWB>
__device__ __forceinline__ void gemm_wide_tile(const GemmProblem &g, int local) {
    const int nf = g.nf, BNW = 16 * nf;
#ifdef GSCAN_GEMM_STAMPS
    unsigned gst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long gst_prev = clock64();
#endif
    const int bz = g.inv_mn ? (int)__umulhi((uint32_t)local, g.inv_mn) : local, rem = local - bz * g.tiles_mn;
    const bool n_major = (g.flags & 16) != 0;
    const int inner = g.inv_in ? (int)__umulhi((uint32_t)rem, g.inv_in) : rem;
    const int by = n_major ? rem - inner * (int)g.tiles_m : inner, bx = n_major ? inner : rem - inner * g.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = by * WBM, n0 = bx * BNW;
    const int kbeg = bz * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
    const bool do_asum = g.asum1 != nullptr && bx == 0;

    PanelIter<WBM, WBK, KCA, VWA, WTHREADS> pa;
    PanelIter<WBM, WBK, KCB, VWB, WTHREADS> pb;
    pa.init(g.a, g.sam, g.sak, g.M, m0, 1 << (g.flags & 3), kbeg, kend, tid);
    pb.init(g.b, g.sbn, g.sbk, min(g.N, n0 + BNW), n0, 1 << ((g.flags >> 2) & 3), kbeg, kend, tid);   // rows past the tile: dead loads

    f32x4 acc[2][WWF];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < WWF; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;

    const int fr = lane & 15, fg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int nf0 = (nf + 1) >> 1, j0 = wn ? nf0 : 0, nfw = wn ? nf - nf0 : nf0;      // this wave's column tiles j0 .. j0 + nfw - 1
    // fragment bases.  k-contiguous image [row][36]: the lane's k = 8 fg + s are 8 consecutive floats (two b128 reads,
    // one per half round).  Row-contiguous image [4 s + fg][144]: one b32 read per step.
    const int fa = KCA ? (32 * wm + fr) * WLDK + 8 * fg : fg * WLDR + 32 * wm + fr;
    const int fb = KCB ? (16 * j0 + fr) * WLDK + 8 * fg : fg * WLDR + 16 * j0 + fr;

    float ra[8], rb[8];
    GST(0)                                   // index math and iterator set-up
    uint32_t ma = pa.load(ra, kbeg);
    uint32_t mb = pb.load(rb, kbeg);
    GST(1)                                   // first loads issued
    auto stage = [&](int buf) {
        float *da = wide_lds + buf * W_FLOATS, *db = wide_lds + (2 + buf) * W_FLOATS;
        if (KCA) pa.template store<WLDR>(da, ra, ma, tid); else pa.template store_rows_permuted<WLDR>(da, ra, ma, tid);
        if (KCB) pb.template store<WLDR>(db, rb, mb, tid); else pb.template store_rows_permuted<WLDR>(db, rb, mb, tid);
    };
    stage(0);
    __syncthreads();
    GST(2)                                   // first panels landed and staged

    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += WBK) {
        const bool more = k0 + WBK < kend;
        if (more) {   // next round's loads fly while this round's MFMAs run
            ma = pa.load(ra, k0 + WBK);
            mb = pb.load(rb, k0 + WBK);
        }
        GST(3)
        const float *la = wide_lds + buf * W_FLOATS + fa, *lb = wide_lds + (2 + buf) * W_FLOATS + fb;
        // ONE wave-uniform dispatch per round to straight-line code for this wave's number of column tiles
        switch (nfw) {
            case 0: break;
            case 1: wide_round<1, KCA, KCB>(la, lb, acc); break;
            case 2: wide_round<2, KCA, KCB>(la, lb, acc); break;
            case 3: wide_round<3, KCA, KCB>(la, lb, acc); break;
            default: wide_round<4, KCA, KCB>(la, lb, acc); break;
        }
        if (do_asum && tid < WBM) {           // row sums of A (bias gradients) over this round's 32 k
            const float *sa = wide_lds + buf * W_FLOATS;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            if (KCA) {
#pragma unroll
                for (int kk = 0; kk < 32; kk += 4) {
                    const float4 x = *reinterpret_cast<const float4 *>(sa + tid * WLDK + kk);
                    t0 += x.x; t1 += x.y; t2 += x.z; t3 += x.w;
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 32; kk += 4) {
                    t0 += sa[kk * WLDR + tid]; t1 += sa[(kk + 1) * WLDR + tid];
                    t2 += sa[(kk + 2) * WLDR + tid]; t3 += sa[(kk + 3) * WLDR + tid];
                }
            }
            asum += (t0 + t1) + (t2 + t3);
        }
        GST(4)                               // fragment reads + MFMAs
        if (more) stage(buf ^ 1);
        GST(5)                               // wait for the next panels + LDS stores
        __syncthreads();
        GST(6)
        buf ^= 1;
    }

    if (do_asum && tid < WBM && m0 + tid < g.M) {
        atomicAdd(&g.asum1[m0 + tid], asum);
        if (g.asum2) atomicAdd(&g.asum2[m0 + tid], asum);
    }

    // epilogue.  MFMA C/D fragment: column index = lane & 15, row index = (lane >> 4) * 4 + reg.
    const uint32_t ldc = (uint32_t)g.ldc;
    const float alpha = g.alpha;
    const bool plain = g.beta == 0.f && !g.bias && g.act == 0 && !g.mask;
    gfloat *gc = as_global(g.c);
    const gfloat *gbias = as_global(g.bias), *ggate = as_global(g.gate), *gmask = as_global(g.mask);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 32 * wm + 16 * t + 4 * fg + r;
            if (row >= g.M) continue;
            const uint32_t roff = (uint32_t)row * ldc;
#pragma unroll
            for (int j = 0; j < WWF; ++j) {
                const int col = n0 + 16 * (j0 + j) + fr;
                if (j >= nfw || col >= g.N) continue;
                const uint32_t at = roff + col;
                float v = alpha * acc[t][j][r];
                if (g.atomic) { atomicAdd(g.c + at, v); continue; }
                if (!plain) {
                    if (g.beta != 0.f) v += g.beta * gc[at];
                    if (gbias) v += gbias[col];
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    else if (g.act == 2) v = tanhf_(v);
                    else if (g.act == 3 && ggate[at] == 0.f) v = 0.f;        // ReLU backward
                    if (gmask) v *= gmask[at];
                }
                gc[at] = v;
            }
        }
    GST(7)                                   // epilogue
#ifdef GSCAN_GEMM_STAMPS
    if (g_trace_buf && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_trace_buf[1500 + i] += gst_acc[i];
#endif
}

__global__ __launch_bounds__(WTHREADS, 2) void gemm_wide_kernel(int tb0, int tb1, int tb2, int tb3, int tb4, int tb5, int tb6,
                                                           int tb7, int tb8, int tb9, int tb10, int tb11, GemmGroup grp) {
    // problem lookup and XCD-aware tile order: as gemm_group_kernel (gemm.hip), preloaded header included
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
    asm volatile("" : "+s"(g.M), "+s"(g.N), "+s"(g.K), "+s"(g.alpha), "+s"(g.beta), "+s"(g.a), "+s"(g.sam), "+s"(g.sak),
                      "+s"(g.b), "+s"(g.sbk), "+s"(g.sbn), "+s"(g.c), "+s"(g.ldc));
    asm volatile("" : "+s"(g.bias), "+s"(g.act), "+s"(g.mask), "+s"(g.gate), "+s"(g.k_chunk), "+s"(g.atomic),
                      "+s"(g.asum1), "+s"(g.asum2), "+s"(g.tiles_n), "+s"(g.tiles_mn), "+s"(g.nsplit), "+s"(g.flags),
                      "+s"(g.inv_mn), "+s"(g.inv_in), "+s"(g.tiles_m), "+s"(g.nf),
                      "+s"(per));
    int local = blockIdx.x - first;
    if (per > 0) {
        const int x = local & 7, j = local >> 3;
        local = x * per + j;
        if (j >= per || local >= g.tiles_mn * g.nsplit) return;
    }
    const bool kca = g.sak == 1, kcb = g.sbk == 1;
    const int wa = g.flags & 3, wb = (g.flags >> 2) & 3;             // log2 of the load widths
    // the layouts the step's tall and long-K products use; GemmBatch::launch_wide only comes here with these
    if (kca && kcb && wa == 2 && wb == 2) gemm_wide_tile<1, 1, 4, 4>(g, local);
    else if (kca && kcb && wa == 1 && wb == 1) gemm_wide_tile<1, 1, 2, 2>(g, local);      // rows of 150 features
    else if (kca && !kcb && wa == 2 && wb == 2) gemm_wide_tile<1, 0, 4, 4>(g, local);
    else if (!kca && !kcb && wa == 2 && wb == 2) gemm_wide_tile<0, 0, 4, 4>(g, local);
    else if (!kca && !kcb && wa == 2 && wb == 1) gemm_wide_tile<0, 0, 4, 2>(g, local);
}

// 1 if the wide kernel has a tile copy for this problem's operand layouts
bool gemm_wide_supports(const GemmProblem &p) {
    const bool kca = p.sak == 1, kcb = p.sbk == 1;
    const int wa = p.flags & 3, wb = (p.flags >> 2) & 3;
    if (kca && kcb) return (wa == 2 && wb == 2) || (wa == 1 && wb == 1);
    if (kca && !kcb) return wa == 2 && wb == 2;
    if (!kca && !kcb) return wa == 2 && (wb == 2 || wb == 1);
    return false;
}

// N -> (column tiles, nf): cover ceil(N / 16) MFMA tiles with the least padding, fewer workgroup tiles on a tie
void gemm_wide_columns(int N, int *tiles_n, int *nf) {
    const int frags = cdiv(N, 16), tmin = cdiv(frags, WMAXF);
    int best_t = tmin, best_nf = cdiv(frags, tmin);
    for (int t = tmin + 1; t <= tmin + 2 && t <= frags; ++t) {
        const int f = cdiv(frags, t);
        if (t * f < best_t * best_nf) { best_t = t; best_nf = f; }
    }
    *tiles_n = best_t;
    *nf = best_nf;
}

int gemm_wide_launch(const GemmGroup &grp, int total, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_wide_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLdsBytes));
        attr_set = true;
    }
    const int *t = grp.tile_begin;
    hipLaunchKernelGGL(gemm_wide_kernel, dim3(total), dim3(WTHREADS), kWideLdsBytes, stream, t[0], t[1], t[2], t[3], t[4], t[5],
                       t[6], t[7], t[8], t[9], t[10], t[11], grp);
    GSCAN_LAUNCHED("gemm_wide_kernel");
    return 0;
}

GSCAN_TRACE_TU(gemm_wide)

}  // namespace gscan
