// Macro-tile variant of the grouped fp32 GEMM (gemm.hip): (64 | 128) x 128 x 32 workgroup tiles, one wave per SIMD.
//
// Why.  The 32 x 64 tiles of gemm.hip re-read 4 (32 + 64) bytes of operand per 2 . 32 . 64 flop of a K step: 10.7
// flop per byte.  A CU of this chip takes in ~10-12 B per cycle through its vector memory path however the lines are
// served (profiles/r04_gemm_stamps.txt: a 32 x 64 workgroup spends as long ISSUING a round's three panel loads as on
// its sixteen MFMAs once five workgroups share the CU), so at 256 flop per cycle and CU those tiles cannot keep the
// matrix cores more than ~40 % busy (measured: 38 %).  A 64 x 128 tile has 21 flop per byte, a 128 x 128 tile 32.
// What the small tiles were for — enough workgroups to fill 256 CUs — is done here by the choice between the two tile
// heights per launch and by splitting K finer (every workgroup of a launch gets about the same number of K rounds,
// GemmBatch::launch_macro_tiles), the partial tiles being added in a FIXED order afterwards instead of with atomics:
//
//   slab mode (split-K, the training step): a workgroup stores its partial tile in accumulator order — one
//   coalesced 256-byte row per register — into a slab of the caller's scratch; gemm_mt_reduce_kernel, launched behind
//   it on the same stream, adds the slabs of every 16 x 16 fragment in slice order (one wave per row of fragments,
//   eight slices' loads in flight) and applies C = beta C + alpha sum.  Bitwise reproducible, unlike the atomics.
//
// Geometry.  256 threads = 4 waves as 2 x 2; a wave owns up to WM x 4 MFMA fragments (WM = 2 or 4: 32 or 64 accumulator
// registers) and issues 8 WM . 4 v_mfma_f32_16x16x4_f32 per K round from WM + 4 operand registers per step.  Tiles at
// the edge of a matrix (N = 100: seven fragments) split their LIVE fragments evenly between the two waves of a
// dimension; the common fragment counts have straight-line code of their own, the rest skip dead fragments behind
// SCALAR branches (the wave number comes through v_readfirstlane: as a vector value every such branch was an EXEC mask
// with copies of the accumulators around it).  One LDS buffer (A | B images, 28-36 KB), two barriers per round; the next
// round's global loads fly during a round's MFMAs, the co-resident workgroups cover each other's barriers.
// Bias gradients (row sums of A) ride along as one more COLUMN of the product: column N of the B image holds ones, so
// the sums appear in the accumulators (and in the slabs) and need no atomics either.
// Operand layouts, load widths, epilogues and the XCD-aware tile order are those of gemm.hip (same GemmProblem, same
// PanelIter); row-contiguous operands store their k rows permuted (gemm_panel.h).
#include "gemm_panel.h"

namespace gscan {

constexpr int MBN = 128, MBK = 32, MLDK = MBK + 4, MTHREADS = 256, MWN = 4;
template <int WM> struct MtGeo {
    static constexpr int BM = 32 * WM;                       // 64 or 128 rows
    static constexpr int LDR_A = BM + 16;                    // row-contiguous image [k][LDR]: 16 mod 32, two lanes per bank
    static constexpr int A_FLOATS = (BM * MLDK > MBK * LDR_A) ? BM * MLDK : MBK * LDR_A;
};
constexpr int MLDR_B = MBN + 16;
constexpr int MB_FLOATS = (MBN * MLDK > MBK * MLDR_B) ? MBN * MLDK : MBK * MLDR_B;
constexpr int kSlabFloats = 4 * 4 * MWN * 4 * 64;            // [wave][i < 4][j < 4][reg][lane]: one partial tile, 64 KB

// One 32-deep round of a wave: MF x NF accumulators, fragments of a half round (4 steps) in registers at a time.
// Step s of half h: lane group g holds k = 8 g + 4 h + s (k-contiguous image: one b128 read per fragment and half;
// row-contiguous image, rows permuted to 4 (4 h + s) + g: one b32 read per fragment and step).
// halves = 1: the slice's last round holds at most four live k (K = 100: k = 96..99), all of them in half 0.
template <int MF, int NF, int WM, int KCA, int KCB>
__device__ __forceinline__ void mt_round(const float *la, const float *lb, f32x4 (&acc)[WM][MWN], int halves) {
    constexpr int LDR_A = MtGeo<WM>::LDR_A;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h >= halves) break;
        float af[MF][4], bf[NF][4];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            if (KCA) {
                const float4 x = *reinterpret_cast<const float4 *>(la + 16 * i * MLDK + 4 * h);
                af[i][0] = x.x; af[i][1] = x.y; af[i][2] = x.z; af[i][3] = x.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) af[i][s] = la[(4 * (4 * h + s)) * LDR_A + 16 * i];
            }
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (KCB) {
                const float4 x = *reinterpret_cast<const float4 *>(lb + 16 * j * MLDK + 4 * h);
                bf[j][0] = x.x; bf[j][1] = x.y; bf[j][2] = x.z; bf[j][3] = x.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) bf[j][s] = lb[(4 * (4 * h + s)) * MLDR_B + 16 * j];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
}

// any fragment counts (edge tiles): dead fragments skipped behind scalar branches
template <int WM, int KCA, int KCB>
__device__ __forceinline__ void mt_round_any(const float *la, const float *lb, int mfw, int nfw, f32x4 (&acc)[WM][MWN],
                                             int halves) {
    constexpr int LDR_A = MtGeo<WM>::LDR_A;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h >= halves) break;
#pragma unroll
        for (int j = 0; j < MWN; ++j) {
            if (j >= nfw) break;
            float bf[4];
            if (KCB) {
                const float4 x = *reinterpret_cast<const float4 *>(lb + 16 * j * MLDK + 4 * h);
                bf[0] = x.x; bf[1] = x.y; bf[2] = x.z; bf[3] = x.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) bf[s] = lb[(4 * (4 * h + s)) * MLDR_B + 16 * j];
            }
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                if (i >= mfw) break;
                float af[4];
                if (KCA) {
                    const float4 x = *reinterpret_cast<const float4 *>(la + 16 * i * MLDK + 4 * h);
                    af[0] = x.x; af[1] = x.y; af[2] = x.z; af[3] = x.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[s] = la[(4 * (4 * h + s)) * LDR_A + 16 * i];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc[i][j], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ int frags_of(int n) { return (n + 15) >> 4; }
// Where a tile's live fragments go: the first ceil(live / 2) to wave 0 of the dimension, the rest to wave 1.
__device__ __forceinline__ void mt_share(int live, int w, int *first, int *count) {
    const int h0 = (live + 1) >> 1;
    *first = w ? h0 : 0;
    *count = w ? live - h0 : h0;
}

__shared__ __attribute__((aligned(16))) float mt_lds[MtGeo<4>::A_FLOATS + MB_FLOATS];     // [A image | B image]

// what a workgroup knows about its tile (scalars), shared by the K loop and the epilogue
struct MtTile { int m0, n0, kbeg, kend, rem, bz, NV, fm0, mfw, fn0, nfw, wave; };

// The K loop of one tile: panels global -> registers -> LDS, fragments LDS -> registers, MFMAs.  ANY = run-time load
// widths and fragment counts only (the rare small products: rows of 9 or 25 floats).
template <int WM, int KCA, int KCB, int VWA, int VWB>
__device__ __forceinline__ void mt_k_loop(const GemmProblem &g, const MtTile &t, f32x4 (&acc)[WM][MWN]) {
    constexpr int BM = MtGeo<WM>::BM, LDR_A = MtGeo<WM>::LDR_A;
    constexpr bool ANY = VWA == 0;
#ifdef GSCAN_GEMM_STAMPS
    unsigned gst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long gst_prev = clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int BNW = 16 * g.nf;
    const int ones_col = g.asum1 ? g.N - t.n0 : -1;    // the virtual column of ones: inside this tile when 0 <= ones_col < BNW
    const bool has_ones = ones_col >= 0 && ones_col < BNW;
    const int kbeg = t.kbeg, kend = t.kend;

    PanelIter<BM, MBK, KCA, VWA, MTHREADS> pa;
    PanelIter<MBN, MBK, KCB, VWB, MTHREADS> pb;
    pa.init(g.a, g.sam, g.sak, g.M, t.m0, 1 << (g.flags & 3), kbeg, kend, tid);
    pb.init(g.b, g.sbn, g.sbk, min(g.N, t.n0 + BNW), t.n0, 1 << ((g.flags >> 2) & 3), kbeg, kend, tid);   // rows past the tile: dead loads

    const int fr = lane & 15, fg = lane >> 4;
    const int mfw = t.mfw, nfw = t.nfw;
    const int fa = KCA ? (16 * t.fm0 + fr) * MLDK + 8 * fg : fg * LDR_A + 16 * t.fm0 + fr;
    const int fb = KCB ? (16 * t.fn0 + fr) * MLDK + 8 * fg : fg * MLDR_B + 16 * t.fn0 + fr;

    float ra[BM * MBK / MTHREADS], rb[MBN * MBK / MTHREADS];
    GST(0)                                   // index math and iterator set-up
    uint32_t ma = pa.load(ra, kbeg);
    uint32_t mb = pb.load(rb, kbeg);
    GST(1)                                   // first loads issued
    float *const da = mt_lds, *const db = mt_lds + MtGeo<WM>::A_FLOATS;
    for (int k0 = kbeg; k0 < kend; k0 += MBK) {
        if (KCA) pa.template store<LDR_A>(da, ra, ma, tid); else pa.template store_rows_permuted<LDR_A>(da, ra, ma, tid);
        if (KCB) pb.template store<MLDR_B>(db, rb, mb, tid); else pb.template store_rows_permuted<MLDR_B>(db, rb, mb, tid);
        if (has_ones) {                       // workgroup-uniform; column `ones_col` of B = 1 on this round's live k
            __syncthreads();                  // its owners' zeros are in
            if (tid < MBK) {
                const float one = k0 + tid < kend ? 1.f : 0.f;
                if (KCB) db[ones_col * MLDK + tid] = one;
                else db[(4 * (tid & 7) + (tid >> 3)) * MLDR_B + ones_col] = one;
            }
        }
        GST(2)                                // panels landed (wait) and staged
        __syncthreads();
        GST(3)
        if (k0 + MBK < kend) {                // next round's loads fly while this round's MFMAs run
            ma = pa.load(ra, k0 + MBK);
            mb = pb.load(rb, k0 + MBK);
        }
        GST(4)                                // next loads issued
        const int halves = kend - k0 <= 4 ? 1 : 2;
        const float *la = da + fa, *lb = db + fb;
        // ONE wave-uniform dispatch per round to straight-line code for this wave's fragment counts
        if (!ANY && mfw == WM && nfw == 4) mt_round<WM, 4, WM, KCA, KCB>(la, lb, acc, halves);
        else if (!ANY && mfw == WM && nfw == 3) mt_round<WM, 3, WM, KCA, KCB>(la, lb, acc, halves);
        else if (!ANY && mfw == WM - 1 && nfw == 4) mt_round<WM - 1, 4, WM, KCA, KCB>(la, lb, acc, halves);
        else if (!ANY && mfw == WM - 1 && nfw == 3) mt_round<WM - 1, 3, WM, KCA, KCB>(la, lb, acc, halves);
        else mt_round_any<WM, KCA, KCB>(la, lb, mfw, nfw, acc, halves);
        GST(5)                                // fragment reads + MFMAs
        __syncthreads();
        GST(6)
    }
#ifdef GSCAN_GEMM_STAMPS
    if (g_trace_buf && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_trace_buf[1500 + i] += gst_acc[i];
#endif
}

template <int WM>
__global__ __launch_bounds__(MTHREADS, WM == 4 ? 2 : 3) void gemm_mt_kernel(int tb0, int tb1, int tb2, int tb3, int tb4, int tb5,
                                                                           int tb6, int tb7, int tb8, int tb9, int tb10,
                                                                           int tb11, GemmGroup grp) {
    // problem lookup and XCD-aware tile order: as gemm_group_kernel (gemm.hip), preloaded header included.  The
    // descriptor is read from the kernel arguments where it is needed (a workgroup lives for thousands of cycles:
    // keeping all of it in SGPRs from the start, as the small-tile kernel does, only makes the compiler spill them)
    const int tb[kMaxGroup] = {tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7, tb8, tb9, tb10, tb11};
    static_assert(kMaxGroup == 12, "the preloaded header is twelve scalars");
    constexpr int BM = MtGeo<WM>::BM;
    TraceScope trace_scope(TK_GEMM);
    int pi = 0, first = tb[0];
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if ((int)blockIdx.x >= tb[i]) { pi = i; first = tb[i]; }
    asm volatile("" : "+s"(pi), "+s"(first));
    const int per = grp.xcd_per[pi];
    const GemmProblem &g = grp.p[pi];
    int local = blockIdx.x - first;
    if (per > 0) {
        const int x = local & 7, j = local >> 3;
        local = x * per + j;
        if (j >= per || local >= g.tiles_mn * g.nsplit) return;
    }
    MtTile t;
    t.bz = g.inv_mn ? (int)__umulhi((uint32_t)local, g.inv_mn) : local;
    t.rem = local - t.bz * g.tiles_mn;
    const bool n_major = (g.flags & 16) != 0;
    const int inner = g.inv_in ? (int)__umulhi((uint32_t)t.rem, g.inv_in) : t.rem;
    const int by = n_major ? t.rem - inner * (int)g.tiles_m : inner, bx = n_major ? inner : t.rem - inner * g.tiles_n;
    // the wave number as a SCALAR: everything derived from it (this wave's fragment counts, the branches around dead
    // fragments) must be wave-uniform to the compiler
    t.wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    t.m0 = by * BM; t.n0 = bx * 16 * g.nf;
    t.kbeg = t.bz * g.k_chunk; t.kend = min(g.K, t.kbeg + g.k_chunk);
    t.NV = g.N + (g.asum1 ? 1 : 0);                   // the virtual column of ones (row sums of A) is column N
    mt_share(min(2 * WM, frags_of(g.M - t.m0)), t.wave >> 1, &t.fm0, &t.mfw);
    mt_share(min(g.nf, frags_of(t.NV - t.n0)), t.wave & 1, &t.fn0, &t.nfw);

    f32x4 acc[WM][MWN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < MWN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool kca = g.sak == 1, kcb = g.sbk == 1;
    const int wa = g.flags & 3, wb = (g.flags >> 2) & 3;             // log2 of the load widths
    // straight-line load code for the layouts of the step's large products, run-time load widths for the rest
    if (kca && kcb && wa == 2 && wb == 2) mt_k_loop<WM, 1, 1, 4, 4>(g, t, acc);
    else if (kca && kcb && wa == 1 && wb == 1) mt_k_loop<WM, 1, 1, 2, 2>(g, t, acc);      // rows of 150 features
    else if (kca && !kcb && wa == 2 && wb == 2) mt_k_loop<WM, 1, 0, 4, 4>(g, t, acc);
    else if (!kca && !kcb && wa == 2 && wb == 2) mt_k_loop<WM, 0, 0, 4, 4>(g, t, acc);
    else if (!kca && !kcb && wa == 2 && wb == 1) mt_k_loop<WM, 0, 0, 4, 2>(g, t, acc);
    else if (kca && kcb) mt_k_loop<WM, 1, 1, 0, 0>(g, t, acc);
    else if (kca) mt_k_loop<WM, 1, 0, 0, 0>(g, t, acc);
    else if (kcb) mt_k_loop<WM, 0, 1, 0, 0>(g, t, acc);
    else mt_k_loop<WM, 0, 0, 0, 0>(g, t, acc);

    // ---- epilogue.  MFMA C/D fragment: column index = lane & 15, row index = (lane >> 4) * 4 + reg.
    const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const int mode = g.atomic;
    if (mode == 2) {              // slab: accumulator order, one coalesced 256-byte row per register
        gfloat *slab = as_global(g.slab) + ((size_t)(t.rem * g.nsplit + t.bz) * 4 + t.wave) * (4 * MWN * 4 * 64) + lane;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < MWN; ++j)
                if (i < t.mfw && j < t.nfw) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[((i * MWN + j) * 4 + r) * 64] = acc[i][j][r];
                }
        return;
    }
    const uint32_t ldc = (uint32_t)g.ldc;
    const int M = g.M, N = g.N;
    const float alpha = g.alpha, beta = g.beta;
    const int act = g.act;
    float *const asum1 = g.asum1, *const asum2 = g.asum2;
    float *const c = g.c;
    gfloat *gc = as_global(c);
    const gfloat *gbias = as_global(g.bias), *ggate = as_global(g.gate), *gmask = as_global(g.mask);
    const bool plain = beta == 0.f && !gbias && act == 0 && !gmask;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = t.m0 + 16 * (t.fm0 + i) + 4 * fg + r;
            if (i >= t.mfw || row >= M) continue;
            const uint32_t roff = (uint32_t)row * ldc;
#pragma unroll
            for (int j = 0; j < MWN; ++j) {
                const int col = t.n0 + 16 * (t.fn0 + j) + fr;
                if (j >= t.nfw || col >= t.NV) continue;
                if (col == N) {                        // the ones column: row sums of A (bias gradients), NOT scaled by
                    const float sa = acc[i][j][r];     // alpha: asum += sum_k A(m,k), as the small-tile kernel and the header say
                    if (mode) { atomicAdd(asum1 + row, sa); if (asum2) atomicAdd(asum2 + row, sa); }
                    else { asum1[row] += sa; if (asum2) asum2[row] += sa; }
                    continue;
                }
                float v = alpha * acc[i][j][r];
                const uint32_t at = roff + col;
                if (mode) { atomicAdd(c + at, v); continue; }
                if (!plain) {
                    if (beta != 0.f) v += beta * gc[at];
                    if (gbias) v += gbias[col];
                    if (act == 1) v = fmaxf(v, 0.f);
                    else if (act == 2) v = tanhf_(v);
                    else if (act == 3 && ggate[at] == 0.f) v = 0.f;          // ReLU backward
                    if (gmask) v *= gmask[at];
                }
                gc[at] = v;
            }
        }
}

// The second pass of slab mode.  One WAVE per live 16 x 16 fragment: the lane adds its four registers of every slice
// in slice order (sixteen slices' loads in flight: the pass is a handful of dependent round trips to L2 / the
// memory-side cache whatever the slice count), then C = beta C + alpha sum, and the ones column goes to the bias
// gradients.  The unit list is the GEMM's own index arithmetic run backwards: problem -> tile -> wave -> (i, j).
__global__ __launch_bounds__(256) void gemm_mt_reduce_kernel(int tb0, int tb1, int tb2, int tb3, int tb4, int tb5, int tb6,
                                                            int tb7, int tb8, int tb9, int tb10, int tb11, int bm,
                                                            GemmGroup grp) {
    const int tb[kMaxGroup] = {tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7, tb8, tb9, tb10, tb11};
    TraceScope trace_scope(TK_GEMM);
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int pi = 0, first = tb[0];
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if (unit >= tb[i]) { pi = i; first = tb[i]; }
    if (pi >= grp.count) return;
    const GemmProblem &g = grp.p[pi];
    if (g.atomic != 2) return;
    const int u = unit - first;                                     // = ((tile * 4 + wave) * 4 + i) * 4 + j
    if (u >= g.tiles_mn * 64) return;
    const int j = u & 3, i = (u >> 2) & 3, wave = (u >> 4) & 3, rem = u >> 6;
    const bool n_major = (g.flags & 16) != 0;
    const int inner = n_major ? rem / g.tiles_m : rem / g.tiles_n;
    const int by = n_major ? rem - inner * g.tiles_m : inner, bx = n_major ? inner : rem - inner * g.tiles_n;
    const int m0 = by * bm, n0 = bx * 16 * g.nf;
    const int NV = g.N + (g.asum1 ? 1 : 0);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fg = lane >> 4;
    int fm0, mfw, fn0, nfw;
    mt_share(min(bm / 16, frags_of(g.M - m0)), wm, &fm0, &mfw);
    mt_share(min(g.nf, frags_of(NV - n0)), wn, &fn0, &nfw);
    if (i >= mfw || j >= nfw) return;
    constexpr size_t slice = (size_t)4 * 4 * MWN * 4 * 64;
    const gfloat *p = as_global(g.slab) + ((size_t)rem * g.nsplit * 4 + wave) * (4 * MWN * 4 * 64) +
                      (size_t)((i * MWN + j) * 4) * 64 + lane;
    const int ns = g.nsplit;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 16 <= ns; z += 16) {             // 64 independent loads, then the adds in slice order
        float x[16][4];
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[q][r] = p[(size_t)(z + q) * slice + r * 64];
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] += x[q][r];
    }
    {                                           // the last (up to fifteen) slices: loads clamped to a live slice, adds masked
        float x[16][4];
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[q][r] = p[(size_t)min(z + q, ns - 1) * slice + r * 64];
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] += z + q < ns ? x[q][r] : 0.f;
    }
    const int col = n0 + 16 * (fn0 + j) + fr;
    if (col >= NV) return;
    gfloat *gc = as_global(g.c);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * (fm0 + i) + 4 * fg + r;
        if (row >= g.M) continue;
        if (col == g.N) { g.asum1[row] += s[r]; if (g.asum2) g.asum2[row] += s[r]; continue; }   // row sums of A: unscaled
        const float v = g.alpha * s[r];
        const uint32_t at = (uint32_t)row * (uint32_t)g.ldc + col;
        gc[at] = g.beta != 0.f ? g.beta * gc[at] + v : v;
    }
}

// every operand layout has a tile copy (the common ones with compile-time load widths)
bool gemm_mt_supports(const GemmProblem &p) { return p.M > 0; }

// N (the ones column included) -> (column tiles, fragments per tile): tiles of up to eight fragments, cut evenly
// (25 fragments: 7 + 7 + 7 + 4, not 8 + 8 + 8 + 1)
void gemm_mt_columns(int NV, int *tiles_n, int *nf) {
    const int frags = cdiv(NV, 16), t = cdiv(frags, 2 * MWN);
    *nf = cdiv(frags, t);
    *tiles_n = cdiv(frags, *nf);
}

size_t gemm_mt_slab_floats() { return kSlabFloats; }
size_t gemm_slab_floats() { return kSlabFloats; }

int gemm_mt_launch(const GemmGroup &grp, int total, int bm, hipStream_t stream) {
    const int *t = grp.tile_begin;
    if (bm == 128)
        hipLaunchKernelGGL(gemm_mt_kernel<4>, dim3(total), dim3(MTHREADS), 0, stream, t[0], t[1], t[2], t[3], t[4], t[5], t[6],
                           t[7], t[8], t[9], t[10], t[11], grp);
    else
        hipLaunchKernelGGL(gemm_mt_kernel<2>, dim3(total), dim3(MTHREADS), 0, stream, t[0], t[1], t[2], t[3], t[4], t[5], t[6],
                           t[7], t[8], t[9], t[10], t[11], grp);
    GSCAN_LAUNCHED("gemm_mt_kernel");
    return 0;
}

// reduce pass: unit_begin = prefix sums of tiles_mn * 64 over the split problems (the others take no units)
int gemm_mt_reduce_launch(const GemmGroup &grp, const int (&unit_begin)[kMaxGroup], int total_units, int bm,
                          hipStream_t stream) {
    const int *t = unit_begin;
    hipLaunchKernelGGL(gemm_mt_reduce_kernel, dim3(cdiv(total_units, 4)), dim3(256), 0, stream, t[0], t[1], t[2], t[3], t[4],
                       t[5], t[6], t[7], t[8], t[9], t[10], t[11], bm, grp);
    GSCAN_LAUNCHED("gemm_mt_reduce_kernel");
    return 0;
}

GSCAN_TRACE_TU(gemm_mt)

}  // namespace gscan
