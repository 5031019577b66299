// Weights-stationary persistent GEMM for the tall-skinny products of the forward pass (round 5).
//
//   C[M,N] = act(A . B^T + bias)     A [M,K] activations (k-contiguous rows), B [N,K] a weight matrix in the reference's
//                                    own [out,in] layout, K <= 160, N a few hundred, M thousands of rows
//
// These are the prelude's products (seq2seq_model.py:466-469 and the gate images / embedded gates of DESIGN.md 4.1):
// projected keys 9216 x 100 x 150 and 2560 x 100 x 100, gate images 9216 x 400 x 150 and 2560 x 400 x 100, the embedded
// target tokens' part of the gates 5120 x 400 x 100, the bridge.  On the 32 x 64 tiles of gemm.hip every workgroup
// re-reads a B panel that would fit LDS once, pays a prologue and an epilogue for four or five K rounds, and five
// workgroups per CU burst their loads at the same barrier (a tile's life is 22 k cycles of which 4.4 k are
// MFMAs).  Here instead:
//   * ONE workgroup per CU lives for the whole launch (grid = number of CUs).  The host deals (product, column block)
//     pairs out to the workgroups in proportion to their multiply-adds; a workgroup serves one pair and a contiguous
//     range of its 64-row tiles.
//   * B-stationary: the pair's B block (up to 112 columns x K) is copied to LDS ONCE (zero-padded to whole MFMA
//     fragments and K blocks), then A streams through: 64-row tiles, double-buffered in LDS, the next tile's loads
//     (16-byte, into registers) issued before the current tile's MFMAs and written to the other buffer behind them.
//   * one wave = 16 rows x all of the block's columns (up to seven 16 x 16 fragments); the A and B fragments of the next
//     K block are read from LDS while the current block's MFMAs run.
//   * every lane ALWAYS stores its C values — lanes outside the matrix to a dump word —, so the epilogue is straight-line
//     code and the wait for the next tile's loads leaves exactly those stores in flight (vmcnt counts in order).
// What it takes: both operands k-contiguous, beta = 0, no mask, no split, K even and <= 160, A rows either contiguous
// (lda = K) or 16-byte aligned; anything else stays on gemm.hip (GemmBatch::launch decides).
#include "gemm_panel.h"

namespace gscan {

constexpr int WS_THREADS = 256, WS_BM = 64, WS_MAXNF = 7, WS_MAXK = 160, WS_MAXPAIRS = 48;

struct WsProblem {
    const float *a, *b, *bias;
    float *c;
    int lda, ldb, ldc, M, N, K, act;
    int vr;             // floats per LDS fragment read: 4 (K % 4 == 0) or 2
    int kpb;            // row stride of the B block in LDS (zero-padded K, conflict-free for vr-wide reads)
    int flat;           // A rows are contiguous (lda == K): a tile is one linear block
};
struct WsPair { int prob, n0, ncols, nf, wg_begin, wg_count, mtiles; };
struct WsArgs { int npairs; WsPair pair[WS_MAXPAIRS]; WsProblem prob[kMaxGroup]; };

// where the lanes outside a matrix put their (unconditional) stores: a row per workgroup (every workgroup storing to the
// same few lines serialises in the L2)
constexpr int WS_DUMP_ROWS = 1024;
static __device__ float g_ws_dump[WS_DUMP_ROWS][WS_THREADS];

__host__ __device__ inline int ws_a_floats(int K) { return WS_BM * K + 32; }      // a tile + the overrun of its last row's last block

// One 64-row tile of A, global -> registers -> LDS in 16-byte units (unit u of the tile = LDS floats 4u..4u+3: rows packed
// at stride K).  Loads are unconditional (a dead one re-reads unit 0) so that their count is the same in every thread.
// (Round 5 first copied tiles straight to LDS, global_load_lds_dwordx4: a CU accepts one such wave instruction every
// ~50 cycles — 20 B per cycle — and the C stores queue behind them: 1.6 k cycles to issue a tile's ten copies.)
constexpr int WS_NL = (WS_BM * WS_MAXK / 4 + WS_THREADS - 1) / WS_THREADS;     // loads per thread of the largest tile
struct WsTile { f32x4 v[WS_NL]; };
__device__ __forceinline__ void ws_load_tile(const WsProblem &p, int m0, WsTile &r, int tid) {
    const int rows = min(WS_BM, p.M - m0);
    const int q = p.K / 4;                                  // units per row (strided form; K % 4 == 0 there)
    const int units = p.flat ? (rows * p.K) / 4 : rows * q;
    const gfloat *base = as_global(p.a) + (int64_t)m0 * p.lda;
#pragma unroll
    for (int i = 0; i < WS_NL; ++i) {
        const int u = i * WS_THREADS + tid, uc = u < units ? u : 0;
        const gfloat *src;
        if (p.flat) src = base + 4 * (int64_t)uc;
        else { const int rr = uc / q, c4 = uc - rr * q; src = base + (int64_t)rr * p.lda + 4 * c4; }
        if (i * WS_THREADS < WS_BM * p.K / 4) r.v[i] = *reinterpret_cast<const GSCAN_GLOBAL f32x4 *>(src);   // uniform: the tile's unit count
    }
}
__device__ __forceinline__ void ws_store_tile(const WsProblem &p, float *buf, const WsTile &r, int tid) {
#pragma unroll
    for (int i = 0; i < WS_NL; ++i) {
        const int u = i * WS_THREADS + tid;
        if (u < WS_BM * p.K / 4) *reinterpret_cast<f32x4 *>(buf + 4 * u) = r.v[i];
    }
}

template <int VR>
__device__ __forceinline__ void ws_read(float (&x)[VR], const float *p) {
    if constexpr (VR == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else {
        const float2 v = *reinterpret_cast<const float2 *>(p);
        x[0] = v.x; x[1] = v.y;
    }
}

// One pair: B block -> LDS, then the workgroup's tiles [t_begin, t_end) of A.  NF = 16-column fragments of the block: a
// template parameter, so that a K block is straight-line code — 1 + NF fragment reads, VR . NF MFMAs — (as a run-time
// count every read and every MFMA sat behind a branch of its own: 16 k cycles per tile against 6 k of MFMA time).
template <int VR, int NF>
__device__ __forceinline__ void ws_pair(const WsProblem &p, const WsPair &pr, int t_begin, int t_end, float *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = p.K, kpb = p.kpb;
#ifdef GSCAN_GEMM_STAMPS
    unsigned gst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long gst_prev = clock64();
#endif
    float *bp = lds, *abuf0 = lds + 16 * WS_MAXNF * kpb, *abuf1 = abuf0 + ws_a_floats(K);
    WsTile tile;
    ws_load_tile(p, t_begin * WS_BM, tile, tid);             // the first tile's loads go out before anything else
    for (int i = tid; i < 32; i += WS_THREADS) { abuf0[WS_BM * K + i] = 0.f; abuf1[WS_BM * K + i] = 0.f; }   // overrun words: finite
    {   // B block [16 NF][kpb]: rows past the block's columns and columns past K are zero.  Twenty-four loads in flight per
        // thread: the copy is one or two round trips to L2 (eight at a time it was six: 14.5 k cycles, two tiles' MFMAs).
        // Thread tid takes the VR-wide groups tid, tid + 256, ...: (row, group of the row) advance by a fixed step.
        constexpr int UB = 24;
        const gfloat *gb = as_global(p.b) + (int64_t)pr.n0 * p.ldb;
        const int kq = kpb / VR, total = 16 * NF * kq;       // VR-wide groups per row, groups of the block
        const int step_n = WS_THREADS / kq, step_k = WS_THREADS - step_n * kq;
        int n = tid / kq, kg = tid - n * kq;
        for (int e0 = tid; e0 < total; e0 += WS_THREADS * UB) {
            float x[UB][VR];
            int at[UB];
            bool live[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k = VR * kg;
                live[u] = e0 + u * WS_THREADS < total && n < pr.ncols && k < K;   // K is a multiple of VR: whole groups are live
                at[u] = e0 + u * WS_THREADS < total ? n * kpb + k : -1;
                const gfloat *src = gb + (live[u] ? (int64_t)n * p.ldb + k : 0);
                if constexpr (VR == 4) {
                    const f32x4 v = *reinterpret_cast<const GSCAN_GLOBAL f32x4 *>(src);
                    x[u][0] = v[0]; x[u][1] = v[1]; x[u][2] = v[2]; x[u][3] = v[3];
                } else {
                    const f32x2 v = *reinterpret_cast<const GSCAN_GLOBAL f32x2 *>(src);
                    x[u][0] = v[0]; x[u][1] = v[1];
                }
                kg += step_k; n += step_n;
                if (kg >= kq) { kg -= kq; ++n; }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (at[u] < 0) continue;
                if constexpr (VR == 4)
                    *reinterpret_cast<float4 *>(bp + at[u]) = live[u] ? float4{x[u][0], x[u][1], x[u][2], x[u][3]} : float4{0.f, 0.f, 0.f, 0.f};
                else *reinterpret_cast<float2 *>(bp + at[u]) = live[u] ? float2{x[u][0], x[u][1]} : float2{0.f, 0.f};
            }
        }
    }
    ws_store_tile(p, abuf0, tile, tid);
    const int fi = lane & 15, fg = lane >> 4;                // MFMA 16x16x4: A[i = fi][k], B[k][j = fi]; lane group fg
    const int nblocks = (K + 4 * VR - 1) / (4 * VR);         // K blocks of 4 VR: group fg holds k = 4 VR b + VR fg + s, s < VR
    gfloat *gc = as_global(p.c), *dump = as_global(&g_ws_dump[blockIdx.x % WS_DUMP_ROWS][0]) + tid;
    float bias[NF];                                          // of this lane's column of every fragment
#pragma unroll
    for (int f = 0; f < NF; ++f)
        bias[f] = (p.bias && 16 * f + fi < pr.ncols) ? as_global(p.bias)[pr.n0 + 16 * f + fi] : 0.f;
    // the last K block reaches past K: those products are zeroed on the A side too (what lies behind a row in LDS is
    // the next row)
    const int k_last = 4 * VR * (nblocks - 1) + VR * fg;
    GST(0)                                                   // B block and first tile stored
    lds_barrier();
    for (int t = t_begin; t < t_end; ++t) {
        float *abuf = ((t - t_begin) & 1) ? abuf1 : abuf0;
        const bool more = t + 1 < t_end;
        if (more) ws_load_tile(p, (t + 1) * WS_BM, tile, tid);              // in flight during this tile's MFMAs
        GST(3)
        f32x4 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *ar = abuf + (16 * wave + fi) * K + VR * fg, *br = bp + fi * kpb + VR * fg;
        // two register sets of fragments, the K loop unrolled by two: block b + 1 is read while block b multiplies
        float a0[VR], a1[VR], b0[NF][VR], b1[NF][VR];
        auto frags = [&](float (&a)[VR], float (&b)[NF][VR], int blk) {
            ws_read<VR>(a, ar + 4 * VR * blk);
            if (blk + 1 == nblocks) {
#pragma unroll
                for (int s = 0; s < VR; ++s) a[s] = k_last + s < K ? a[s] : 0.f;
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) ws_read<VR>(b[f], br + 16 * f * kpb + 4 * VR * blk);
        };
        auto mfmas = [&](const float (&a)[VR], const float (&b)[NF][VR]) {
#pragma unroll
            for (int s = 0; s < VR; ++s)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[f][s], acc[f], 0, 0, 0);
        };
        frags(a0, b0, 0);
        for (int blk = 0; blk < nblocks; blk += 2) {
            if (blk + 1 < nblocks) frags(a1, b1, blk + 1);
            mfmas(a0, b0);
            if (blk + 2 < nblocks) frags(a0, b0, blk + 2);
            if (blk + 1 < nblocks) mfmas(a1, b1);
        }
        GST(4)                                               // fragment reads + MFMAs
        // C fragment: row = 4 (lane >> 4) + r, column = lane & 15.  EVERY lane stores, 4 NF instructions per thread (lanes
        // outside the matrix to a dump word): the count of memory operations behind the next tile's loads is exact
        const int m0 = t * WS_BM + 16 * wave + 4 * fg;
        gfloat *rowp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rowp[r] = m0 + r < p.M ? gc + (int64_t)(m0 + r) * p.ldc + pr.n0 + fi : nullptr;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const bool col_ok = 16 * f + fi < pr.ncols;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[f][r] + bias[f];
                if (p.act == 2) v = tanhf_(v);
                gfloat *at = (col_ok && rowp[r]) ? rowp[r] + 16 * f : dump;
                *at = v;
            }
        }
        GST(5)                                               // C stores issued
        if (more) ws_store_tile(p, ((t - t_begin) & 1) ? abuf0 : abuf1, tile, tid);   // (waits for its loads: the stores above stay in flight)
        GST(1)
        lds_barrier();
        GST(2)
#ifdef GSCAN_GEMM_STAMPS
        gst_acc[6] += 1;
#endif
    }
#ifdef GSCAN_GEMM_STAMPS
    if (g_trace_buf && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_trace_buf[1500 + i] += gst_acc[i];
#endif
}

__global__ __launch_bounds__(WS_THREADS) void gemm_ws_kernel(WsArgs args) {
    TraceScope trace_scope(TK_GEMM);
    extern __shared__ __attribute__((aligned(16))) float ws_lds[];
    int pi = 0;
    for (int i = 1; i < args.npairs; ++i)
        if ((int)blockIdx.x >= args.pair[i].wg_begin) pi = i;
    const WsPair pr = args.pair[pi];
    if ((int)blockIdx.x >= pr.wg_begin + pr.wg_count) return;
    const WsProblem p = args.prob[pr.prob];
    const int local = blockIdx.x - pr.wg_begin;
    const int t_begin = (int)((int64_t)pr.mtiles * local / pr.wg_count), t_end = (int)((int64_t)pr.mtiles * (local + 1) / pr.wg_count);
    if (t_begin >= t_end) return;
    switch (2 * pr.nf + (p.vr == 4 ? 1 : 0)) {
#define WS_CASE(NF) case 2 * NF: ws_pair<2, NF>(p, pr, t_begin, t_end, ws_lds); break; case 2 * NF + 1: ws_pair<4, NF>(p, pr, t_begin, t_end, ws_lds); break;
        WS_CASE(1) WS_CASE(2) WS_CASE(3) WS_CASE(4) WS_CASE(5) WS_CASE(6) WS_CASE(7)
#undef WS_CASE
        default: break;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host: eligibility, the deal of pairs to workgroups, launch
// ------------------------------------------------------------------------------------------------------------------
bool gemm_ws_eligible(const GemmProblem &g) {
    if (g.sak != 1 || g.sbk != 1 || g.beta != 0.f || g.alpha != 1.f || g.mask || g.gate || g.asum1 || g.asum2) return false;
    if (g.act != 0 && g.act != 2) return false;
    if (g.K > WS_MAXK || (g.K & 1) || g.k_chunk < g.K || g.atomic) return false;
    const int vr = (g.K % 4 == 0) ? 4 : 2;
    auto aligned = [](const void *ptr, int bytes) { return (reinterpret_cast<uintptr_t>(ptr) & (uintptr_t)(bytes - 1)) == 0; };
    // A: a tile goes to LDS in 16-byte units: contiguous rows (the tile is one linear block) or 16-byte aligned rows
    const bool flat = g.sam == g.K;
    if (!aligned(g.a, 16)) return false;
    if (flat ? (((int64_t)g.M * g.K) % 4 != 0 || ((int64_t)WS_BM * g.K) % 4 != 0) : (g.K % 4 != 0 || g.sam % 4 != 0)) return false;
    // B: vr-wide loads of its rows
    if (!aligned(g.b, 4 * vr) || g.sbn % vr != 0) return false;
    if (g.sam >= (1ll << 31) || g.sbn >= (1ll << 31) || g.ldc >= (1ll << 31)) return false;
    return true;
}

static int ws_kpb(int K, int vr) { return vr == 4 ? (K + 15) / 16 * 16 + 4 : (K + 7) / 8 * 8 + 2; }

int gemm_ws_launch(const GemmGroup &grp, hipStream_t stream) {
    static const int cus = [] {
#ifndef GSCAN_PLAN_ONLY
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) return n;
#endif
        return 256;
    }();
    WsArgs args{};
    double mac[WS_MAXPAIRS], total = 0.0;
    size_t lds_floats = 0;
    for (int i = 0; i < grp.count; ++i) {
        const GemmProblem &g = grp.p[i];
        WsProblem &p = args.prob[i];
        const int vr = (g.K % 4 == 0) ? 4 : 2;
        p = WsProblem{g.a, g.b, g.bias, g.c, (int)g.sam, (int)g.sbn, (int)g.ldc, g.M, g.N, g.K, g.act, vr, ws_kpb(g.K, vr),
                      g.sam == g.K ? 1 : 0};
        lds_floats = std::max(lds_floats, (size_t)16 * WS_MAXNF * p.kpb + 2 * (size_t)ws_a_floats(g.K));
        // column blocks: everything at once up to seven fragments, else blocks of five (80 columns) and the rest
        for (int n0 = 0; n0 < g.N;) {
            const int left = g.N - n0, ncols = left <= 16 * WS_MAXNF ? left : 80;
            GSCAN_CHECK(args.npairs < WS_MAXPAIRS, "gemm (weights-stationary): more than %d column blocks in one launch", WS_MAXPAIRS);
            WsPair &pr = args.pair[args.npairs];
            pr = WsPair{i, n0, ncols, cdiv(ncols, 16), 0, 0, cdiv(g.M, WS_BM)};
            mac[args.npairs] = (double)g.M * (16.0 * pr.nf) * g.K;
            total += mac[args.npairs++];
            n0 += ncols;
        }
    }
    GSCAN_CHECK(lds_floats * sizeof(float) <= 160 * 1024, "gemm (weights-stationary): %zu bytes of LDS", lds_floats * sizeof(float));
    // workgroups per pair in proportion to its multiply-adds (at least one, at most one per tile), largest remainders first
    int given = 0;
    double frac[WS_MAXPAIRS];
    for (int i = 0; i < args.npairs; ++i) {
        const double share = cus * mac[i] / total;
        int n = std::max(1, std::min(args.pair[i].mtiles, (int)share));
        frac[i] = share - n;
        args.pair[i].wg_count = n;
        given += n;
    }
    while (given < cus) {
        int best = -1;
        for (int i = 0; i < args.npairs; ++i)
            if (args.pair[i].wg_count < args.pair[i].mtiles && (best < 0 || frac[i] > frac[best])) best = i;
        if (best < 0) break;
        ++args.pair[best].wg_count; frac[best] -= 1.0; ++given;
    }
    int at = 0;
    for (int i = 0; i < args.npairs; ++i) { args.pair[i].wg_begin = at; at += args.pair[i].wg_count; }
    static bool attr_set = false;
    if (!attr_set) {
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_ws_kernel, dim3(at), dim3(WS_THREADS), lds_floats * sizeof(float), stream, args);
    GSCAN_LAUNCHED("gemm_ws_kernel");
    return 0;
}

GSCAN_TRACE_TU(gemm_ws)

}  // namespace gscan
