// Strided fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, one
// rounding per product, so results track an fp32 fmaf chain).
//
//   C[M,N] = act(alpha * A.B + beta * C + bias[n]) * mask[m,n]           (+ optional  asum[m] += sum_k A(m,k))
//
// Every dense contraction of the training step that is not inside a per-row kernel goes through
// this kernel: key/value projections, LSTM input projections, data-gradient products
// and all weight-gradient products (K = B*T rows, split over workgroups).
// Operands are addressed with (row, col) strides so the reference's [out,in] parameter
// layout and its transposes are consumed in place; nothing is re-packed in HBM.
//
// These products are small (K <= 400 or M,N <= 400), so a workgroup's time is a chain of
// K/BK dependent "load tile -> MFMA" rounds, not FLOPs.  Hence: BK = 32 (few rounds), the next
// tile's global loads are issued into registers BEFORE the current tile's MFMAs and written to
// the other LDS buffer after them (one barrier per round), and the long-K weight-gradient
// products are split into K slices with float-atomic accumulation into the zeroed gradient.
// Tile geometry and LDS images are described next to the kernel below.
#include "gemm_panel.h"

namespace gscan {


template <int TMW, int BK, int KCA, int KCB, int VWA, int VWB>
__device__ __forceinline__ void gemm_tile(const GemmProblem &g, int local, float (&lds_a)[2][TileM<TMW, BK>::A_FLOATS],
                                          float (&lds_b)[2][b_floats<BK>()]) {
    constexpr int BM = TileM<TMW, BK>::BM, LDR_A = TileM<TMW, BK>::LDR_A;
    constexpr int LDK = BK + 4;
#ifdef GSCAN_GEMM_STAMPS
    unsigned gst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long gst_prev = clock64();
#endif
    // quotients by multiply-high with the host's reciprocals: four integer divisions (each a float-reciprocal sequence
    // through the vector unit) used to sit between the descriptor and the first global load
    const int bz = g.inv_mn ? (int)__umulhi((uint32_t)local, g.inv_mn) : local, rem = local - bz * g.tiles_mn;   // inv 0: d = 1
    // Tile order inside a problem: M tiles slowest by default; N tiles slowest (flag 16) when B is the larger operand,
    // so that an XCD's contiguous share of the tiles reads a slice of the LARGE operand and all of the small one.
    const bool n_major = (g.flags & 16) != 0;
    const int inner = g.inv_in ? (int)__umulhi((uint32_t)rem, g.inv_in) : rem;   // rem / tiles_m (N-major) or rem / tiles_n
    const int by = n_major ? rem - inner * (int)g.tiles_m : inner, bx = n_major ? inner : rem - inner * g.tiles_n;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const bool do_asum = g.asum1 != nullptr && bx == 0;

    PanelIter<BM, BK, KCA, VWA> pa;
    PanelIter<BN, BK, KCB, VWB> pb;
#ifdef GSCAN_GEMM_SAMEPANEL   // timing experiment (wrong results): every workgroup reads tile (0, 0)'s panels: L1 / L2 hits only
    pa.init(g.a, g.sam, g.sak, g.M, 0, 1 << (g.flags & 3), kbeg, kend, tid);
    pb.init(g.b, g.sbn, g.sbk, g.N, 0, 1 << ((g.flags >> 2) & 3), kbeg, kend, tid);
#else
    pa.init(g.a, g.sam, g.sak, g.M, m0, 1 << (g.flags & 3), kbeg, kend, tid);
    pb.init(g.b, g.sbn, g.sbk, g.N, n0, 1 << ((g.flags >> 2) & 3), kbeg, kend, tid);
#endif

    f32x4 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;

    const int fr = lane & 15;   // MFMA row (A) / column (B) index
    const int fg = lane >> 4;   // lane group: k = 8*fg + step
    // this lane's fragment bases inside an LDS image (constant over the K loop)
    const int fa = pa.is_kc() ? (wm * 16 * TMW + fr) * LDK + 8 * fg : (8 * fg) * LDR_A + wm * 16 * TMW + TMW * fr;
    const int fb = pb.is_kc() ? (wn * 16 * TNW + fr) * LDK + 8 * fg : (8 * fg) * LDR_B + wn * 16 * TNW + TNW * fr;

    // K loop.  A round = fragment reads of the current LDS buffer, then the round's MFMAs with the panel loads of a LATER round
    // pinned between them (sched_group_barrier: one 16-byte load per 5-8 MFMAs — issued in front, a wave sat in the memory
    // pipeline's queue for ~1 000 cycles before it reached its own matrix work), then the panels of the NEXT round go from
    // registers to the other LDS buffer, one barrier.
    // GSCAN_GEMM_PIPE=3 (round 6): the 32-row kernel keeps TWO register sets and loads two rounds ahead — a round's
    // panels have a whole round to arrive before they are stored, where one round ahead the wave waited ~1 500 cycles in
    // front of its LDS stores every round (12 more VGPRs: 111, still four workgroups per CU; the 64-row kernel has no
    // registers for it and stays one round ahead).  2: one round ahead everywhere.
#ifndef GSCAN_GEMM_PIPE
#define GSCAN_GEMM_PIPE 2
#endif
    constexpr bool DEEP = GSCAN_GEMM_PIPE == 3 && TMW == 1 && BK == 32;
    constexpr int NSET = DEEP ? 2 : 1, AHEAD = DEEP ? 2 : 1;
    float ra[NSET][BM * BK / 256], rb[NSET][BN * BK / 256];
    uint32_t ma[NSET], mb[NSET];
    GST(0)                                   // index math and iterator set-up
    ma[0] = pa.load(ra[0], kbeg);
    mb[0] = pb.load(rb[0], kbeg);
    GST(1)                                   // first loads issued
    pa.template store<LDR_A>(lds_a[0], ra[0], ma[0], tid);
    pb.template store<LDR_B>(lds_b[0], rb[0], mb[0], tid);
    if constexpr (DEEP) {                    // the second round's panels: in flight across the first barrier
        ma[1] = pa.load(ra[1], kbeg + BK);
        mb[1] = pb.load(rb[1], kbeg + BK);
    }
    __syncthreads();
    GST(2)                                   // first panels landed and staged

    // one round on LDS buffer `buf`: loads of round + AHEAD into register set L, set S (the next round's panels) to the other buffer
    auto round = [&](int k0, int buf, auto lset, auto sset) {
        constexpr int L = decltype(lset)::value, S = decltype(sset)::value;
        // (the two inlined copies of this body differ in the register sets only: with `buf` a literal per copy the
        // compiler kept both buffers' fragment and store addresses live across the loop — 134 VGPRs instead of 91)
        asm volatile("" : "+s"(buf));
        const bool more = k0 + BK < kend;
        GST(3)
#pragma unroll
        for (int kh = 0; kh < BK; kh += 32) {
        const float *la = lds_a[buf] + fa + (pa.is_kc() ? kh : kh * LDR_A), *lb = lds_b[buf] + fb + (pb.is_kc() ? kh : kh * LDR_B);
        float af[TMW][8], bf[TNW][8];        // [tile][step]
        if (pa.is_kc()) {                         // natural tiles: row = 16 t + fr
#pragma unroll
            for (int t = 0; t < TMW; ++t) {
                const float4 *q = reinterpret_cast<const float4 *>(la + 16 * t * LDK);
                const float4 x = q[0], y = q[1];
                af[t][0] = x.x; af[t][1] = x.y; af[t][2] = x.z; af[t][3] = x.w;
                af[t][4] = y.x; af[t][5] = y.y; af[t][6] = y.z; af[t][7] = y.w;
            }
        } else {                             // interleaved tiles: row = TMW fr + t
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if constexpr (TMW == 2) {
                    const float2 x = *reinterpret_cast<const float2 *>(la + s * LDR_A);
                    af[0][s] = x.x; af[1][s] = x.y;
                } else {
                    af[0][s] = la[s * LDR_A];
                }
            }
        }
        if (pb.is_kc()) {                         // natural tiles: col = 16 t + fr
#pragma unroll
            for (int t = 0; t < TNW; ++t) {
                const float4 *q = reinterpret_cast<const float4 *>(lb + 16 * t * LDK);
                const float4 x = q[0], y = q[1];
                bf[t][0] = x.x; bf[t][1] = x.y; bf[t][2] = x.z; bf[t][3] = x.w;
                bf[t][4] = y.x; bf[t][5] = y.y; bf[t][6] = y.z; bf[t][7] = y.w;
            }
        } else {                             // interleaved tiles: col = 2 fr + t
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float2 x = *reinterpret_cast<const float2 *>(lb + s * LDR_B);
                bf[0][s] = x.x; bf[1][s] = x.y;
            }
        }
        if (kh == 0) {                       // always issued: a load beyond the K range reads element 0 and is zeroed when stored
            ma[L] = pa.load(ra[L], k0 + AHEAD * BK);
            mb[L] = pb.load(rb[L], k0 + AHEAD * BK);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        if constexpr (VWA == 4 && VWB == 4 && BK == 32) {      // TMW + 2 sixteen-byte loads, 16 TMW MFMAs
#define GSCAN_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
            if constexpr (TMW == 1) {
                GSCAN_SGB(0x020, 1); GSCAN_SGB(0x008, 5);      // one VMEM read, then its share of the MFMAs
                GSCAN_SGB(0x020, 1); GSCAN_SGB(0x008, 5);
                GSCAN_SGB(0x020, 1); GSCAN_SGB(0x008, 6);
            } else {
                GSCAN_SGB(0x020, 1); GSCAN_SGB(0x008, 8);
                GSCAN_SGB(0x020, 1); GSCAN_SGB(0x008, 8);
                GSCAN_SGB(0x020, 1); GSCAN_SGB(0x008, 8);
                GSCAN_SGB(0x020, 1); GSCAN_SGB(0x008, 8);
            }
#undef GSCAN_SGB
        }
        if (do_asum && tid < BM) {           // column sums of A (bias gradients): sum over these 32 k
            const float *sa = lds_a[buf] + (pa.is_kc() ? kh : kh * LDR_A);
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            if (pa.is_kc()) {
#pragma unroll
                for (int kk = 0; kk < 32; kk += 4) {
                    const float4 x = *reinterpret_cast<const float4 *>(sa + tid * LDK + kk);
                    t0 += x.x; t1 += x.y; t2 += x.z; t3 += x.w;
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 32; kk += 4) {
                    t0 += sa[kk * LDR_A + tid]; t1 += sa[(kk + 1) * LDR_A + tid];
                    t2 += sa[(kk + 2) * LDR_A + tid]; t3 += sa[(kk + 3) * LDR_A + tid];
                }
            }
            asum += (t0 + t1) + (t2 + t3);
        }
        }
        GST(4)                               // fragment reads + MFMAs
        if (more) {
            pa.template store<LDR_A>(lds_a[buf ^ 1], ra[S], ma[S], tid);
            pb.template store<LDR_B>(lds_b[buf ^ 1], rb[S], mb[S], tid);
        }
        GST(5)                               // wait for the next panels + LDS stores
        __syncthreads();
        GST(6)
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, DEEP ? 1 : 0>;
    if constexpr (DEEP) {
        for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {     // two copies of the body: the register sets swap roles every round
            round(k0, 0, Set0{}, Set1{});                  // loads into set 0, stores set 1
            if (k0 + BK >= kend) break;
            round(k0 + BK, 1, Set1{}, Set0{});
        }
    } else {
        int buf = 0;
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            round(k0, buf, Set0{}, Set0{});
            buf ^= 1;
        }
    }

    if (do_asum && tid < BM && m0 + tid < g.M) {
        atomicAdd(&g.asum1[m0 + tid], asum);
        if (g.asum2) atomicAdd(&g.asum2[m0 + tid], asum);
    }

    // epilogue.  MFMA C/D fragment: column index = lane & 15, row index = (lane >> 4) * 4 + reg.
    // Rows / columns of this lane and their validity first, then ONE of three store loops (split-K atomics, plain
    // store, general epilogue): no per-element mode tests or 64-bit index arithmetic.
    int coff[TNW];
    bool cok[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int col = n0 + wn * 16 * TNW + (pb.is_kc() ? 16 * j + fr : TNW * fr + j);
        coff[j] = col;
        cok[j] = col < g.N;
    }
    const uint32_t ldc = (uint32_t)g.ldc;
    const float alpha = g.alpha;
    gfloat *gc = as_global(g.c);
    const gfloat *gbias = as_global(g.bias), *ggate = as_global(g.gate), *gmask = as_global(g.mask);
    const bool plain = g.beta == 0.f && !g.bias && g.act == 0 && !g.mask;
    if (g.atomic || plain) {
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ri = fg * 4 + r;                                    // MFMA row index 0..15
                const int row = m0 + wm * 16 * TMW + (pa.is_kc() ? 16 * i + ri : TMW * ri + i);
                if (row >= g.M) continue;
                const uint32_t roff = (uint32_t)row * ldc;
                if (g.atomic) {
#pragma unroll
                    for (int j = 0; j < TNW; ++j)
                        if (cok[j]) atomicAdd(g.c + (roff + coff[j]), alpha * acc[i][j][r]);
                } else {
#pragma unroll
                    for (int j = 0; j < TNW; ++j)
                        if (cok[j]) gc[roff + coff[j]] = alpha * acc[i][j][r];
                }
            }
    } else {
        // General epilogue (round 5): EVERY value it reads — the old C (beta), the bias, the ReLU gate, the mask — is requested
        // first, for all of the thread's elements, and waited for once.  It used to read them element by element inside the
        // store loop: the compiler's code was load, s_waitcnt vmcnt(0), use, eight times over — eight dependent L2 round trips
        // per workgroup of the dS += launch (beta = 1, on the step's critical chain) and of the biased products.
        float cold[TMW][4][TNW], gat[TMW][4][TNW], msk[TMW][4][TNW], bia[TNW];
        uint32_t at[TMW][4][TNW];
        bool ok[TMW][4][TNW];
#pragma unroll
        for (int j = 0; j < TNW; ++j) bia[j] = (gbias && cok[j]) ? gbias[coff[j]] : 0.f;
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ri = fg * 4 + r;
                const int row = m0 + wm * 16 * TMW + (pa.is_kc() ? 16 * i + ri : TMW * ri + i);
                const uint32_t roff = (uint32_t)min(row, g.M - 1) * ldc;
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    ok[i][r][j] = row < g.M && cok[j];
                    at[i][r][j] = roff + (cok[j] ? coff[j] : 0);              // a valid address either way: the loads are unconditional
                    cold[i][r][j] = g.beta != 0.f ? gc[at[i][r][j]] : 0.f;
                    gat[i][r][j] = g.act == 3 ? ggate[at[i][r][j]] : 1.f;
                    msk[i][r][j] = gmask ? gmask[at[i][r][j]] : 1.f;
                }
            }
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    float v = alpha * acc[i][j][r];
                    if (g.beta != 0.f) v += g.beta * cold[i][r][j];
                    if (gbias) v += bia[j];
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    else if (g.act == 2) v = tanhf_(v);
                    else if (g.act == 3 && gat[i][r][j] == 0.f) v = 0.f;      // ReLU backward
                    if (gmask) v *= msk[i][r][j];
                    if (ok[i][r][j]) gc[at[i][r][j]] = v;
                }
    }
    GST(7)                                   // epilogue
#ifdef GSCAN_GEMM_STAMPS
    if (g_trace_buf && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_trace_buf[1500 + i] += gst_acc[i];
#endif
}

// GSCAN_GEMM_WAVES5 (round 6 A/B): the 32-row kernel needs 99 registers = a 104-register allocation = four workgroups per
// CU; asking for five waves per SIMD (<= 96 registers) lets the fifth workgroup its 27.6 KB of LDS already allow in.
#ifndef GSCAN_GEMM_WAVES5
#define GSCAN_GEMM_WAVES5 0
#endif
template <int TMW, int BK>
#ifndef GSCAN_GEMM_WAVES4
#define GSCAN_GEMM_WAVES4 0        // 1 (A/B): the 64-row kernel held to 128 registers = four workgroups per CU (it takes 134 with the pinned loads)
#endif
__global__ __launch_bounds__(256, (GSCAN_GEMM_WAVES5 && TMW == 1 && BK == 32) ? 5 : (GSCAN_GEMM_WAVES4 && TMW == 2 && BK == 32) ? 4 : 1) void gemm_group_kernel(int tb0, int tb1, int tb2, int tb3, int tb4, int tb5, int tb6,
                                                         int tb7, int tb8, int tb9, int tb10, int tb11, GemmGroup grp) {
    // The first workgroup id of every problem arrives as twelve leading scalar arguments: this file is compiled with
    // -mllvm -amdgpu-kernarg-preload-count=12 (build.py), so they sit in SGPRs when the wave starts and the problem
    // lookup costs no memory round trip (firmware without kernarg preload runs the compiler's loading preamble).
    const int tb[kMaxGroup] = {tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7, tb8, tb9, tb10, tb11};
    static_assert(kMaxGroup == 12, "the preloaded header is twelve scalars");
    TraceScope trace_scope(TK_GEMM);
    constexpr int A_FLOATS = TileM<TMW, BK>::A_FLOATS, B_FLOATS = b_floats<BK>();
    // two scalar-load round trips before the first global load: the header (unused entries hold INT_MAX), then the
    // problem's whole descriptor by value
    int pi = 0, first = tb[0];
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if ((int)blockIdx.x >= tb[i]) { pi = i; first = tb[i]; }
    asm volatile("" : "+s"(pi), "+s"(first));                 // selected from the preloaded registers
    int per = grp.xcd_per[pi];
    GemmProblem g = grp.p[pi];
    // every field is wanted NOW (one batch of scalar loads, one wait), not each at its first use behind the
    // previous one's wait
    asm volatile("" : "+s"(g.M), "+s"(g.N), "+s"(g.K), "+s"(g.alpha), "+s"(g.beta), "+s"(g.a), "+s"(g.sam), "+s"(g.sak),
                      "+s"(g.b), "+s"(g.sbk), "+s"(g.sbn), "+s"(g.c), "+s"(g.ldc));
    asm volatile("" : "+s"(g.bias), "+s"(g.act), "+s"(g.mask), "+s"(g.gate), "+s"(g.k_chunk), "+s"(g.atomic),
                      "+s"(g.asum1), "+s"(g.asum2), "+s"(g.tiles_n), "+s"(g.tiles_mn), "+s"(g.nsplit), "+s"(g.flags),
                      "+s"(g.inv_mn), "+s"(g.inv_in), "+s"(g.tiles_m),
                      "+s"(per));
    // XCD-aware order (GSCAN_GEMM_XCD=0 disables): workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.
    // Workgroup l of a problem takes tile (l % 8) * per + l / 8, so an XCD works on one contiguous eighth of the
    // tile space and its L2 sees each operand panel of that eighth once.
    int local = blockIdx.x - first;
    if (per > 0) {
        const int x = local & 7, j = local >> 3;
        local = x * per + j;
        if (j >= per || local >= g.tiles_mn * g.nsplit) return;
    }
    __shared__ __attribute__((aligned(16))) float lds_a[2][A_FLOATS];
    __shared__ __attribute__((aligned(16))) float lds_b[2][B_FLOATS];
    // one workgroup-uniform dispatch to a copy of the tile code specialised for the problem's operand layouts
    // (16-byte loads on both operands: every large product of the step); anything else takes the generic copy
    const bool kca = g.sak == 1, kcb = g.sbk == 1;
    const int wa = g.flags & 3, wb = (g.flags >> 2) & 3;             // log2 of the load widths
    if (wa == 2 && wb == 2) {
        if (kca && kcb) gemm_tile<TMW, BK, 1, 1, 4, 4>(g, local, lds_a, lds_b);
        else if (kca) gemm_tile<TMW, BK, 1, 0, 4, 4>(g, local, lds_a, lds_b);
        else if (!kcb) gemm_tile<TMW, BK, 0, 0, 4, 4>(g, local, lds_a, lds_b);
        else gemm_tile<TMW, BK, 0, 1, 4, 4>(g, local, lds_a, lds_b);
    } else if (wa == 1 && wb == 1 && kca && kcb) {                 // rows of 150 features: 8-byte loads
        gemm_tile<TMW, BK, 1, 1, 2, 2>(g, local, lds_a, lds_b);
    } else if (wa == 2 && wb == 1 && !kca && !kcb) {               // weight gradient against 150-wide rows
        gemm_tile<TMW, BK, 0, 0, 4, 2>(g, local, lds_a, lds_b);
    } else {
        gemm_tile<TMW, BK, -1, -1, 0, 0>(g, local, lds_a, lds_b);
    }
}



void GemmBatch::add(int M, int N, int K, const float *a, int64_t sam, int64_t sak, const float *b, int64_t sbk,
                    int64_t sbn, float *c, int64_t ldc, float beta, const float *bias, int act, const float *mask,
                    int split_k, float *asum1, float *asum2, const float *gate, float alpha) {
    if (bad_) return;
    if (grp_.count >= kMaxGroup || M <= 0 || N <= 0 || K <= 0 || !a || !b || !c || act < 0 || act > 3 ||
        (act == 3 && !gate)) {
        bad_ = true;
        set_error("gemm batch: bad problem %d (%dx%dx%d act=%d) or more than %d problems", grp_.count, M, N, K, act,
                  kMaxGroup);
        return;
    }
    // split_k < 0: accumulate with atomics even if K turns out too short to be split (several products of one launch
    // adding into the same C)
    const bool force_atomic = split_k < 0;
    if (force_atomic) split_k = -split_k;
    if (split_k < 1) split_k = 1;
    int chunk = cdiv(K, split_k);
    chunk = cdiv(chunk, 64) * 64;          // K slices start at multiples of the deepest K round
    split_k = cdiv(K, chunk);
    if ((split_k > 1 || force_atomic) && !(beta == 1.f && act == 0 && !bias && !mask)) {
        bad_ = true;
        set_error("gemm batch: split-K needs beta=1 and no epilogue (problem %d, beta=%g act=%d)", grp_.count, beta, act);
        return;
    }
    // Width of the global loads of an operand (4 / 2 / 1 floats): unit stride along the load direction, the other
    // stride and the extent along the load direction multiples of the width (tile edges included: K slices start at
    // multiples of 32), and a base aligned to it.  flags = log2(width of A) | log2(width of B) << 2.
    auto width = [](const float *ptr, int64_t s_row, int64_t s_k, int rows, int K) {
        for (int w = 4; w > 1; w >>= 1) {
            const bool base = (reinterpret_cast<uintptr_t>(ptr) & (4 * w - 1)) == 0;
            if (base && ((s_k == 1 && s_row % w == 0 && K % w == 0) || (s_row == 1 && s_k % w == 0 && rows % w == 0)))
                return w == 4 ? 2 : 1;
        }
        return 0;
    };
    const int flags = width(a, sam, sak, M, K) | (width(b, sbn, sbk, N, K) << 2);
    if ((int64_t)M * sam + (int64_t)K * sak >= (1ll << 32) || (int64_t)N * sbn + (int64_t)K * sbk >= (1ll << 32) ||
        (int64_t)M * ldc >= (1ll << 32)) {
        bad_ = true;
        set_error("gemm batch: operand of problem %d spans more than 2^32 elements", grp_.count);
        return;
    }
    GemmProblem &p = grp_.p[grp_.count++];
    p = GemmProblem{M, N, K, alpha, beta, a, sam, sak, b, sbk, sbn, c, ldc, bias, act, mask, gate, chunk,
                    (split_k > 1 || force_atomic) ? 1 : 0, asum1, asum2, 0, 0, 0, 0u, 0u, flags, 0, 0,   // tile bookkeeping: at launch
                    nullptr, force_atomic ? 2 : (split_k > 1 ? 1 : 0)};   // 2: shares its C with another product of the launch
    tiles_ += cdiv(N, BN) * cdiv(M, 64) * split_k;                            // in 64-row tiles
    last_flops_ = 2.0 * M * N * K;
    flops_ += last_flops_;
    alg_flops_ += last_flops_;
}

// Launches whose 64-row tiling has fewer workgroups than this use 32-row tiles.  Measured on the training step's
// shapes (tools/gemm_shapes.py): 32-row tiles win or tie on every product but the largest (uv 9216x400x150: 35 vs
// 40 us; conv 256x5400x576: 36 vs 44 us; dW_ih 400x300x5120 split 8: 44 vs 48 us), 64-row tiles win once a launch
// has thousands of workgroups (4096^3: 78 vs 72 TFLOP/s).  GSCAN_GEMM_TMW=1|2 forces a shape, for experiments.
constexpr int kWideTileMinGroups = 2048;

// Launch on the 128 x 128 macro tiles of gemm_mt.hip.  Every problem keeps its tiles whole unless the caller allowed a
// K split (weight gradients: beta = 1, no epilogue); those are cut into slices of R rounds each, ONE R for the whole
// launch, the largest for which the launch has about kMtTarget workgroups (three per CU) — so that every workgroup of
// the launch does about the same work and the launch is a whole number of chip generations — but never below
// kMtMinRounds rounds per slice (a slice pays its first panel's latency and its slab).  With scratch the slices
// leave partial tiles in slabs that a second launch adds up in slice order; without, they add with float atomics.
constexpr int kMtTarget = 768, kMtMinRounds = 6;      // three workgroups per CU: 0.515 ms per step against 0.566 at 512 (r04 A/B)
int GemmBatch::launch_macro_tiles(hipStream_t stream) {
    static const int target = [] { const char *e = getenv("GSCAN_MT_TARGET"); return e ? atoi(e) : kMtTarget; }();
    static const int min_rounds = [] { const char *e = getenv("GSCAN_MT_MINR"); return e ? atoi(e) : kMtMinRounds; }();
    static const int forced_bm = [] { const char *e = getenv("GSCAN_MT_BM"); return e ? atoi(e) : 0; }();
    static const int xcd = [] { const char *e = getenv("GSCAN_GEMM_XCD"); return e ? atoi(e) : 1; }();
    static const int order = [] { const char *e = getenv("GSCAN_GEMM_ORDER"); return e ? atoi(e) : 1; }();
    const size_t slab = gemm_mt_slab_floats();
    const size_t slab_cap = scratch_ ? scratch_floats_ / slab : 0;
    // Products that share their C with another product of the launch (split_ok == 2: float atomics, never slabs) are cut
    // into K slices like any split product — except in a launch that was handed scratch, i.e. one that asked for
    // fixed-order sums: there each of them stays ONE slice, so that an element of C receives exactly one add per
    // product onto the zeroed buffer, and two float adds commute bit for bit (the deep encoder's dX products of the two
    // directions, step.hip; with K = 4 He = 400 cut into six slices the last bits moved from run to run).
    auto splittable = [&](const GemmProblem &p) { return p.split_ok == 1 || (p.split_ok == 2 && !scratch_); };
    // tiles of `bm` rows; returns the workgroups of the launch at R rounds per slice and the slabs that takes
    auto plan = [&](int bm, int R, size_t *slabs) {
        int total = 0;
        *slabs = 0;
        for (int i = 0; i < grp_.count; ++i) {
            const GemmProblem &p = grp_.p[i];
            int tn, nf;
            gemm_mt_columns(p.N + (p.asum1 ? 1 : 0), &tn, &nf);
            const int tiles = tn * cdiv(p.M, bm);
            const int slices = splittable(p) ? cdiv(cdiv(p.K, 32), R) : 1;
            total += tiles * slices;
            if (slices > 1 && p.split_ok == 1) *slabs += (size_t)tiles * slices;
        }
        return total;
    };
    int max_rounds = 1;
    for (int i = 0; i < grp_.count; ++i)
        if (splittable(grp_.p[i])) max_rounds = std::max(max_rounds, cdiv(grp_.p[i].K, 32));
    auto pick_rounds = [&](int bm, int *total) {
        int R = max_rounds;
        size_t slabs = 0;
        while (R > min_rounds && plan(bm, R, &slabs) < target) {
            size_t next_slabs;
            plan(bm, R - 1, &next_slabs);
            if (scratch_ && next_slabs > slab_cap) break;       // the slabs of a finer split would not fit the scratch
            --R;
        }
        *total = plan(bm, R, &slabs);
        return R;
    };
    // 128-row tiles when they alone give the launch its workgroups, else 64-row tiles (twice the workgroups, 2/3 of
    // the flop per byte); GSCAN_MT_BM=64|128 forces one (experiments)
    int total128 = 0, total64 = 0;
    const int R128 = pick_rounds(128, &total128), R64 = pick_rounds(64, &total64);
    const int bm = forced_bm == 64 || forced_bm == 128 ? forced_bm : (total128 >= target ? 128 : 64);
    const int R = bm == 128 ? R128 : R64;
    size_t slabs = 0;
    plan(bm, R, &slabs);
    const bool use_slabs = scratch_ && slabs <= slab_cap;
    int total = 0, total_units = 0, unit_begin[kMaxGroup];
    size_t slab_at = 0;
    for (int i = 0; i < kMaxGroup; ++i) { grp_.tile_begin[i] = INT_MAX; unit_begin[i] = INT_MAX; }
    for (int i = 0; i < grp_.count; ++i) {
        GemmProblem &p = grp_.p[i];
        gemm_mt_columns(p.N + (p.asum1 ? 1 : 0), &p.tiles_n, &p.nf);
        p.tiles_m = cdiv(p.M, bm);
        p.tiles_mn = p.tiles_n * p.tiles_m;
        p.k_chunk = splittable(p) ? 32 * R : cdiv(p.K, 32) * 32;
        p.nsplit = cdiv(p.K, p.k_chunk);
        // several products of one launch adding into one C (split_k < 0 at add()): float atomics whatever the slice count
        p.atomic = p.split_ok == 2 ? 1 : (p.nsplit > 1 ? (use_slabs ? 2 : 1) : 0);
        unit_begin[i] = total_units;
        if (p.atomic == 2) {
            p.slab = scratch_ + slab_at * slab;
            slab_at += (size_t)p.tiles_mn * p.nsplit;
            total_units += p.tiles_mn * 64;
        }
        if (order && p.nsplit == 1 && p.N > p.M) p.flags |= 16;
        const uint32_t inner = (p.flags & 16) ? (uint32_t)p.tiles_m : (uint32_t)p.tiles_n;
        p.inv_mn = p.tiles_mn > 1 ? (uint32_t)((1ull << 32) / (uint32_t)p.tiles_mn) + 1u : 0u;    // 0 stands for d = 1
        p.inv_in = inner > 1 ? (uint32_t)((1ull << 32) / inner) + 1u : 0u;
        if (((int64_t)p.tiles_mn * p.nsplit + p.tiles_mn) * p.tiles_mn >= (1ll << 32)) {
            bad_ = true;
            set_error("gemm batch: problem %d has too many tiles for the reciprocal index arithmetic", i);
            return 1;
        }
        grp_.tile_begin[i] = total;
        const int n = p.tiles_mn * p.nsplit;
        grp_.xcd_per[i] = (xcd && n >= 16) ? cdiv(n, 8) : 0;
        total += grp_.xcd_per[i] ? 8 * grp_.xcd_per[i] : n;
    }
    ProbeScope probe(P_GEMM, stream, flops_, alg_flops_);
    if (int rc = gemm_mt_launch(grp_, total, bm, stream)) return rc;
    if (total_units > 0) return gemm_mt_reduce_launch(grp_, unit_begin, total_units, bm, stream);
    return 0;
}

// GSCAN_GEMM_MT if set (0 never, 1 always), else 1 in deterministic mode (GSCAN_DETERMINISTIC=1), else -1 (by rule)
int gemm_macro_tile_mode() {
    static const int mode = [] {
        const char *e = getenv("GSCAN_GEMM_MT"), *d = getenv("GSCAN_DETERMINISTIC");
        return e ? atoi(e) : (d && atoi(d) ? 1 : -1);
    }();
    return mode;
}

// Launches with at least this many 128-row macro tiles whose products are all at least kMacroMinK deep take gemm_mt.hip
// by rule (4096^3: 105 TFLOP/s there against 94 on the 64 x 64 tiles below).  The depth condition keeps the training
// step's launches off it at every batch size: its many-tile launches are the forward products with K = 100-150 (four
// K rounds, then a 16 K-float epilogue per workgroup), and with them on the macro tiles S3 (T = 120: 1 400 macro tiles in
// the forward launch) ran 1.817 ms per step against 1.783, S1 at 1 024 rows 1.776 against 1.745
// (profiles/r04_gemm_macro_rule_ab.txt).
constexpr int kMacroMinTiles = 1024, kMacroMinK = 512;

int GemmBatch::launch(hipStream_t stream) {
    if (bad_) return 1;
    if (grp_.count == 0) return 0;
    // Which kernel.  The training step's launches are SMALL — 0.17 to 1.4 GMAC each, one to four 128 x 128 x 32 rounds
    // per CU — and what bounds them is how evenly and how early the work reaches all 256 CUs, not bytes per flop: on
    // the whole step the macro tiles of gemm_mt.hip lose to the 32 x 64 tiles below with every planning parameter
    // tried (0.515-0.577 ms per step against 0.494: profiles/r04_gemm_macro_tiles_ab.txt), although they win once a
    // launch has thousands of tiles.  So: macro tiles for large launches by rule, and for every launch in
    // DETERMINISTIC mode (GSCAN_DETERMINISTIC=1, or GSCAN_GEMM_MT=1), where their split-K partial tiles are added in a
    // fixed order instead of with float atomics — bitwise reproducible weight gradients for 4 % of step time.
    // GSCAN_GEMM_MT=0: never.
    // A launch that was handed scratch for split-K slabs asked for the fixed-order sums: macro tiles too.
    // The weights-stationary persistent kernel of gemm_ws.hip (built, parity-green, measured — and
    // SLOWER than the 32 x 64 tiles on every launch of the step: forward launch 63 -> 46 us over three versions against
    // 36-43 us, step 0.488-0.505 ms against 0.476; profiles/r05_gemm_weights_stationary_*.txt).  OFF unless
    // GSCAN_GEMM_WS asks for it: 1 launches of tall-skinny forward products (every operand k-contiguous, nothing split,
    // K <= 160) from 32 M multiply-adds on, 2 every eligible launch, 3 the same and an ineligible launch is an error (tests).
    {
        static const int ws_mode = [] { const char *e = getenv("GSCAN_GEMM_WS"); return e ? atoi(e) : 0; }();
        if (ws_mode > 0 && !scratch_ && !force_mt_) {
            bool all = true;
            double macs = 0.0;
            for (int i = 0; i < grp_.count; ++i) {
                all = all && gemm_ws_eligible(grp_.p[i]);
                macs += (double)grp_.p[i].M * grp_.p[i].N * grp_.p[i].K;
            }
            if (all && (ws_mode > 1 || macs >= 32e6)) {
                ProbeScope probe(P_GEMM, stream, flops_, alg_flops_);
                return gemm_ws_launch(grp_, stream);
            }
            if (ws_mode == 3) {                              // tests: the launch was meant for that kernel
                set_error("gemm batch: GSCAN_GEMM_WS=3 but a product of the launch is not eligible for the weights-stationary kernel");
                return 1;
            }
        }
    }
    const int mt_mode = (scratch_ || (force_mt_ && gemm_macro_tile_mode() != 0)) ? 1 : gemm_macro_tile_mode();
    if (mt_mode != 0) {
        int macro_tiles = 0, k_min = INT_MAX;
        for (int i = 0; i < grp_.count; ++i) {
            int tn, nf;
            gemm_mt_columns(grp_.p[i].N + (grp_.p[i].asum1 ? 1 : 0), &tn, &nf);
            macro_tiles += tn * cdiv(grp_.p[i].M, 128);
            k_min = std::min(k_min, grp_.p[i].K);
        }
        if (mt_mode > 0 || (macro_tiles >= kMacroMinTiles && k_min >= kMacroMinK)) return launch_macro_tiles(stream);
    }
    static const int forced = [] { const char *e = getenv("GSCAN_GEMM_TMW"); return e ? atoi(e) : 0; }();
    // Launches whose 64-row tiling has fewer workgroups than this use 32-row tiles (see kWideTileMinGroups above)
    const int tmw = forced == 1 || forced == 2 ? forced : (tiles_ < kWideTileMinGroups ? 1 : 2);
    static const int xcd = [] { const char *e = getenv("GSCAN_GEMM_XCD"); return e ? atoi(e) : 1; }();   // on by default
    static const int order = [] { const char *e = getenv("GSCAN_GEMM_ORDER"); return e ? atoi(e) : 1; }();
    int total = 0;
    for (int i = 0; i < kMaxGroup; ++i) grp_.tile_begin[i] = INT_MAX;
    for (int i = 0; i < grp_.count; ++i) {
        GemmProblem &p = grp_.p[i];
        p.tiles_n = cdiv(p.N, BN);
        p.tiles_mn = p.tiles_n * cdiv(p.M, 32 * tmw);
        p.nsplit = cdiv(p.K, p.k_chunk);
        if (order && p.nsplit == 1 && p.N > p.M) p.flags |= 16;
        p.tiles_m = p.tiles_mn / p.tiles_n;
        const uint32_t inner = (p.flags & 16) ? (uint32_t)p.tiles_m : (uint32_t)p.tiles_n;
        p.inv_mn = p.tiles_mn > 1 ? (uint32_t)((1ull << 32) / (uint32_t)p.tiles_mn) + 1u : 0u;    // 0 stands for d = 1
        p.inv_in = inner > 1 ? (uint32_t)((1ull << 32) / inner) + 1u : 0u;
        if (((int64_t)p.tiles_mn * p.nsplit + p.tiles_mn) * p.tiles_mn >= (1ll << 32)) {   // exactness of umulhi(x, inv)
            bad_ = true;
            set_error("gemm batch: problem %d has too many tiles for the reciprocal index arithmetic", i);
            return 1;
        }
        grp_.tile_begin[i] = total;
        const int n = p.tiles_mn * p.nsplit;
        grp_.xcd_per[i] = (xcd && n >= 16) ? cdiv(n, 8) : 0;
        total += grp_.xcd_per[i] ? 8 * grp_.xcd_per[i] : n;
    }
    const int *t = grp_.tile_begin;
#define TB t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11]
    ProbeScope probe(P_GEMM, stream, flops_, alg_flops_);
    // 64-deep K rounds (GSCAN_GEMM_BK=64, experiments): isolated long-K split products gain 10-20 % (a round's
    // load latency is paid half as often), but the overlapped training step loses 3 % to the larger workgroups
    // (52 KB of LDS, +40 VGPRs), so 32 is what every launch uses.
    static const int forced_bk = [] { const char *e = getenv("GSCAN_GEMM_BK"); return e ? atoi(e) : 0; }();
    const bool deep = forced_bk == 64;
    if (tmw == 1 && deep) hipLaunchKernelGGL((gemm_group_kernel<1, 64>), dim3(total), dim3(256), 0, stream, TB, grp_);
    else if (tmw == 1) hipLaunchKernelGGL((gemm_group_kernel<1, 32>), dim3(total), dim3(256), 0, stream, TB, grp_);
    else if (deep) hipLaunchKernelGGL((gemm_group_kernel<2, 64>), dim3(total), dim3(256), 0, stream, TB, grp_);
    else hipLaunchKernelGGL((gemm_group_kernel<2, 32>), dim3(total), dim3(256), 0, stream, TB, grp_);
#undef TB
    GSCAN_LAUNCHED("gemm_group_kernel");
    return 0;
}

int gemm_f32(int M, int N, int K, float alpha, const float *a, int64_t sam, int64_t sak, const float *b,
             int64_t sbk, int64_t sbn, float beta, float *c, int64_t ldc, const float *bias, int act,
             const float *mask, int split_k, hipStream_t stream) {
    GSCAN_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem %dx%dx%d", M, N, K);
    GSCAN_CHECK(a && b && c, "gemm: null operand");
    GSCAN_CHECK(act >= 0 && act <= 2, "gemm: unknown activation %d", act);
    GemmBatch batch;
    batch.add(M, N, K, a, sam, sak, b, sbk, sbn, c, ldc, beta, bias, act, mask, split_k, nullptr, nullptr, nullptr,
              alpha);
    return batch.launch(stream);
}

// The same product with the two extras the training step uses: asum (optional, [M]) += sum_k A(m,k) (bias gradients),
// and a scratch for split-K partial tiles, which are then added in a fixed order instead of with atomics.
int gemm_f32_ex(int M, int N, int K, float alpha, const float *a, int64_t sam, int64_t sak, const float *b,
                int64_t sbk, int64_t sbn, float beta, float *c, int64_t ldc, const float *bias, int act,
                const float *mask, int split_k, float *asum, float *scratch, size_t scratch_floats, hipStream_t stream) {
    GSCAN_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem %dx%dx%d", M, N, K);
    GSCAN_CHECK(a && b && c, "gemm: null operand");
    GSCAN_CHECK(act >= 0 && act <= 2, "gemm: unknown activation %d", act);
    GemmBatch batch;
    batch.scratch(scratch, scratch_floats);
    batch.add(M, N, K, a, sam, sak, b, sbk, sbn, c, ldc, beta, bias, act, mask, split_k, asum, nullptr, nullptr, alpha);
    return batch.launch(stream);
}

GSCAN_TRACE_TU(gemm)

}  // namespace gscan
