// Strided fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, one
// rounding per product, so results track an fp32 fmaf chain).
//
//   C[M,N] = act(alpha * A.B + beta * C + bias[n]) * mask[m,n]           (+ optional  asum[m] += sum_k A(m,k))
//
// Every dense contraction of the training step that is not inside a time loop goes through
// this kernel: the im2col convolutions, key/value projections, LSTM input projections, the
// output head, and all weight-gradient products (K = B*T rows, split over workgroups).
// Operands are addressed with (row, col) strides so the reference's [out,in] parameter
// layout and its transposes are consumed in place; nothing is re-packed in HBM.
//
// These products are small (K <= 400 or M,N <= 400), so a workgroup's time is a chain of
// K/BK dependent "load tile -> MFMA" rounds, not FLOPs.  Hence: BK = 32 (few rounds), the next
// tile's global loads are issued into registers BEFORE the current tile's MFMAs and written to
// the other LDS buffer after them (one barrier per round), and the long-K weight-gradient
// products are split over blockIdx.z with float-atomic accumulation into the zeroed gradient.
// Tile: 64x64x32 per 256-thread workgroup, 4 waves as 2x2, each wave 32x32 = 2x2 MFMA tiles.
// The LDS image of an operand follows its unit-stride dimension, so both the global read
// (128-B segments) and the ds_read_b32 fragment reads are conflict-free.
#include "step.h"

namespace gscan {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LD_K = BK + 2;    // image [row][k]: fragment read banks (2*row + k) mod 32 are distinct
constexpr int LD_R = BM + 16;   // image [k][row]: k rows 16 banks apart -> 32-lane halves conflict-free
constexpr int TILE_FLOATS = (BK * LD_R > BM * LD_K) ? BK * LD_R : BM * LD_K;

// Independent products are launched together as one grid ("grouped GEMM"): the step issues ~45 small
// products, each of which alone cannot fill 256 CUs and costs a launch; workgroup -> (problem, tile, k-slice)
// is a scan over at most kMaxGroup prefix sums held in kernel arguments.

// One [64 rows x 32 k] panel = 8 elements per thread.  element(row,k) = src[row*s_row + k*s_k].
// k_contig (k stride 1): threads run along k first, image [row][k]; else along rows, image [k][row].
__device__ __forceinline__ void panel_load(float (&v)[8], const float *src, int64_t s_row, int64_t s_k, int row0,
                                           int nrows, int k0, int kend, bool k_contig, int tid) {
    if (k_contig) {
        const int k = k0 + (tid & 31);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = row0 + (tid >> 5) + 8 * i;
            v[i] = (r < nrows && k < kend) ? src[(int64_t)r * s_row + (int64_t)k * s_k] : 0.f;
        }
    } else {
        const int r = row0 + (tid & 63);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + (tid >> 6) + 4 * i;
            v[i] = (r < nrows && k < kend) ? src[(int64_t)r * s_row + (int64_t)k * s_k] : 0.f;
        }
    }
}
__device__ __forceinline__ void panel_store(float *lds, const float (&v)[8], bool k_contig, int tid) {
    if (k_contig) {
#pragma unroll
        for (int i = 0; i < 8; ++i) lds[((tid >> 5) + 8 * i) * LD_K + (tid & 31)] = v[i];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) lds[((tid >> 6) + 4 * i) * LD_R + (tid & 63)] = v[i];
    }
}

__global__ __launch_bounds__(256) void gemm_group_kernel(GemmGroup grp) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if (i < grp.count && (int)blockIdx.x >= grp.p[i].tile_begin) pi = i;
    const GemmProblem &g = grp.p[pi];
    const int local = blockIdx.x - g.tile_begin;
    const int bz = local / g.tiles_mn, rem = local % g.tiles_mn;
    const int by = rem / g.tiles_n, bx = rem % g.tiles_n;
    __shared__ float lds_a[2][TILE_FLOATS];
    __shared__ float lds_b[2][TILE_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const bool a_kc = (g.sak == 1), b_kc = (g.sbk == 1);
    const int a_sr = a_kc ? LD_K : 1, a_sk = a_kc ? 1 : LD_R;
    const int b_sr = b_kc ? LD_K : 1, b_sk = b_kc ? 1 : LD_R;
    const bool do_asum = g.asum1 != nullptr && bx == 0;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;

    const int fr = lane & 15;   // fragment row (A) / column (B)
    const int fk = lane >> 4;   // fragment k within a 4-deep MFMA step

    float ra[8], rb[8];
    panel_load(ra, g.a, g.sam, g.sak, m0, g.M, kbeg, kend, a_kc, tid);
    panel_load(rb, g.b, g.sbn, g.sbk, n0, g.N, kbeg, kend, b_kc, tid);
    panel_store(lds_a[0], ra, a_kc, tid);
    panel_store(lds_b[0], rb, b_kc, tid);
    __syncthreads();

    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {   // next tile's loads fly while this tile's MFMAs run
            panel_load(ra, g.a, g.sam, g.sak, m0, g.M, k0 + BK, kend, a_kc, tid);
            panel_load(rb, g.b, g.sbn, g.sbk, n0, g.N, k0 + BK, kend, b_kc, tid);
        }
        const float *la = lds_a[buf], *lb = lds_b[buf];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = la[(wm * 32 + i * 16 + fr) * a_sr + (kk + fk) * a_sk];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = lb[(wn * 32 + j * 16 + fr) * b_sr + (kk + fk) * b_sk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (do_asum && tid < BM) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) asum += la[tid * a_sr + kk * a_sk];
        }
        if (more) {
            panel_store(lds_a[buf ^ 1], ra, a_kc, tid);
            panel_store(lds_b[buf ^ 1], rb, b_kc, tid);
        }
        __syncthreads();
        buf ^= 1;
    }

    if (do_asum && tid < BM && m0 + tid < g.M) {
        atomicAdd(&g.asum1[m0 + tid], asum);
        if (g.asum2) atomicAdd(&g.asum2[m0 + tid], asum);
    }

    // C/D fragment: column = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 32 + j * 16 + (lane & 15);
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (row >= g.M) continue;
                float *cp = g.c + (int64_t)row * g.ldc + col;
                float v = g.alpha * acc[i][j][r];
                if (g.atomic) {
                    atomicAdd(cp, v);
                } else {
                    if (g.beta != 0.f) v += g.beta * (*cp);
                    if (g.bias) v += g.bias[col];
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    else if (g.act == 2) v = tanhf_(v);
                    else if (g.act == 3 && g.gate[(int64_t)row * g.ldc + col] == 0.f) v = 0.f;   // ReLU backward
                    if (g.mask) v *= g.mask[(int64_t)row * g.ldc + col];
                    *cp = v;
                }
            }
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
    if (split_k < 1) split_k = 1;
    int chunk = cdiv(K, split_k);
    chunk = cdiv(chunk, BK) * BK;
    split_k = cdiv(K, chunk);
    if (split_k > 1 && !(beta == 1.f && act == 0 && !bias && !mask)) {
        bad_ = true;
        set_error("gemm batch: split-K needs beta=1 and no epilogue (problem %d, beta=%g act=%d)", grp_.count, beta, act);
        return;
    }
    GemmProblem &p = grp_.p[grp_.count++];
    p = GemmProblem{M, N, K, alpha, beta, a, sam, sak, b, sbk, sbn, c, ldc, bias, act, mask, gate, chunk,
                    split_k > 1 ? 1 : 0, asum1, asum2, cdiv(N, BN), cdiv(N, BN) * cdiv(M, BM), tiles_};
    tiles_ += p.tiles_mn * split_k;
    flops_ += 2.0 * M * N * K;
}

int GemmBatch::launch(hipStream_t stream) {
    if (bad_) return 1;
    if (grp_.count == 0) return 0;
    ProbeScope probe(P_GEMM, stream, flops_);
    hipLaunchKernelGGL(gemm_group_kernel, dim3(tiles_), dim3(256), 0, stream, grp_);
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

}  // namespace gscan
