// Strided fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, one
// rounding per product, so results track an fp32 fmaf chain).
//
//   C[M,N] = act(alpha * A.B + beta * C + bias[n]) * mask[m,n]
//
// Every dense contraction of the training step that is not inside a time loop goes through
// this kernel: the im2col convolutions, key/value projections, LSTM input projections, the
// output head, and all weight-gradient products (K = B*T rows, split over workgroups).
// Operands are addressed with (row, col) strides so the reference's [out,in] parameter
// layout and its transposes are consumed in place; nothing is re-packed in HBM.
//
// Tile: 64x64x16 per 256-thread workgroup, 4 waves as 2x2, each wave 32x32 = 2x2 MFMA tiles.
// LDS image per operand is chosen from the operand's unit-stride dimension so both the
// global read (64-B segments) and the ds_read_b32 fragment reads are (near) conflict-free.
#include "step.h"

namespace gscan {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LD_CONTIG_K = BK + 1;   // image [row][k], k contiguous, padded
constexpr int LD_CONTIG_R = BM + 16;  // image [k][row], row contiguous, stride = 16 mod 32 banks
constexpr int TILE_FLOATS = (BK * LD_CONTIG_R > BM * LD_CONTIG_K) ? BK * LD_CONTIG_R : BM * LD_CONTIG_K;

struct GemmArgs {
    int M, N, K;
    float alpha, beta;
    const float *a; int64_t sam, sak;
    const float *b; int64_t sbk, sbn;
    float *c; int64_t ldc;
    const float *bias; int act; const float *mask;
    int k_chunk;   // K range per blockIdx.z
    int atomic;    // split-K: accumulate with atomics
};

// Stage a [64 rows x 16 k] panel.  element(row,k) = src[row*s_row + k*s_k].
// k_contig: the k stride is 1 -> threads run along k first (coalesced), image [row][k];
// otherwise threads run along rows first, image [k][row].
__device__ __forceinline__ void stage_panel(float *lds, const float *src, int64_t s_row, int64_t s_k,
                                            int row0, int nrows, int k0, int kend, bool k_contig, int tid) {
    if (k_contig) {
        const int k = tid & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 4) + 16 * i;
            float v = 0.f;
            if (row0 + r < nrows && k0 + k < kend) v = src[(int64_t)(row0 + r) * s_row + (int64_t)(k0 + k) * s_k];
            lds[r * LD_CONTIG_K + k] = v;
        }
    } else {
        const int r = tid & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = (tid >> 6) + 4 * i;
            float v = 0.f;
            if (row0 + r < nrows && k0 + k < kend) v = src[(int64_t)(row0 + r) * s_row + (int64_t)(k0 + k) * s_k];
            lds[k * LD_CONTIG_R + r] = v;
        }
    }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ float lds_a[TILE_FLOATS];
    __shared__ float lds_b[TILE_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const bool a_kc = (g.sak == 1), b_kc = (g.sbk == 1);
    const int a_sr = a_kc ? LD_CONTIG_K : 1, a_sk = a_kc ? 1 : LD_CONTIG_R;
    const int b_sr = b_kc ? LD_CONTIG_K : 1, b_sk = b_kc ? 1 : LD_CONTIG_R;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;   // fragment row (A) / column (B)
    const int fk = lane >> 4;   // fragment k within a 4-deep MFMA step

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        stage_panel(lds_a, g.a, g.sam, g.sak, m0, g.M, k0, kend, a_kc, tid);
        stage_panel(lds_b, g.b, g.sbn, g.sbk, n0, g.N, k0, kend, b_kc, tid);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = lds_a[(wm * 32 + i * 16 + fr) * a_sr + (kk + fk) * a_sk];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = lds_b[(wn * 32 + j * 16 + fr) * b_sr + (kk + fk) * b_sk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
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
                    if (g.mask) v *= g.mask[(int64_t)row * g.ldc + col];
                    *cp = v;
                }
            }
        }
}

int gemm_f32(int M, int N, int K, float alpha, const float *a, int64_t sam, int64_t sak, const float *b,
             int64_t sbk, int64_t sbn, float beta, float *c, int64_t ldc, const float *bias, int act,
             const float *mask, int split_k, hipStream_t stream) {
    GSCAN_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem %dx%dx%d", M, N, K);
    GSCAN_CHECK(a && b && c, "gemm: null operand");
    GSCAN_CHECK(act >= 0 && act <= 2, "gemm: unknown activation %d", act);
    if (split_k < 1) split_k = 1;
    int chunk = cdiv(K, split_k);
    chunk = cdiv(chunk, BK) * BK;
    split_k = cdiv(K, chunk);
    if (split_k > 1)
        GSCAN_CHECK(beta == 1.f && act == 0 && !bias && !mask,
                    "gemm: split-K needs beta=1 and no epilogue (got beta=%g act=%d)", beta, act);
    GemmArgs g{M, N, K, alpha, beta, a, sam, sak, b, sbk, sbn, c, ldc, bias, act, mask, chunk, split_k > 1 ? 1 : 0};
    dim3 grid(cdiv(N, BN), cdiv(M, BM), split_k);
    ProbeScope probe(P_GEMM, stream, 2.0 * M * N * K);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, stream, g);
    GSCAN_LAUNCHED("gemm_f32_kernel");
    return 0;
}

}  // namespace gscan
