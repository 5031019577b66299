// Shared by gemm.hip and gemm_mt.hip: tile constants and the global -> registers -> LDS panel mover.
#pragma once
#include "step.h"

namespace gscan {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
// Operand pointers arrive inside a by-value descriptor struct, which hides from the compiler that they point to global
// memory: every access became a FLAT instruction (counts on lgkmcnt as well as vmcnt, so the wait in front of a
// round's LDS fragment reads also waited for the NEXT round's panel loads).  Casting to address space 1 gives
// global_load / global_store / global_atomic.
#define GSCAN_GLOBAL __attribute__((address_space(1)))
using gfloat = GSCAN_GLOBAL float;
__device__ __forceinline__ const gfloat *as_global(const float *p) { return (const gfloat *)p; }
__device__ __forceinline__ gfloat *as_global(float *p) { return (gfloat *)p; }

constexpr int TNW = 2;           // MFMA tiles per wave along N; along M it is the kernel's template parameter TMW
constexpr int BN = 2 * 16 * TNW;
// Depth of a K round (template parameter BK of the kernel): 32; 64 is compiled for experiments (see launch()).
template <int BK> struct TileK { static constexpr int LDK = BK + 4; };      // k-contiguous image [row][LDK]: 16-byte rows, b128 fragment reads 2-way at worst
constexpr int LDR_B = BN + 4;    // row-contiguous image [k][LDR]: b64 reads of 2 adjacent columns, conflict-free
template <int BK> constexpr int b_floats() { return (BN * (BK + 4) > BK * LDR_B) ? BN * (BK + 4) : BK * LDR_B; }
// Workgroup tile = (32 TMW) x 64 x 32.  TMW = 2 (64 rows) is the throughput shape; TMW = 1 (32 rows) doubles the
// number of workgroups of a launch whose 64-row tiling would leave CUs with one or two resident workgroups and
// nothing to hide a K round's load latency behind (most launches of the training step).
template <int TMW, int BK = 32> struct TileM {
    static constexpr int BM = 2 * 16 * TMW;
    static constexpr int LDR_A = BM + 4;   // row-contiguous image [k][LDR]
    static constexpr int A_FLOATS = (BM * (BK + 4) > BK * LDR_A) ? BM * (BK + 4) : BK * LDR_A;
};

// Independent products are launched together as one grid ("grouped GEMM"): the step issues ~45 small
// products, each of which alone cannot fill 256 CUs and costs a launch; workgroup -> (problem, tile, k-slice)
// is a scan over at most kMaxGroup prefix sums held in kernel arguments.
//
// Tile geometry.  Workgroup (32 TMW) x 64 x 32, 4 waves as 2 x 2, each wave (16 TMW) x 32 = TMW x 2 MFMA tiles
// of 16 x 16 (many small tiles: these products are latency-bound, occupancy hides more than a bigger tile saves).
// Within a 32-deep tile the MFMA k index of lane group g (= lane >> 4) at step s (0..7) is k = 8 g + s, so a lane
// of a k-contiguous operand reads its 8 values with two ds_read_b128.  A row-contiguous operand interleaves its
// MFMA tiles instead (row = 2 i + tile for A, col = 2 i + tile for B), so one ds_read_b64 per k feeds both
// of a wave's tiles.  Either way a wave issues ~1 LDS read per 4-8 MFMAs instead of 1 per MFMA.
// Global loads are 16-byte whenever the operand's pointer, strides and extent allow, else 8- or 4-byte.


// How one operand's [ROWS x 32] panels move global -> registers -> LDS.  A thread's loads of one K round form an
// arithmetic progression (same k / stepping rows for a k-contiguous operand, same rows / stepping k for a
// row-contiguous one), so everything about them is computed ONCE: the K loop pays one compare and one add per
// load, no multiplies, no 64-bit arithmetic (measured before this: issuing a round's loads cost as many cycles
// as its MFMAs).  vw = floats per load (4 / 2 / 1: what the operand's alignment allows).
// KCT / VWT: layout known at compile time (1 / 0 = k-contiguous or not, 4 / 2 / 1 floats per load) or -1 / 0 = read
// from the problem at run time.  With compile-time layouts the K loop is straight-line code and the compiler's
// s_waitcnt insertion counts outstanding loads exactly; with run-time branches around the loads it falls back to
// vmcnt(0) at the joins, which serialises a round's A and B loads.
// NT: threads of the workgroup that share a panel's loads
template <int ROWS, int BK, int KCT = -1, int VWT = 0, int NT = 256>
struct PanelIter {
    static_assert(BK % 32 == 0, "a K round is one or more 32-deep halves (fragment maps of the kernel)");
    static constexpr int LDK = BK + 4;
    static constexpr int N = ROWS * BK / NT;     // floats per thread per K round
    const gfloat *src;
    uint32_t off;       // element offset of load 0 of the current round
    uint32_t istep;     // offset step between a thread's loads of one round
    uint32_t kinc;      // offset step per K round
    int klim;           // k-contiguous: every load reads while k0 < klim; row-contiguous: load i while k0 + kp*i < klim
    int nlive;          // k-contiguous: loads i < nlive touch rows inside the matrix
    int kp;             // row-contiguous: k rows between a thread's loads
    int vw_rt;
    bool kc_rt;
    __device__ __forceinline__ bool is_kc() const { return KCT >= 0 ? (KCT != 0) : kc_rt; }
    __device__ __forceinline__ int width() const { return VWT > 0 ? VWT : vw_rt; }

    __device__ __forceinline__ void init(const float *base, int64_t s_row, int64_t s_k, int nrows, int row0, int vw_,
                                         int kbeg, int kend, int tid) {
        src = as_global(base); vw_rt = vw_; kc_rt = (s_k == 1);
        const int vw = width();
        const uint32_t sr = (uint32_t)s_row, sk = (uint32_t)s_k;
        if (is_kc()) {
            const int ch = BK / vw, rp = NT / ch;                 // chunks per row, rows per pass
            const int r = row0 + tid / ch, kk = vw * (tid % ch);
            off = (uint32_t)r * sr + (uint32_t)(kbeg + kk);
            istep = (uint32_t)rp * sr;
            kinc = BK;
            klim = kend - kk;
            nlive = r < nrows ? (nrows - r + rp - 1) / rp : 0;
            kp = 0;
        } else {
            const int rq = ROWS / vw;                              // loads per k row
            const int r = row0 + vw * (tid % rq), kk = tid / rq;
            kp = NT / rq;
            off = (uint32_t)(kbeg + kk) * sk + (uint32_t)r * sr;
            istep = (uint32_t)kp * sk;
            kinc = (uint32_t)BK * sk;
            klim = r < nrows ? kend - kk : INT_MIN;
            nlive = N;
        }
    }

    // Loads are UNCONDITIONAL (a dead load reads element 0 of the operand instead) and their registers are not
    // written before them; dead values are zeroed when they are consumed (store()).  The obvious form — zero the
    // registers, then load under a lane mask — makes the compiler put s_waitcnt vmcnt(0) in front of every load
    // (write-after-write on the registers of the previous round's load, counted conservatively across the
    // branches), which serialised the A and B panel loads of a round and exposed a full load latency per round.
    template <int VW>
    __device__ __forceinline__ uint32_t load_vw(float (&v)[N], int k0) const {
        constexpr int NL = N / VW;
        uint32_t mask = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const bool live = is_kc() ? (i < nlive && k0 < klim) : (k0 + kp * i < klim);
            mask |= (live ? 1u : 0u) << i;
            const gfloat *ptr = src + (live ? off + (uint32_t)i * istep : 0u);
            if constexpr (VW == 4) {
                const f32x4 x = *reinterpret_cast<const GSCAN_GLOBAL f32x4 *>(ptr);
                v[4 * i] = x[0]; v[4 * i + 1] = x[1]; v[4 * i + 2] = x[2]; v[4 * i + 3] = x[3];
            } else if constexpr (VW == 2) {
                const f32x2 x = *reinterpret_cast<const GSCAN_GLOBAL f32x2 *>(ptr);
                v[2 * i] = x[0]; v[2 * i + 1] = x[1];
            } else {
                v[i] = *ptr;
            }
        }
        return mask;
    }
    // loads of the round starting at k0 (returns the mask of live loads), then step to the next round
    __device__ __forceinline__ uint32_t load(float (&v)[N], int k0) {
        uint32_t mask;
        const int vw = width();
        if (vw == 4) mask = load_vw<4>(v, k0);
        else if (vw == 2) mask = load_vw<2>(v, k0);
        else mask = load_vw<1>(v, k0);
        off += kinc;
        return mask;
    }

    // registers -> LDS, row-contiguous operand, image rows PERMUTED: k row kl = 8 g + s of the round is stored as row
    // 4 s + g, so the four lane groups of a fragment read (same s, g = 0..3) touch adjacent rows; with LDR = 16 mod 32
    // they fall into two disjoint sets of 16 banks each: two lanes per bank, the best 64 lanes can do on 32 banks
    // (natural rows 8 g + s put all four groups on the same 16 banks).  32-deep rounds only.
    template <int LDR>
    __device__ __forceinline__ void store_rows_permuted(float *lds, const float (&v)[N], uint32_t mask, int tid) const {
        static_assert(BK == 32, "the row permutation is written for 32-deep rounds");
        auto val = [&](int load, int e) { return ((mask >> load) & 1u) ? v[e] : 0.f; };
        const int vw = width();
        const int rq = ROWS / vw, k0 = tid / rq;
        float *dst = lds + vw * (tid % rq);
        auto prow = [&](int i) { const int kl = k0 + i * kp; return 4 * (kl & 7) + (kl >> 3); };
        if (vw == 4) {
#pragma unroll
            for (int i = 0; i < N / 4; ++i)
                *reinterpret_cast<float4 *>(dst + prow(i) * LDR) =
                    float4{val(i, 4 * i), val(i, 4 * i + 1), val(i, 4 * i + 2), val(i, 4 * i + 3)};
        } else if (vw == 2) {
#pragma unroll
            for (int i = 0; i < N / 2; ++i)
                *reinterpret_cast<float2 *>(dst + prow(i) * LDR) = float2{val(i, 2 * i), val(i, 2 * i + 1)};
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) dst[prow(i) * LDR] = val(i, i);
        }
    }

    // registers -> LDS image: k-contiguous [row][LDK], row-contiguous [k][LDR]
    template <int LDR>
    __device__ __forceinline__ void store(float *lds, const float (&v)[N], uint32_t mask, int tid) const {
        auto val = [&](int load, int e) { return ((mask >> load) & 1u) ? v[e] : 0.f; };
        const int vw = width();
        if (is_kc()) {
            const int ch = BK / vw, rp = NT / ch;
            float *dst = lds + (tid / ch) * LDK + vw * (tid % ch);
            if (vw == 4) {
#pragma unroll
                for (int i = 0; i < N / 4; ++i)
                    *reinterpret_cast<float4 *>(dst + i * rp * LDK) =
                        float4{val(i, 4 * i), val(i, 4 * i + 1), val(i, 4 * i + 2), val(i, 4 * i + 3)};
            } else if (vw == 2) {
#pragma unroll
                for (int i = 0; i < N / 2; ++i)
                    *reinterpret_cast<float2 *>(dst + i * rp * LDK) = float2{val(i, 2 * i), val(i, 2 * i + 1)};
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) dst[i * rp * LDK] = val(i, i);
            }
        } else {
            const int rq = ROWS / vw;
            float *dst = lds + (tid / rq) * LDR + vw * (tid % rq);
            if (vw == 4) {
#pragma unroll
                for (int i = 0; i < N / 4; ++i)
                    *reinterpret_cast<float4 *>(dst + i * kp * LDR) =
                        float4{val(i, 4 * i), val(i, 4 * i + 1), val(i, 4 * i + 2), val(i, 4 * i + 3)};
            } else if (vw == 2) {
#pragma unroll
                for (int i = 0; i < N / 2; ++i)
                    *reinterpret_cast<float2 *>(dst + i * kp * LDR) = float2{val(i, 2 * i), val(i, 2 * i + 1)};
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) dst[i * kp * LDR] = val(i, i);
            }
        }
    }
};


#ifdef GSCAN_GEMM_STAMPS   // experiment build: cycle stamps of one workgroup's life, in the tail of the trace buffer
#define GST(i) { const long long n_ = clock64(); gst_acc[i] += (unsigned)(n_ - gst_prev); gst_prev = n_; }
#else
#define GST(i)
#endif

// gemm_ws.hip: weights-stationary persistent kernel for launches of k-contiguous, unsplit, beta = 0 products with K <= 160
bool gemm_ws_eligible(const GemmProblem &p);
int gemm_ws_launch(const GemmGroup &grp, hipStream_t stream);

// gemm_mt.hip
bool gemm_mt_supports(const GemmProblem &p);
void gemm_mt_columns(int NV, int *tiles_n, int *nf);
size_t gemm_mt_slab_floats();
int gemm_mt_launch(const GemmGroup &grp, int total, int bm, hipStream_t stream);
int gemm_mt_reduce_launch(const GemmGroup &grp, const int (&unit_begin)[kMaxGroup], int total_units, int bm,
                          hipStream_t stream);

}  // namespace gscan
