// World encoder (seq2seq/cnn_model.py:22-36): three same-padded convolutions (kernels 1, 5, K3) over the
// [B, G, G, C] world tensor, concatenated along the channel dimension, ReLU, dropout — and their weight / bias
// gradients — as INPUT-SPARSE kernels.
//
// A gSCAN world is a grid of one-hot attribute vectors: one agent cell (2 ones) and up to a dozen objects (3 ones
// each) on 36 cells x 16 channels, i.e. <= ~40 non-zeros among 576 inputs (gym_minigrid/minigrid.py:380-399,
// gSCAN_dataset.py:229-231).  And the grid (6x6) is smaller than the kernels (7x7, 13x13), so the "convolution"
// is almost a fully connected layer: every non-zero input reaches most output cells.  Hence
//     out[b, q, f] = bias[f] + sum over the non-zeros (p, ch, v) of example b within reach of q of
//                    v * W_conv(f)[o(f), ch, kh = col(p) - col(q) + pad, kw = row(p) - row(q) + pad]
// costs ~80 k multiply-adds per example instead of the 1.21 M of the dense form (3.11 M as a Toeplitz product,
// which is what round 1 ran on the matrix cores, 61 % of it multiplying zeros), with no 12.4 MB weight image to
// rebuild and no 12.4 MB gradient image to fold.  The result is exact for ANY input (a zero contributes nothing to a
// sum), dense worlds just take proportionally longer.  kh walks grid columns and kw grid rows because the reference
// convolves the transposed image (cnn_model.py:28,34).  The world may arrive as float32 or as uint8 (the batcher
// ships 576 B per example and nothing widens it in HBM).
//
// Forward: four workgroups per example.  Each compacts the example's non-zeros into LDS with ballots (scan order,
// deterministic), then every wave walks (output cell, convolution) pairs with a lane per output channel: the
// 64 lanes test 64 non-zeros against the pair's kernel window at once, and the hits are consumed four at a time
// (v_readlane of value and weight-row offset, one coalesced load of a Co-float weight row each, one FMA).  The
// weight rows come from a [tap][ch][o] image of the three kernels written by the step prologue (60 k floats,
// L2-resident).  Bound: L2 -> CU traffic of the weight rows, ~400 KB per example.
//
// Backward: d W_conv[o, ch, kh, kw] = sum over b, non-zeros (p, ch, v) of v * dfeat[b, q(p, tap), conv, o].  A first
// small kernel compacts the non-zeros per (input channel, quarter of the batch); in the second a workgroup owns two
// kernel taps of one input channel, a wave per (tap, batch quarter), a lane per output channel: one Co-float row of
// dfeat per hit, the four quarter sums added in LDS, and the tap's gradients leave as plain adds into the reference's
// [Co, C, k, k] layout — no atomics, fixed summation order.  Extra workgroups add the bias gradients (column sums of
// dfeat over 64-row chunks, ONE float atomic per chunk and column: the only sums of this file whose order varies).
#include "step.h"
#include "prologue.h"
#include <atomic>
#include <chrono>

namespace gscan {

constexpr int kConvThreads = 512;
constexpr int kConvWaves = kConvThreads / 64;
#ifndef GSCAN_CONV_SEGMENTS
#define GSCAN_CONV_SEGMENTS 8
#endif
// The batch is cut into this many segments for the backward kernel's lists: a wave of the gradient kernel walks ONE segment's
// non-zeros of its (channel, tap), a chain of ~16-deep gather rounds.  Round 5: 8 (a workgroup = one tap x eight segments; it
// was 4 = two taps x four segments): the kernel ends the step together with the caller's chain, and half as
// long a chain per wave takes it from 46 to 41 us and the step from 0.4724 to 0.4707 ms (2: 50.6 us / 0.4772 ms;
// profiles/r05_conv_gradient_segments_ab.txt).
constexpr int kConvSegments = GSCAN_CONV_SEGMENTS;

__global__ void conv_image_kernel(const float *__restrict__ w1, const float *__restrict__ w2,
                                  const float *__restrict__ w3, int C, int Co, int K3, float *__restrict__ img) {
    const int total = (26 + K3 * K3) * C * conv_row_floats(Co);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
        img[i] = conv_image_element(w1, w2, w3, C, Co, K3, i);
}

int conv_weight_image(const float *const (&w)[3], int C, int Co, int K3, float *img, hipStream_t stream) {
    const int total = (int)conv_image_floats(C, Co, K3);
    hipLaunchKernelGGL(conv_image_kernel, dim3(std::min(cdiv(total, 256), 1024)), dim3(256), 0, stream, w[0], w[1], w[2],
                       C, Co, K3, img);
    GSCAN_LAUNCHED("conv_image_kernel");
    return 0;
}

struct ConvArgs {
    int B, G, C, Co, K3;
    const float *img;            // forward: [tap][ch][o]
    const float *b[3];           // forward: biases
    const float *mask;           // forward: [B, G*G, 3Co] scaled keep mask or NULL
    DropSpec drop;               // forward: drop.on — the dropout is drawn here (dropout.h), `mask` is not read
    float *feat;                 // forward: [B, G*G, 3Co]
    const float *dfeat;          // backward: [B, G*G, 3Co] gradient wrt the pre-activation
    float *gw[3], *gb[3];        // backward: gradients (added to)
    int slice, nw_blocks;       // backward: examples per list segment, workgroups that own weight gradients
    uint32_t *seg_keys;          // backward: per (channel, segment) lists of non-zeros [C][kConvSegments][slice * G*G]
    float *seg_vals;
    int *seg_count;              // [C][kConvSegments]
    float *bias_part;            // backward, fixed-order sums: per-chunk bias partials [chunks][3Co] (NULL: float atomics)
};

// Compaction of the non-zeros among n elements into LDS by the whole workgroup, in scan order (deterministic).
// get(e) -> value, key(e) -> packed coordinates.  Wave w takes a contiguous range of the elements; pass 1 counts its
// non-zeros, pass 2 (after the counts of the waves in front are known) writes them.  A pass fetches eight 64-element
// rows before it looks at any of them: the scan is a chain of global-load latencies otherwise (one wave alone over a
// 576-byte example: 9 dependent round trips to HBM, most of the kernel).  Returns the count; ends with a barrier.
template <typename Get, typename Key>
__device__ __forceinline__ int compact_nonzeros(int n, uint32_t *keys, float *vals, int *wave_counts, Get get, Key key) {
    constexpr int U = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (((n + kConvWaves - 1) / kConvWaves) + 63) / 64 * 64;
    const int lo = wave * per, hi = min(n, lo + per);
    int cnt = 0;
    const bool single = per <= 64 * U;                      // a wave's whole range in one pass (uniform in the workgroup)
    float first[U];
#pragma unroll
    for (int u = 0; u < U; ++u) first[u] = 0.f;
    for (int base = lo; base < hi; base += 64 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * 64 + lane;
            v[u] = e < hi ? get(e) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cnt += __popcll(__ballot(v[u] != 0.f));
        if (single) {
#pragma unroll
            for (int u = 0; u < U; ++u) first[u] = v[u];
        }
    }
    if (lane == 0) wave_counts[wave] = cnt;
    __syncthreads();
    int at = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kConvWaves; ++w) {
        const int c = wave_counts[w];
        at += w < wave ? c : 0;
        total += c;
    }
    for (int base = lo; base < hi; base += 64 * U) {
        float v[U];
        if (single) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = first[u];    // pass 1's loads, kept across the barrier: no second round trip
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + u * 64 + lane;
                v[u] = e < hi ? get(e) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool nz = v[u] != 0.f;
            const unsigned long long m = __ballot(nz);
            if (nz) {
                const int i = at + __popcll(m & ((1ull << lane) - 1ull));
                keys[i] = key(base + u * 64 + lane);
                vals[i] = v[u];
            }
            at += __popcll(m);
        }
    }
    __syncthreads();
    return total;
}

// acc += sum over the set lanes j of `hits` of val_j * src[row_j + o].  A hit costs one load of a short row and nothing
// is reused, so what matters is how many loads a wave keeps in flight: R hits per round (a missing hit repeats
// the round's first address with a zero value), value and row offset fetched from the hit lanes with v_readlane.
template <int R>
__device__ __forceinline__ float consume_hits(unsigned long long hits, int row, float v, const float *__restrict__ src,
                                              int o) {
    float acc[R];
#pragma unroll
    for (int u = 0; u < R; ++u) acc[u] = 0.f;
    while (hits) {
        int j[R];
        bool has[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            has[u] = hits != 0;
            j[u] = has[u] ? __builtin_ctzll(hits) : j[0];
            hits = has[u] ? (hits & (hits - 1)) : hits;
        }
        float w[R], x[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int r = __builtin_amdgcn_readlane(row, j[u]);
            x[u] = has[u] ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j[u])) : 0.f;
            w[u] = src[r + o];
        }
#pragma unroll
        for (int u = 0; u < R; ++u) acc[u] = fmaf(x[u], w[u], acc[u]);
    }
#pragma unroll
    for (int w = R / 2; w > 0; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    return acc[0];
}

#ifndef GSCAN_CONV_FWD_HITS
#define GSCAN_CONV_FWD_HITS 8
#endif
constexpr int kFwdHits = GSCAN_CONV_FWD_HITS;     // weight-row gathers a wave keeps in flight
// The forward pass of one workgroup: example b, share y of ny of its (output cell, convolution) pairs.  flags != NULL
// (the launch that also runs the step prologue, below): the weight image is being written by other workgroups of the
// SAME launch; the workgroup compacts its non-zeros first — that needs the world only — and then waits until every
// image workgroup has published `epoch` in its flag.
// Chunk `chunk` (kImageElems elements) of the [tap][ch][o] weight image, written THROUGH (agent-scope stores: a plain
// store would sit dirty in the writer's L2, and the release that pushes it out is a write-back of that whole L2 —
// tools/micro/flag_wait.hip: 57 us against 10).  All threads of the workgroup take part.
constexpr int kImageElems = 4 * kConvThreads;
struct FusedImageArgs { const float *w[3]; float *img; };
__device__ __forceinline__ void fused_image_chunk(const FusedImageArgs &ia, const ConvArgs &a, int chunk) {
    const int total = (26 + a.K3 * a.K3) * a.C * conv_row_floats(a.Co);
#pragma unroll
    for (int j = 0; j < kImageElems / kConvThreads; ++j) {
        const int i = chunk * kImageElems + j * kConvThreads + (int)threadIdx.x;
        if (i < total)
            __hip_atomic_store(ia.img + i, conv_image_element(ia.w[0], ia.w[1], ia.w[2], a.C, a.Co, a.K3, i),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, bool DRAWN = false>
__device__ __forceinline__ void world_conv_fwd_body(const ConvArgs &a, const T *__restrict__ world, int b, int y, int ny,
                                                    uint32_t *flags, int nflags, uint32_t epoch,
                                                    const FusedImageArgs &img_args = FusedImageArgs{}, int late_after = 0,
                                                    bool acquire = false) {
    extern __shared__ uint32_t conv_lds[];
    const int G = a.G, C = a.C, Co = a.Co, M = G * G, F = 3 * Co, MC = M * C;
    uint32_t *keys = conv_lds;
    float *vals = reinterpret_cast<float *>(conv_lds + MC);
    int *wave_counts = reinterpret_cast<int *>(conv_lds + 2 * MC);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T *x = world + (int64_t)b * MC;
    const int count = compact_nonzeros(
        MC, keys, vals, wave_counts, [&](int e) { return (float)x[e]; },
        [&](int e) { const int p = e / C, ch = e - p * C, pr = p / G; return (uint32_t)(pr | ((p - pr * G) << 8) | (ch << 16)); });
    if (flags) {
        // Thread i polls image workgroup i's flag (agent-scope loads: served past this XCD's L2).  The wait is BOUNDED and
        // needs no dispatch order: a workgroup that has not seen a flag after kFlagPolls polls writes that chunk of
        // the image ITSELF (the values are a pure function of the parameters: a second writer stores the same
        // bytes) and publishes the flag, so no resident workgroup ever depends on one that is not running yet.
        // In dispatch order (image workgroups first, what this chip does) the self-service path never runs; the
        // count of self-served chunks is kept in the word behind the flags (gscan_fused_prologue_selfserved).
        bool late = false;
        if ((int)threadIdx.x < nflags) {
            int polls = 0;
            while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch &&
                   polls < late_after) {
                __builtin_amdgcn_s_sleep(1);
                ++polls;
            }
            late = polls >= late_after;
        }
        if (__syncthreads_or(late)) {
            for (int i = 0; i < nflags; ++i) {
                const int missing = __syncthreads_or(
                    threadIdx.x == 0 && __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch);
                if (!missing) continue;
                fused_image_chunk(img_args, a, i);
                __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the written-through stores have landed
                __syncthreads();
                if (threadIdx.x == 0) {
                    __hip_atomic_store(flags + i, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicAdd(flags + kFusedMaxFlags, 1u);
                }
            }
        }
        // The image is read with ordinary loads behind ONE agent-scope acquire by one lane (invalidates this CU's L1;
        // the wait holds the barrier until the invalidate has completed): cdna_hip_programming.md Guideline 16's
        // consumer form.  (Without it — GSCAN_FUSED_ACQUIRE=0 — the reads rely on no cache holding a line of the
        // image when they are issued: caches are invalidated at the launch boundary and nothing in this launch reads
        // the image before the flags are up.)
        if (acquire && threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    // an example's (output cell, convolution) pairs are dealt to ny workgroups x 8 waves
    const int ochunks = (Co + 63) >> 6, npairs = M * 3 * ochunks, CoP = conv_row_floats(Co);
    auto one_pair = [&](int pair, float keep, bool use_keep) {
        const int q = pair / (3 * ochunks), rem = pair - q * 3 * ochunks, conv = rem / ochunks, oc = rem - conv * ochunks;
        const int o = oc * 64 + lane, oo = min(o, Co - 1);
        const int k = conv_ksize(conv, a.K3), h = k >> 1, tap0 = conv_tap0(conv, a.K3);
        const int qr = q / G, qc = q - qr * G;
        float acc = 0.f;
        for (int c0 = 0; c0 < count; c0 += 64) {
            const int i = c0 + lane;
            const bool live = i < count;
            const uint32_t key = live ? keys[i] : 0u;
            const float v = live ? vals[i] : 0.f;
            const int kw = (int)(key & 255u) - qr + h, kh = (int)((key >> 8) & 255u) - qc + h, ch = (int)(key >> 16);
            const bool hit = live && (unsigned)kw < (unsigned)k && (unsigned)kh < (unsigned)k;
            const int row = ((tap0 + kh * k + kw) * C + ch) * CoP;
            acc += consume_hits<kFwdHits>(__ballot(hit), row, v, a.img, oo);
        }
        if (o < Co) {
            const int f = conv * Co + o;
            const int64_t at = ((int64_t)b * M + q) * F + f;
            float val = fmaxf(acc + a.b[conv][o], 0.f);
            if (use_keep) val *= keep;
            else if (a.mask) val *= a.mask[at];
            a.feat[at] = val;
        }
    };
    // Dealing.  Masks in memory (or none): pair = first, first + stride, ...  Dropout drawn here (a.drop.on, dropout.h): a
    // wave takes FOUR consecutive pairs at a time and one Philox call per lane serves them — word j of counter
    // (b, group, lane) is the keep value of channel `lane` of pair 4 group + j.  (Strided pairs would need a call each;
    // the longest wave has four pairs either way.)  One call site for both: the pair's code exists once.
    const bool drawn = DRAWN && a.drop.on != 0;
    const int stride = ny * kConvWaves, first = y * kConvWaves + wave, ngroups = (npairs + 3) >> 2;
    uint32_t keep_bits = 0;          // one register across the group's four pairs (the fused launch runs on 63 VGPRs)
#pragma unroll 1
    for (int it = 0;; ++it) {
        int pair;
        if (drawn) {
            const int grp = first + (it >> 2) * stride;
            if (grp >= ngroups) break;
            if ((it & 3) == 0) keep_bits = drop_quad_bits(a.drop, kDropSegCnn, ((uint64_t)b * ngroups + grp) * 64 + lane);
            pair = 4 * grp + (it & 3);
            if (pair >= npairs) continue;
        } else {
            pair = first + it * stride;
            if (pair >= npairs) break;
        }
        one_pair(pair, ((keep_bits >> (it & 3)) & 1u) ? a.drop.scale : 0.f, drawn);
    }
}

template <typename T, bool DRAWN = false>
__global__ __launch_bounds__(kConvThreads) void world_conv_fwd_kernel(ConvArgs a, const T *__restrict__ world) {
    TraceScope trace_scope(TK_CONV_FWD);
    world_conv_fwd_body<T, DRAWN>(a, world, blockIdx.x, blockIdx.y, gridDim.y, nullptr, 0, 0u);
}

// The step prologue and the world encoder in ONE launch (step.hip's default prelude).  The two have nothing in common
// but one dependency — the encoder reads the [tap][ch][o] image of the convolution weights that the prologue writes —
// and as two launches they cost the chain a kernel boundary (6-7 us: the release of the prologue's 8 MB of dirty lines,
// the dispatch ramp) plus the time either one leaves most of the chip idle (both are chains of L2 round trips).
// Workgroup roles by block index (the chip dispatches in index order, which makes the waits short; CORRECTNESS does not
// depend on it: a waiting workgroup that runs out of patience writes the missing chunk itself, world_conv_fwd_body):
//   [0, n_img)            the weight image, kImageElems elements per workgroup (fused_image_chunk); when the stores
//                         are acknowledged the workgroup publishes the launch's epoch in its flag
//   [n_img, n_img+n_pro)  every other prologue segment, grid-stride over its index space
//   the rest              the world encoder's workgroups (example b, share y): they wait for the flags behind their
//                         own compaction pass (world_conv_fwd_body)
// The epoch is a process-wide counter seeded from the clock: flags need no reset, whatever the workspace held before.
struct FusedPrologueArgs {
    int n_img, n_pro, n_conv, n_examples, ny, order;
    uint32_t *flags;            // n_img flags, then (at kFusedMaxFlags) the count of self-served chunks
    uint32_t epoch;
    int late_after;             // polls before a waiting workgroup writes a missing chunk itself
    int acquire;                // agent-scope acquire between the flags and the first read of the image
    int skip_image;             // test hook: the image workgroups do nothing (every chunk is then self-served)
};
template <typename T, bool DRAWN = false>
__global__ __launch_bounds__(kConvThreads, 8) void prologue_world_kernel(PrologueArgs pa, ConvArgs a, FusedPrologueArgs f,
                                                                         const T *__restrict__ world) {
    TraceScope trace_scope(TK_PROLOGUE);
    const int blk = blockIdx.x;
    const FusedImageArgs ia{{pa.conv_w[0], pa.conv_w[1], pa.conv_w[2]}, pa.conv_img};
    if (blk < f.n_img) {
        if (f.skip_image) return;
        fused_image_chunk(ia, a, blk);
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this thread's written-through stores have landed
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(f.flags + blk, f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // order of the two bulk roles behind the image workgroups: 0 prologue first, 1 world encoder first, 2 interleaved
    // in proportion (workgroup j of n is the world encoder's when the scaled count j n_conv / n steps up)
    const int j = blk - f.n_img, n = f.n_pro + f.n_conv;
    int cb, pb;
    bool is_conv;
    if (f.order == 0) { is_conv = j >= f.n_pro; cb = j - f.n_pro; pb = j; }
    else if (f.order == 1) { is_conv = j < f.n_conv; cb = j; pb = j - f.n_conv; }
    else {
        const int c0 = (int)(((int64_t)j * f.n_conv) / n), c1 = (int)(((int64_t)(j + 1) * f.n_conv) / n);
        is_conv = c1 > c0; cb = c0; pb = j - c0;
    }
    if (!is_conv) {
        const int64_t total = pa.end[13];
        for (int64_t idx = (int64_t)pb * kConvThreads + threadIdx.x; idx < total; idx += (int64_t)f.n_pro * kConvThreads)
            prologue_element<8, DRAWN>(pa, idx);
        return;
    }
    world_conv_fwd_body<T, DRAWN>(a, world, cb % f.n_examples, cb / f.n_examples, f.ny, f.flags, f.n_img, f.epoch, ia,
                                  f.late_after, f.acquire != 0);
}

// Backward, pass 1: the non-zeros of input channel ch among the examples of batch segment s, compacted in scan
// order into seg_keys / seg_vals [ch][s][...] (key = cell row | cell column << 8 | example << 16).
template <typename T>
__global__ __launch_bounds__(kConvThreads) void world_channel_lists_kernel(ConvArgs a, const T *__restrict__ world) {
    TraceScope trace_scope(TK_CONV_BWD);
    __shared__ int wave_counts[kConvWaves];
    const int G = a.G, C = a.C, M = G * G;
    const int ch = blockIdx.x, sg = blockIdx.y, b0 = sg * a.slice, nb = max(0, min(a.B, b0 + a.slice) - b0);
    const int64_t base = ((int64_t)ch * kConvSegments + sg) * a.slice * M;
    const T *x = world + (int64_t)b0 * M * C + ch;
    const int count = compact_nonzeros(
        nb * M, a.seg_keys + base, a.seg_vals + base, wave_counts, [&](int e) { return (float)x[(int64_t)e * C]; },
        [&](int e) { const int bl = e / M, p = e - bl * M, pr = p / G; return (uint32_t)(pr | ((p - pr * G) << 8) | ((b0 + bl) << 16)); });
    if (threadIdx.x == 0) a.seg_count[ch * kConvSegments + sg] = count;
}

// Backward, pass 2.  Workgroup (ch, y): its 8 waves are 2 kernel taps (x output-channel chunk) times the 4 batch
// segments; wave (tap, segment) walks that segment's list of channel ch in chunks of 64 (a lane per non-zero tests it
// against the tap), consumes the hits 16 row loads at a time with a lane per output channel, and the four segment
// sums of a tap are added in LDS in a fixed order: the workgroup OWNS the tap's Co gradients, which leave as plain adds.
#ifndef GSCAN_CONV_BWD_HITS
#define GSCAN_CONV_BWD_HITS 16
#endif
constexpr int kBwdHits = GSCAN_CONV_BWD_HITS;     // row gathers a wave keeps in flight
template <int DUMMY>
__global__ __launch_bounds__(kConvThreads) void world_conv_bwd_kernel(ConvArgs a) {
    TraceScope trace_scope(TK_CONV_BWD);
    __shared__ float partial[kConvWaves][64];
    const int G = a.G, C = a.C, Co = a.Co, M = G * G, F = 3 * Co;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 1-D grid: nw_blocks weight workgroups (channel = id % C, tap group = id / C), then the bias chunks
    if ((int)blockIdx.x >= a.nw_blocks) {
        // bias gradients: db_i[o] += sum over rows of dfeat[row, f] for a chunk of 64 rows of the [B*G*G, F] view
        const int chunk = blockIdx.x - a.nw_blocks, rows = a.B * M;
        const int r0 = chunk * 64, r1 = min(rows, r0 + 64);
        // thread = (column, one of kSl slices of the chunk's rows): sixteen loads in flight per pass (four per pass made
        // the 64 rows sixteen dependent round trips to L2 — the longest chain of the whole launch), slices added in LDS
        constexpr int kSl = 2;
        __shared__ float bsl[kSl][kConvThreads / kSl];
        const int fl = threadIdx.x % (kConvThreads / kSl), sl = threadIdx.x / (kConvThreads / kSl);
        const int per = (r1 - r0 + kSl - 1) / kSl, q0 = r0 + sl * per, q1 = min(r1, q0 + per);
        // F = 3 Co columns in groups of kConvThreads / kSl (more than one group from Co = 86 on: the first version of this
        // code summed the first 256 columns only — found by tools/fuzz_parity.py --extremes at Co = 200)
        for (int f0 = 0; f0 < F; f0 += kConvThreads / kSl) {
            const int f = f0 + fl;
            float acc = 0.f;
            if (f < F) {
                for (int r = q0; r < q1; r += 16) {
                    float x[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) x[u] = (r + u < q1) ? a.dfeat[(int64_t)(r + u) * F + f] : 0.f;
#pragma unroll
                    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
                        for (int u = 0; u < w; ++u) x[u] += x[u + w];
                    acc += x[0];
                }
            }
            bsl[sl][fl] = acc;
            __syncthreads();
            if (sl == 0 && f < F) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < kSl; ++q) s += bsl[q][fl];
                if (a.bias_part) a.bias_part[(int64_t)chunk * F + f] = s;      // added in chunk order by conv_bias_reduce_kernel
                else atomicAdd(&a.gb[f / Co][f % Co], s);
            }
            __syncthreads();
        }
        return;
    }
    constexpr int kTapsPerGroup = kConvWaves / kConvSegments;
    // Heaviest channels FIRST in dispatch order (round 5): the grid is more workgroups than the chip holds at once (1 200 for
    // 1 024 slots at the benchmark shape), a workgroup's time is its channel's count of non-zeros (420-610 for the attribute
    // channels of a gSCAN world, ~60 for the four direction channels), and with the channels interleaved (channel = block % C)
    // the second generation was a sixth of every channel — heavy ones included.  Now block / groups is a RANK by count
    // (from the lists' own counts: C x segments integers, one wave), so what is left for the second generation is the lightest.
    __shared__ int ch_of_rank;
    const int ngroups = a.nw_blocks / C, rank = blockIdx.x / ngroups, by = blockIdx.x - rank * ngroups, sg = wave % kConvSegments;
    if (wave == 0) {
        int mine = -1;                                     // lane c < C: total of channel c (C <= 64; more channels: identity order)
        if (C <= 64 && lane < C) {
            mine = 0;
            for (int q = 0; q < kConvSegments; ++q) mine += a.seg_count[lane * kConvSegments + q];
        }
        int r = 0;
        for (int c = 0; c < min(C, 64); ++c) {
            const int other = __builtin_amdgcn_readlane(mine, c);
            r += (other > mine || (other == mine && c < lane)) ? 1 : 0;
        }
        if (C > 64) { if (lane == 0) ch_of_rank = rank; }
        else if (lane < C && r == rank) ch_of_rank = lane;
    }
    __syncthreads();
    const int ch = ch_of_rank;
    const int ochunks = (Co + 63) >> 6, npairs = (26 + a.K3 * a.K3) * ochunks;
    const int pair = by * kTapsPerGroup + wave / kConvSegments;
    const bool owns = pair < npairs;
    const int tg = owns ? pair / ochunks : 0, oc = owns ? pair - tg * ochunks : 0;
    const int o = oc * 64 + lane, oo = min(o, Co - 1);
    const int conv = tg < 1 ? 0 : (tg < 26 ? 1 : 2), k = conv_ksize(conv, a.K3), h = k >> 1;
    const int t = tg - conv_tap0(conv, a.K3), kh = t / k, kw = t - kh * k;
    float acc = 0.f;
    if (owns) {
        const int64_t base = ((int64_t)ch * kConvSegments + sg) * a.slice * M;
        const int count = a.seg_count[ch * kConvSegments + sg];
        // the non-zero at (pr, pc) reaches output cell (pr - (kw - h), pc - (kh - h)) through this tap
        for (int c0 = 0; c0 < count; c0 += 64) {
            const int i = c0 + lane;
            const bool live = i < count;
            const uint32_t key = live ? a.seg_keys[base + i] : 0u;
            const float v = live ? a.seg_vals[base + i] : 0.f;
            const int qr = (int)(key & 255u) - (kw - h), qc = (int)((key >> 8) & 255u) - (kh - h), b = (int)(key >> 16);
            const bool hit = live && (unsigned)qr < (unsigned)G && (unsigned)qc < (unsigned)G;
            const int row = (b * M + qr * G + qc) * F + conv * Co;
            acc += consume_hits<kBwdHits>(__ballot(hit), row, v, a.dfeat, oo);
        }
    }
    partial[wave][lane] = acc;
    __syncthreads();
    if (owns && sg == 0 && o < Co) {
        const int w0 = wave;          // waves w0 .. w0 + kConvSegments - 1 hold the segment sums of this tap: added in a fixed order
        float sum = 0.f;
        if (kConvSegments == 4) {
            sum = (partial[w0][lane] + partial[w0 + 1][lane]) + (partial[w0 + 2][lane] + partial[w0 + 3][lane]);
        } else {
#pragma unroll
            for (int q = 0; q < kConvSegments; ++q) sum += partial[w0 + q][lane];
        }
        a.gw[conv][(o * C + ch) * k * k + t] += sum;
    }
}

// MAC_conv = C * Co * sum over the three kernels of valid(G, k)^2, valid = taps that hit real cells
static double conv_algorithmic_flops(int B, int G, int C, int Co, int K3) {
    double mac = 0.0;
    for (int k : {1, 5, K3}) {
        int valid = 0;
        for (int i = 0; i < G; ++i)
            for (int j = 0; j < G; ++j) valid += (i - j <= k / 2 && j - i <= k / 2);
        mac += (double)valid * valid;
    }
    return 2.0 * B * C * Co * mac;
}

static int conv_check(int B, int G, int C, int Co, int K3) {
    GSCAN_CHECK(B > 0 && G > 0 && G <= 255 && C > 0 && C <= 255 && Co > 0 && K3 > 0 && (K3 & 1),
                "world encoder: unsupported dimensions (B=%d G=%d C=%d Co=%d K3=%d)", B, G, C, Co, K3);
    GSCAN_CHECK((int64_t)B * G * G * 3 * Co < (1ll << 31) && (int64_t)(26 + K3 * K3) * C * Co < (1ll << 31),
                "world encoder: batch or kernels too large for 32-bit offsets (B=%d G=%d Co=%d K3=%d)", B, G, Co, K3);
    return 0;
}

int world_conv_forward(const void *world, int world_is_u8, const float *img, const float *const (&b)[3],
                       const float *mask, int B, int G, int C, int Co, int K3, float *feat, hipStream_t stream,
                       const DropSpec *drop) {
    TRY_RC(conv_check(B, G, C, Co, K3));
    const size_t lds = (size_t)G * G * C * 8 + 4 * kConvWaves;
    GSCAN_CHECK(lds <= 128 * 1024, "world encoder: a %dx%dx%d world does not fit the non-zero list in LDS", G, G, C);
    ConvArgs a{};
    a.B = B; a.G = G; a.C = C; a.Co = Co; a.K3 = K3; a.img = img; a.mask = mask; a.feat = feat;
    if (drop) a.drop = *drop;
    for (int i = 0; i < 3; ++i) a.b[i] = b[i];
    static bool attr_set = false;
    if (!attr_set) {
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&world_conv_fwd_kernel<float, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&world_conv_fwd_kernel<uint8_t, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&world_conv_fwd_kernel<float, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&world_conv_fwd_kernel<uint8_t, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    ProbeScope probe(P_CONV_FWD, stream, 0.0, conv_algorithmic_flops(B, G, C, Co, K3));
    // four workgroups per example: 32 waves per CU keep ~256 row loads in flight, and the examples' unequal numbers
    // of non-zeros (5 ... 38) spread over the chip instead of one CU carrying the heaviest example alone
    const dim3 grid(B, 4);
    const bool drawn = a.drop.on != 0;                      // dropout drawn in the kernel: its own instantiation
    if (world_is_u8) {
        if (drawn) hipLaunchKernelGGL((world_conv_fwd_kernel<uint8_t, true>), grid, dim3(kConvThreads), lds, stream, a, static_cast<const uint8_t *>(world));
        else hipLaunchKernelGGL((world_conv_fwd_kernel<uint8_t, false>), grid, dim3(kConvThreads), lds, stream, a, static_cast<const uint8_t *>(world));
    } else {
        if (drawn) hipLaunchKernelGGL((world_conv_fwd_kernel<float, true>), grid, dim3(kConvThreads), lds, stream, a, static_cast<const float *>(world));
        else hipLaunchKernelGGL((world_conv_fwd_kernel<float, false>), grid, dim3(kConvThreads), lds, stream, a, static_cast<const float *>(world));
    }
    GSCAN_LAUNCHED("world_conv_fwd_kernel");
    return 0;
}

// Prologue + world encoder as one launch (prologue_world_kernel).  `pa` carries every prologue segment but the
// convolution image (its segment must be empty: the image workgroups of this launch write it); flags = kFusedMaxFlags
// words of the workspace.  Returns -1 (nothing launched) when the shape does not fit the fused form.
int prologue_world_forward(const PrologueArgs &pa, const void *world, int world_is_u8, const float *const (&b)[3],
                           const float *mask, int B, int G, int C, int Co, int K3, float *feat, uint32_t *flags,
                           hipStream_t stream, const DropSpec *drop) {
    TRY_RC(conv_check(B, G, C, Co, K3));
    const size_t lds = (size_t)G * G * C * 8 + 4 * kConvWaves;
    const int n_img = cdiv(conv_image_floats(C, Co, K3), kImageElems);
    if (lds > 128 * 1024 || n_img > kFusedMaxFlags || pa.end[10] != pa.end[9]) return -1;
    ConvArgs a{};
    a.B = B; a.G = G; a.C = C; a.Co = Co; a.K3 = K3; a.img = pa.conv_img; a.mask = mask; a.feat = feat;
    if (drop) a.drop = *drop;
    for (int i = 0; i < 3; ++i) a.b[i] = b[i];
    static bool attr_set = false;
    if (!attr_set) {
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&prologue_world_kernel<float, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&prologue_world_kernel<uint8_t, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&prologue_world_kernel<float, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        GSCAN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&prologue_world_kernel<uint8_t, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    // a value no earlier launch of this process used and, seeded from the clock, none a previous process is likely to
    // have left in the same memory
    static std::atomic<uint32_t> epoch{(uint32_t)std::chrono::steady_clock::now().time_since_epoch().count() | 1u};
    // Both roles want the chip's wave slots to themselves (the world encoder's 4 B workgroups ARE 32 waves per CU at the
    // benchmark batch), so they overlap little; what the one launch saves is the boundary.  Measured on the step
    // (profiles/r03_e_fused_prologue_ab.txt): prologue first, 1024 workgroups (one generation, three passes per thread)
    // 0.5017 ms; 2048: 0.5051; 4096: 0.5057; world encoder first 0.507-0.511; interleaved 0.509-0.521; two launches 0.5092
    static const int pro_cap = [] { const char *e = getenv("GSCAN_PROLOGUE_BLOCKS"); return e ? atoi(e) : 1024; }();
    FusedPrologueArgs f{};
    f.n_img = n_img;
    f.n_pro = (int)std::min<int64_t>(cdiv(pa.end[13], kConvThreads), pro_cap);
    f.n_examples = B; f.ny = 4; f.n_conv = B * f.ny;
    static const int order = [] { const char *e = getenv("GSCAN_FUSED_ORDER"); return e ? atoi(e) : 0; }();
    f.order = order;
    f.flags = flags;
    f.epoch = epoch.fetch_add(2u);
    // ~100 cycles per poll: 2 000 polls are some 80 us, ten times the longest wait seen in dispatch order
    static const int late_after = [] { const char *e = getenv("GSCAN_FUSED_LATE_AFTER"); return e ? atoi(e) : 2000; }();
    // acquire between the flags and the image reads: ON (cdna_hip_programming.md Guideline 16's consumer form).  It costs
    // 1-4 us of the ~5 the one launch saves (profiles/r04_fused_prologue_acquire_ab.txt: 0.4785 ms per step without,
    // 0.4810 with, 0.4823 as two launches); GSCAN_FUSED_ACQUIRE=0 for A/B runs.
    static const int acquire = [] { const char *e = getenv("GSCAN_FUSED_ACQUIRE"); return e ? atoi(e) : 1; }();
    static const int skip = [] { const char *e = getenv("GSCAN_FUSED_SKIP_IMAGE"); return e ? atoi(e) : 0; }();
    f.late_after = late_after; f.acquire = acquire; f.skip_image = skip;
    ProbeScope probe(P_CONV_FWD, stream, 0.0, conv_algorithmic_flops(B, G, C, Co, K3));
    const dim3 grid(f.n_img + f.n_pro + B * f.ny);
    const bool drawn = a.drop.on || pa.drop_enc.on || pa.drop_dec.on;     // any dropout drawn in this launch: its own instantiation
    if (world_is_u8) {
        if (drawn) hipLaunchKernelGGL((prologue_world_kernel<uint8_t, true>), grid, dim3(kConvThreads), lds, stream, pa, a, f, static_cast<const uint8_t *>(world));
        else hipLaunchKernelGGL((prologue_world_kernel<uint8_t, false>), grid, dim3(kConvThreads), lds, stream, pa, a, f, static_cast<const uint8_t *>(world));
    } else {
        if (drawn) hipLaunchKernelGGL((prologue_world_kernel<float, true>), grid, dim3(kConvThreads), lds, stream, pa, a, f, static_cast<const float *>(world));
        else hipLaunchKernelGGL((prologue_world_kernel<float, false>), grid, dim3(kConvThreads), lds, stream, pa, a, f, static_cast<const float *>(world));
    }
    GSCAN_LAUNCHED("prologue_world_kernel");
    return 0;
}

size_t world_conv_backward_scratch_floats(int B, int G, int C) {
    const size_t seg = (size_t)cdiv(B, kConvSegments) * G * G;
    return 2 * (size_t)C * kConvSegments * seg + (size_t)C * kConvSegments + 64;
}

static ConvArgs conv_backward_args(int B, int G, int C, int Co, int K3, float *scratch) {
    ConvArgs a{};
    a.B = B; a.G = G; a.C = C; a.Co = Co; a.K3 = K3;
    a.slice = cdiv(B, kConvSegments);
    a.nw_blocks = C;
    const size_t seg = (size_t)a.slice * G * G;
    a.seg_keys = reinterpret_cast<uint32_t *>(scratch);
    a.seg_vals = scratch + (size_t)C * kConvSegments * seg;
    a.seg_count = reinterpret_cast<int *>(scratch + 2 * (size_t)C * kConvSegments * seg);
    return a;
}

// Pass 1 of the backward: the per-(channel, batch quarter) lists of non-zeros.  It needs the world only, so the step
// launches it long before the gradient wrt the features exists.
int world_conv_lists(const void *world, int world_is_u8, int B, int G, int C, float *scratch, hipStream_t stream) {
    GSCAN_CHECK(B > 0 && B < 65536 && G > 0 && G <= 255 && C > 0 && C <= 255 && world && scratch,
                "world encoder lists: unsupported arguments (B=%d G=%d C=%d)", B, G, C);
    const ConvArgs a = conv_backward_args(B, G, C, 1, 1, scratch);
    const dim3 lists(C, kConvSegments);
    if (world_is_u8)
        hipLaunchKernelGGL(world_channel_lists_kernel<uint8_t>, lists, dim3(kConvThreads), 0, stream, a,
                           static_cast<const uint8_t *>(world));
    else
        hipLaunchKernelGGL(world_channel_lists_kernel<float>, lists, dim3(kConvThreads), 0, stream, a,
                           static_cast<const float *>(world));
    GSCAN_LAUNCHED("world_channel_lists_kernel");
    return 0;
}

// fixed-order form of the bias gradients: thread f adds the chunks' partial sums of column f in chunk order
__global__ void conv_bias_reduce_kernel(const float *__restrict__ part, int chunks, int F, int Co, float *gb0, float *gb1,
                                        float *gb2) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float s = ordered_chunk_sum(part, chunks, F, f);
    float *gb = f / Co == 0 ? gb0 : (f / Co == 1 ? gb1 : gb2);
    gb[f % Co] += s;
}

size_t world_conv_bias_partial_floats(int B, int G, int Co) { return (size_t)cdiv((int64_t)B * G * G, 64) * 3 * Co; }

// Pass 2: kernel and bias gradients from the lists of world_conv_lists and d(features).  bias_part (optional,
// world_conv_bias_partial_floats): the bias gradients are then added in a fixed order (a second, tiny launch) instead
// of with float atomics — the deterministic mode of the training step.
int world_conv_backward(const float *dfeat, int B, int G, int C, int Co, int K3, float *scratch,
                        float *const (&gw)[3], float *const (&gb)[3], hipStream_t stream, float *bias_part) {
    TRY_RC(conv_check(B, G, C, Co, K3));
    GSCAN_CHECK(B < 65536, "world encoder: more than 65535 examples per call (B=%d)", B);
    ConvArgs a = conv_backward_args(B, G, C, Co, K3, scratch);
    a.dfeat = dfeat;
    for (int i = 0; i < 3; ++i) { a.gw[i] = gw[i]; a.gb[i] = gb[i]; }
    const int npairs = (26 + K3 * K3) * cdiv(Co, 64);
    a.nw_blocks = C * cdiv(npairs, kConvWaves / kConvSegments);
    const dim3 grid(a.nw_blocks + cdiv((int64_t)B * G * G, 64));
    ProbeScope probe(P_CONV_BWD, stream, 0.0, conv_algorithmic_flops(B, G, C, Co, K3));
    a.bias_part = bias_part;
    hipLaunchKernelGGL(world_conv_bwd_kernel<0>, grid, dim3(kConvThreads), 0, stream, a);
    GSCAN_LAUNCHED("world_conv_bwd_kernel");
    if (bias_part) {
        hipLaunchKernelGGL(conv_bias_reduce_kernel, dim3(cdiv(3 * Co, 64)), dim3(64), 0, stream, bias_part,
                           cdiv((int64_t)B * G * G, 64), 3 * Co, Co, gb[0], gb[1], gb[2]);
        GSCAN_LAUNCHED("conv_bias_reduce_kernel");
    }
    return 0;
}

GSCAN_TRACE_TU(conv)

}  // namespace gscan
