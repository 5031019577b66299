// Dropout drawn INSIDE the kernels that consume it (SURVEY.md 7 hard part 3; nn.Dropout of cnn_model.py:34,
// seq2seq_model.py:59 and :384).  Round 5: the production step no longer materialises its three masks — 7.8 MB written by
// the optimiser launch and read back by five kernels at the benchmark shape — but evaluates a counter-based generator
// (Philox-4x32-10: key = seed, counter = [index | segment | stream id]) where the masked value is produced and where its
// gradient is needed.  A call yields FOUR 32-bit words; the index -> (counter, word) maps below are chosen so that a
// consumer thread uses all four:
//   cnn   features [B, M, 3Co]   a lane of the world encoder = one output channel of an (output cell, convolution) PAIR;
//                                pairs are dealt to waves four at a time: counter = (b * NG + pair / 4) * 64 + lane,
//                                word = pair % 4  (pair = (cell * 3 + conv) * ceil(Co / 64) + channel / 64).  The backward
//                                pass does not need the mask at all: feat = relu(conv) * mask is non-zero only where the
//                                mask is 1 / (1 - p), so d conv = (feat != 0) ? d feat / (1 - p) : 0.
//   enc / dec embeddings [rows, D]   counter = (row / 4) * D + column, word = row % 4: a thread of the gather (and of the
//                                embedding gradient) handles one column of four consecutive rows.
// gscan_dropout_masks_kernel_layout() writes the same masks to memory (tests, host-mask parity mode).
#pragma once
#include <stdint.h>

namespace gscan {

struct DropSpec {
    uint32_t on;                 // 0: no in-kernel dropout (the mask POINTER of the consumer decides, as before)
    uint32_t k0, k1, s0, s1;     // seed (the Philox key) and stream id (counter words 2, 3)
    float p, scale;              // drop probability, 1 / (1 - p)
};

enum { kDropSegCnn = 0, kDropSegEnc = 1, kDropSegDec = 2 };

inline DropSpec drop_spec(bool on, uint64_t seed, uint64_t stream_id, float p) {
    DropSpec s{};
    s.on = (on && p > 0.f) ? 1u : 0u;
    s.k0 = (uint32_t)seed; s.k1 = (uint32_t)(seed >> 32);
    s.s0 = (uint32_t)stream_id; s.s1 = (uint32_t)(stream_id >> 32);
    s.p = p; s.scale = 1.0f / (1.0f - p);
    return s;
}

#if defined(__HIPCC__)
__device__ __forceinline__ void philox_round4(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// the four scaled keep values (0 or 1 / (1 - p)) of counter `ctr` of segment `seg`
__device__ __forceinline__ void drop_quad(const DropSpec &s, int seg, uint64_t ctr, float (&m)[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32) | ((uint32_t)seg << 24), c2 = s.s0, c3 = s.s1;
    uint32_t k0 = s.k0, k1 = s.k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round4(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const uint32_t w[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = ((float)(w[j] >> 8) * (1.0f / 16777216.0f) >= s.p) ? s.scale : 0.f;   // u in [0, 1)
}
// a copy of `s` whose fields are (re)defined HERE for the compiler: inside a grid-stride loop the loop-invariant pieces of a
// Philox call are otherwise hoisted out of the loop and held in registers across every other branch of it
__device__ __forceinline__ DropSpec drop_spec_here(const DropSpec &s) {
    DropSpec r = s;
    asm volatile("" : "+s"(r.k0), "+s"(r.k1), "+s"(r.s0), "+s"(r.s1), "+s"(r.p), "+s"(r.scale));
    return r;
}
// the same four decisions as bits (bit j: keep), for consumers that hold them across a long loop
__device__ __forceinline__ uint32_t drop_quad_bits(const DropSpec &s, int seg, uint64_t ctr) {
    float m[4];
    drop_quad(s, seg, ctr, m);
    return (m[0] != 0.f ? 1u : 0u) | (m[1] != 0.f ? 2u : 0u) | (m[2] != 0.f ? 4u : 0u) | (m[3] != 0.f ? 8u : 0u);
}
#endif

// cnn segment: groups of four (cell, convolution, 64-channel chunk) pairs per example
__host__ __device__ inline int drop_cnn_pairs(int M, int Co) { return M * 3 * ((Co + 63) >> 6); }
__host__ __device__ inline int drop_cnn_groups(int M, int Co) { return (drop_cnn_pairs(M, Co) + 3) >> 2; }

}  // namespace gscan
