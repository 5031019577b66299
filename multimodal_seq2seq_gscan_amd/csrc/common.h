// Shared host/device helpers for libgscan_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <algorithm>

namespace gscan {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);

#define GSCAN_CHECK(cond, ...)                                                      \
    do {                                                                            \
        if (!(cond)) { ::gscan::set_error(__VA_ARGS__); return 1; }                 \
    } while (0)

#ifndef GSCAN_PLAN_ONLY
#define GSCAN_HIP(call)                                                             \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            ::gscan::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                               \
        }                                                                           \
    } while (0)
#else
// GSCAN_PLAN_ONLY: the HOST side of every source file, compiled without device code and with
// -fsanitize=address,undefined (tests/planner/: build script, driver and shape lists; SURVEY.md 5 "sanitizers").  The
// sequencing, planning and argument-checking code of the library runs on the CPU exactly as shipped — workspace_layout,
// check_dims, pick_split, GemmBatch::add / launch / launch_macro_tiles, the decoder's LDS budgets, every C-ABI argument
// check — while each device call succeeds without doing anything and each kernel launch is checked against the limits of
// the hardware (plan_record: non-empty grid, <= 1024 threads, <= 160 KB of LDS) and counted instead of launched.
#define GSCAN_HIP(call) do { } while (0)
void plan_record(const char *kernel, dim3 grid, dim3 block, size_t lds_bytes);
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) ::gscan::plan_record(#kernel, grid, block, (size_t)(lds))
#endif

// GSCAN_ROCTX=1: every launch of the library leaves a named roctx marker (probe.hip; rocprofv3 --marker-trace shows
// them next to the kernel rows, so the step's seventeen launches read as what they are); otherwise a flag test.
void roctx_mark(const char *name);

// Launch check: a bad configuration surfaces at launch, not at the next sync.
#define GSCAN_LAUNCHED(name)                                                        \
    do {                                                                            \
        ::gscan::roctx_mark(name);                                                  \
        hipError_t e_ = ::gscan::launch_status();                                   \
        if (e_ != hipSuccess) {                                                     \
            ::gscan::set_error("launch of %s failed: %s", name, hipGetErrorString(e_)); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

#ifndef GSCAN_PLAN_ONLY
static inline hipError_t launch_status() { return hipGetLastError(); }
#else
static inline hipError_t launch_status() { return hipSuccess; }
#endif

#define TRY_RC(expr) do { if (int rc_ = (expr)) return rc_; } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Kernel ids of the in-kernel timeline (diagnostic, tools/device_timeline.py).
enum TraceKernel { TK_DROPOUT = 1, TK_CONV_FWD, TK_PROLOGUE, TK_GEMM, TK_ENCODER_FWD, TK_DECODER_FWD, TK_DECODER_BWD,
                   TK_KEYS_BWD, TK_UNPERMUTE, TK_EMBED_GRAD, TK_ENCODER_BWD, TK_CONV_BWD, TK_ADAM, TK_LOSS, TK_OTHER };
constexpr int kTraceRecords = 256;     // per list; buffer = 2 counters + 2 lists of kTraceRecords x (kid, grid, ticks)

#if defined(__HIPCC__)
// In-kernel timeline without a profiler in the way, compiled in with -DGSCAN_TRACE only (tools/device_timeline.py
// builds that variant): when a trace buffer is set (gscan_trace_set), the first workgroup
// of every launch appends (kernel id, grid size, s_memrealtime) to the start list and the LAST workgroup (highest block
// index: dispatched last) appends the same to the end list when its thread 0 leaves the kernel.  The 100 MHz
// constant clock is shared by the whole device, so the records of kernels on different streams line up.  The pointer is a per-translation-unit device variable (no
// relocatable device code): every .hip file with kernels defines its setter with GSCAN_TRACE_TU.
static __device__ unsigned long long *g_trace_buf;
struct TraceScope {
    int kid;
    __device__ __forceinline__ void put(int list) const {
        unsigned long long *t = g_trace_buf;
        const unsigned long long i = atomicAdd(&t[list], 1ull);
        if (i < (unsigned long long)kTraceRecords) {
            unsigned long long *r = t + 2 + (size_t)list * 3 * kTraceRecords + 3 * i;
            r[0] = (unsigned long long)kid;
            r[1] = (unsigned long long)gridDim.x * gridDim.y;
            r[2] = wall_clock64();
        }
    }
#ifndef GSCAN_TRACE         // the shipped build: no code (the two tests per workgroup cost the step 1.4 %)
    __device__ __forceinline__ explicit TraceScope(int k) : kid(k) {}
#else
    __device__ __forceinline__ explicit TraceScope(int k) : kid(k) {
        if (g_trace_buf && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) put(0);
    }
    __device__ __forceinline__ ~TraceScope() {
        if (g_trace_buf && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && threadIdx.x == 0) put(1);
    }
#endif
};
#define GSCAN_TRACE_TU(name)                                                                        \
    int trace_set_##name(unsigned long long *buf) {                                                 \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1; \
    }

// sum_{c < chunks} part[c * stride + idx] in chunk order, sixteen loads in flight (the fixed-order second phase of the
// deterministic mode's reductions: one thread per output element)
__device__ __forceinline__ float ordered_chunk_sum(const float *__restrict__ part, int chunks, int64_t stride, int64_t idx) {
    float s = 0.f;
    for (int c0 = 0; c0 < chunks; c0 += 16) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = (c0 + u < chunks) ? part[(int64_t)(c0 + u) * stride + idx] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += x[u];
    }
    return s;
}

// sigmoid / tanh through v_exp_f32 (2^x) and v_rcp_f32 (1 ulp each): absolute error ~1e-7, two transcendental
// issues and two (sigmoid) / three (tanh) plain VALU operations per call; both saturate cleanly for large |x|
// (2^(+big) = inf, rcp(inf) = 0).  tanh x = 2 sigmoid(2x) - 1: the recurrent kernels are bound by VALU issue, and
// the (1 - e) / (1 + e) form with |x| and copysign costs three more instructions per call for a relative accuracy
// near 0 that nothing downstream needs.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

__device__ __forceinline__ float tanhf_(float x) {
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x));
    return fmaf(2.0f, r, -1.0f);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)), i.e. every barrier would wait for the global STORES of saved activations issued in
// that phase (hundreds of cycles each, several times per time step in the recurrent kernels).  Data that
// crosses threads inside those kernels goes through LDS, so lgkmcnt(0) + s_barrier is sufficient; values
// loaded from global memory are still guarded by the compiler's own waits at their first use.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Wave-wide (64-lane) reductions on the DPP datapath instead of ds_bpermute shuffles: four in-row steps
// (quad_perm, quad_perm, row_ror:4, row_ror:8) leave each 16-lane row holding its own total, row_bcast:15 /
// row_bcast:31 fold the rows into lane 63, and v_readlane broadcasts it.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xb1, 0xf>(v);
    v += dpp_move<0x4e, 0xf>(v);
    v += dpp_move<0x124, 0xf>(v);
    v += dpp_move<0x128, 0xf>(v);
    v += dpp_move<0x142, 0xa>(v);
    v += dpp_move<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// G independent sums, step by step: the DPP chain of one sum is ~10 dependent instructions with wait states
// between them, G chains issued round-robin fill each other's gaps.
template <int G>
__device__ __forceinline__ void wave_sum_n(float (&v)[G]) {
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] += dpp_move<0xb1, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] += dpp_move<0x4e, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] += dpp_move<0x124, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] += dpp_move<0x128, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] += dpp_move<0x142, 0xa>(v[i]);
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] += dpp_move<0x143, 0xc>(v[i]);
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[i]), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xb1, 0xf>(v));
    v = fmaxf(v, dpp_move<0x4e, 0xf>(v));
    v = fmaxf(v, dpp_move<0x124, 0xf>(v));
    v = fmaxf(v, dpp_move<0x128, 0xf>(v));
    // rows masked out of a row_bcast step receive 0 and may be wrong for negative inputs; lane 63's row
    // takes part in both steps, and lane 63 is the only lane read back
    v = fmaxf(v, dpp_move<0x142, 0xa>(v));
    v = fmaxf(v, dpp_move<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#endif

}  // namespace gscan
