// Optional per-kernel-family timing with HIP events on the launch stream (bench.py's roofline
// numbers).  Off by default; when on, each probed launch is bracketed by two hipEventRecord
// calls on its own stream and the elapsed times are summed at read time.  Not usable while
// the stream is being captured into a graph.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "step.h"

namespace gscan {

static const char *kProbeNames[P_COUNT] = {"decoder_forward", "decoder_backward", "encoder_forward",
                                           "encoder_backward", "gemm", "conv_forward", "conv_backward",
                                           "keys_backward"};
struct ProbeState {
    std::vector<hipEvent_t> begin, end;
    size_t used = 0;
    double flops = 0.0, alg_flops = 0.0;
};
static bool g_probe_on = false;
static bool g_stamps_on = false;
static ProbeState g_probe[P_COUNT];
constexpr size_t kMaxPairs = 1 << 15;

ProbeScope::ProbeScope(int id, hipStream_t st, double flops, double alg_flops) : id_(-1), st_(st) {
    if (!g_probe_on) return;
    ProbeState &p = g_probe[id];
    if (p.used >= kMaxPairs) return;
    if (p.used == p.begin.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        p.begin.push_back(a);
        p.end.push_back(b);
    }
    id_ = id;
    p.flops += flops;
    p.alg_flops += alg_flops < 0.0 ? flops : alg_flops;
    hipEventRecord(p.begin[p.used], st);
}
ProbeScope::~ProbeScope() {
    if (id_ < 0) return;
    ProbeState &p = g_probe[id_];
    hipEventRecord(p.end[p.used], st_);
    ++p.used;
}

// Named markers for profilers (tracing): roctxMarkA from the roctx library, bound at run time and only
// when GSCAN_ROCTX=1 (no link dependency; the ranges ProbeScope would give bracket the ENQUEUE of asynchronous
// launches, which says nothing a marker does not).
void roctx_mark(const char *name) {
    using MarkFn = void (*)(const char *);
    static const MarkFn mark = []() -> MarkFn {
        const char *e = getenv("GSCAN_ROCTX");
        if (!e || atoi(e) == 0) return nullptr;
        for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})
            if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL))
                if (void *f = dlsym(h, "roctxMarkA")) return (MarkFn)f;
        return nullptr;
    }();
    if (mark) mark(name);
}

bool probe_stamps_enabled() { return g_stamps_on; }

// bit 0: HIP-event probes around kernel families; bit 1: in-kernel phase stamps of the decoder kernels
int probe_enable(int on) {
    g_probe_on = (on & 1) != 0;
    g_stamps_on = (on & 2) != 0;
    return 0;
}
int probe_reset() {
    for (auto &p : g_probe) { p.used = 0; p.flops = 0.0; p.alg_flops = 0.0; }
    return 0;
}
int probe_read(const char *name, double *total_ms, double *flops, double *alg_flops, int64_t *launches) {
    for (int i = 0; i < P_COUNT; ++i) {
        if (strcmp(name, kProbeNames[i]) != 0) continue;
        ProbeState &p = g_probe[i];
        double ms = 0.0;
        for (size_t k = 0; k < p.used; ++k) {
            GSCAN_HIP(hipEventSynchronize(p.end[k]));
            float t = 0.f;
            GSCAN_HIP(hipEventElapsedTime(&t, p.begin[k], p.end[k]));
            ms += t;
        }
        *total_ms = ms;
        *flops = p.flops;
        *alg_flops = p.alg_flops;
        *launches = (int64_t)p.used;
        return 0;
    }
    GSCAN_CHECK(false, "probe_read: unknown kernel family '%s'", name);
}

}  // namespace gscan
