// Microbenchmark: what a kernel boundary on ONE stream costs as a function of how the producer's stores were issued.
// Every XCD of gfx950 has its own L2, so the release at the end of a kernel writes the dirty lines of all eight L2s
// back before the next kernel of the stream may start (profiles/r02_micro_cross_stream_latency.txt: 1.0 us behind a
// kernel that wrote nothing, 5.5 us behind one that dirtied 32 MB).  Stores that are written THROUGH while the kernel
// runs leave nothing to write back at its end.  Variants of the same 16-byte store:
//   0 plain   1 nt   2 sc1   3 sc0 sc1   4 sc0 sc1 nt   5 sc0
// Reported per variant and size: producer duration (first workgroup's start -> last workgroup's exit), gap from that
// exit to the start of a dependent kernel on the same stream, and the duration of a dependent kernel that READS the data.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_policy_gap.hip -o gpurun_out/micro/spg && gpurun_out/micro/spg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#include <cstdint>

using f4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__device__ __forceinline__ void store16(f4 *p, f4 v) {
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}

// stamps[0] = max exit time, stamps[2] = min start time (initialised to ~0ull); `work` FMAs between stores make the
// kernel last tens of microseconds, as the step's kernels do, so written-through stores have time to drain
template <int MODE>
__global__ void producer(unsigned long long *stamps, f4 *buf, size_t n4, int work) {
    if (threadIdx.x == 0) stamps[8 + blockIdx.x] = wall_clock64();         // per-workgroup slots: no contended atomics
    float a = threadIdx.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        for (int k = 0; k < work; ++k) a = fmaf(a, 1.0001f, 0.5f);
        store16<MODE>(buf + i, f4{(float)i, a, 1.f, 2.f});
    }
    __syncthreads();
    if (threadIdx.x == 0) stamps[8 + 2048 + blockIdx.x] = wall_clock64();
}
// keeps the stream busy while the host enqueues the three kernels behind it (their launch latency is not the subject)
__global__ void spin(int ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) {}
}
__global__ void consumer(unsigned long long *stamps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = wall_clock64();
}
__global__ void reader(unsigned long long *stamps, const f4 *buf, size_t n4, float *sink) {
    if (threadIdx.x == 0) stamps[8 + 4096 + blockIdx.x] = wall_clock64();
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += buf[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0) stamps[8 + 6144 + blockIdx.x] = wall_clock64();
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
void launch_producer(hipStream_t s, unsigned long long *stamps, f4 *buf, size_t n4, int work) {
    hipLaunchKernelGGL(producer<MODE>, dim3(2048), dim3(256), 0, s, stamps, buf, n4, work);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *stamps;
    CK(hipMalloc(&stamps, 8 * (8 + 8192)));
    std::vector<unsigned long long> h(8 + 8192);
    float *sink;
    CK(hipMalloc(&sink, 4));
    const size_t max_bytes = 128u << 20;
    f4 *big;
    CK(hipMalloc(&big, max_bytes));
    const char *names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0"};
    const int reps = 20;
    printf("%-12s %8s %6s | %10s %8s %10s\n", "stores", "MB", "work", "producer", "gap", "reader");
    for (int work : {0, 40})
        for (size_t mb : {2, 8, 32, 64}) {
            const size_t n4 = (mb << 20) / 16;
            for (int mode = 0; mode < 6; ++mode) {
                std::vector<double> gap, dur, rd;
                for (int r = 0; r < reps; ++r) {
                    unsigned long long init[8] = {0, 0, ~0ull, ~0ull, 0, 0, 0, 0};
                    CK(hipMemcpy(stamps, init, 64, hipMemcpyHostToDevice));
                    CK(hipDeviceSynchronize());
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 20000);      // 200 us
                    switch (mode) {
                        case 0: launch_producer<0>(s, stamps, big, n4, work); break;
                        case 1: launch_producer<1>(s, stamps, big, n4, work); break;
                        case 2: launch_producer<2>(s, stamps, big, n4, work); break;
                        case 3: launch_producer<3>(s, stamps, big, n4, work); break;
                        case 4: launch_producer<4>(s, stamps, big, n4, work); break;
                        default: launch_producer<5>(s, stamps, big, n4, work); break;
                    }
                    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, s, stamps);
                    hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, s, stamps, big, n4, sink);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(h.data(), stamps, 8 * h.size(), hipMemcpyDeviceToHost));
                    const auto p0 = *std::min_element(h.begin() + 8, h.begin() + 8 + 2048);
                    const auto p1 = *std::max_element(h.begin() + 8 + 2048, h.begin() + 8 + 4096);
                    const auto r0 = *std::min_element(h.begin() + 8 + 4096, h.begin() + 8 + 6144);
                    const auto r1 = *std::max_element(h.begin() + 8 + 6144, h.begin() + 8 + 8192);
                    gap.push_back(((double)h[1] - (double)p1) / 100.0);
                    dur.push_back(((double)p1 - (double)p0) / 100.0);
                    rd.push_back(((double)r1 - (double)r0) / 100.0);
                }
                std::sort(gap.begin(), gap.end()); std::sort(dur.begin(), dur.end()); std::sort(rd.begin(), rd.end());
                printf("%-12s %8zu %6d | %8.1f us %6.1f us %8.1f us\n", names[mode], mb, work, dur[reps / 2], gap[reps / 2], rd[reps / 2]);
            }
        }
    return 0;
}
