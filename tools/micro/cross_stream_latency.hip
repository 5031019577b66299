// Microbenchmark: how long after a kernel on stream A ends does a dependent kernel on stream B start, by mechanism:
//   (1) hipEventRecord / hipStreamWaitEvent   (2) hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory
//   (3) same stream (in-order launch), for reference.   (10, 11) hipExtLaunchKernelGGL(..., stopEvent): the kernel's own
//   completion signal is the event, no marker packet behind it
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cross_stream_latency.hip -o gpurun_out/micro/xs && gpurun_out/micro/xs
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <algorithm>
#include <vector>

__global__ void producer(unsigned long long *stamps, int spin) {   // ~spin x 10 ns of busy work, then its end time
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = wall_clock64();
}
// a chip-filling producer that dirties `n4` float4 of memory, its end time = the LAST workgroup's exit (atomicMax)
__global__ void heavy_producer(unsigned long long *stamps, float4 *buf, size_t n4, int passes) {
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
            buf[i] = float4{(float)i, (float)p, 1.f, 2.f};
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&stamps[0], wall_clock64());
}
// background load on a third stream: keeps every CU partly busy for a few hundred microseconds
__global__ void background(float *out, int iters) {
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0001f, 0.5f);
    if (a == 12345.f) out[0] = a;
}
__global__ void consumer(unsigned long long *stamps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = wall_clock64();
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned long long *stamps;
    CK(hipMalloc(&stamps, 16));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
    uint32_t *flag = nullptr;
    bool have_signal = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory) == hipSuccess;
    if (!have_signal) { printf("no signal memory; using device memory for the flag\n"); CK(hipMalloc((void **)&flag, 8)); }
    const int reps = 30, spin = 5000;     // 50 us producer
    hipStream_t c;
    CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    float4 *big; float *bg;
    const size_t n4 = (32u << 20) / 16;
    CK(hipMalloc(&big, n4 * 16)); CK(hipMalloc(&bg, 4));
    hipEvent_t evt;
    CK(hipEventCreateWithFlags(&evt, hipEventDisableSystemFence));      // stop events may need timing enabled
    for (int mode = 1; mode <= 11; ++mode) {
        std::vector<double> lat;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(stamps, 0, 16));
            CK(hipMemset(flag, 0, 8));
            CK(hipDeviceSynchronize());
            if (mode == 1) {
                hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, stamps, spin);
                CK(hipEventRecord(ev, a));
                CK(hipStreamWaitEvent(b, ev, 0));
                hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamps);
            } else if (mode == 2) {
                // the waiting stream is armed FIRST (as a side stream armed at the start of a step would be)
                hipError_t e = hipStreamWaitValue32(b, flag, (uint32_t)(r + 1), hipStreamWaitValueEq, 0xffffffffu);
                if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); break; }
                hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamps);
                hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, stamps, spin);
                e = hipStreamWriteValue32(a, flag, (uint32_t)(r + 1), 0);
                if (e != hipSuccess) { printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e)); break; }
            } else if (mode == 10 || mode == 11) {
                hipEvent_t e = mode == 10 ? ev : evt;
                if (mode == 10) hipExtLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, nullptr, e, 0, stamps, spin);
                else hipExtLaunchKernelGGL(heavy_producer, dim3(2048), dim3(256), 0, a, nullptr, e, 0, stamps, big, n4, 1);
                hipError_t rc = hipStreamWaitEvent(b, e, 0);
                if (rc != hipSuccess) { printf("wait on a stop event: %s\n", hipGetErrorString(rc)); break; }
                hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamps);
            } else if (mode == 3) {
                hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, stamps, spin);
                hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, a, stamps);
            } else {
                // modes 4-6: heavy producer (32 MB dirtied by 2048 workgroups): same stream / event / wait-value
                // modes 7-9: the same with a background kernel running on a third stream
                const int kind = (mode - 4) % 3;
                if (mode >= 7) hipLaunchKernelGGL(background, dim3(2048), dim3(256), 0, c, bg, 400000);
                if (kind == 2) {
                    hipError_t e = hipStreamWaitValue32(b, flag, (uint32_t)(r + 1), hipStreamWaitValueEq, 0xffffffffu);
                    if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); break; }
                    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamps);
                }
                hipLaunchKernelGGL(heavy_producer, dim3(2048), dim3(256), 0, a, stamps, big, n4, 1);
                if (kind == 0) hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, a, stamps);
                if (kind == 1) {
                    CK(hipEventRecord(ev, a));
                    CK(hipStreamWaitEvent(b, ev, 0));
                    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamps);
                }
                if (kind == 2) CK(hipStreamWriteValue32(a, flag, (uint32_t)(r + 1), 0));
            }
            CK(hipDeviceSynchronize());
            unsigned long long h[2];
            CK(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
            lat.push_back(((double)h[1] - (double)h[0]) / 100.0);   // 100 MHz ticks -> us
        }
        if (lat.empty()) continue;
        std::sort(lat.begin(), lat.end());
        const char *names[] = {"", "event record -> stream wait event", "write value -> wait value (armed early)", "same stream",
                               "heavy producer, same stream", "heavy producer, event", "heavy producer, wait value",
                               "heavy + background, same stream", "heavy + background, event", "heavy + background, wait value",
                               "stop event of the kernel (no-timing event)", "heavy producer, stop event (timing event)"};
        printf("%-42s median %6.1f us  min %6.1f  max %6.1f\n", names[mode], lat[lat.size() / 2], lat.front(), lat.back());
    }
    return 0;
}
