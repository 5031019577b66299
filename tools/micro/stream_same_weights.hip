// Microbenchmark: N workgroups of 1024 threads (one per CU) stream the SAME L2-resident weight block over and over, as the
// streaming decoder does per step: bytes per cycle per ACTIVE CU by number of active workgroups and loads in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_same_weights.hip -o gpurun_out/micro/ssw && gpurun_out/micro/ssw
#include <hip/hip_runtime.h>
#include <cstdio>

template <int UNROLL>
__global__ __launch_bounds__(1024) void stream_kernel(const float4 *buf, size_t n4, int passes, float *out) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int p = 0; p < passes; ++p) {
        for (size_t i = tid; i + (size_t)(UNROLL - 1) * 1024 < n4; i += (size_t)UNROLL * 1024) {
            float4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = buf[i + (size_t)u * 1024];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
        __syncthreads();
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t bytes = 1u << 20;          // 1 MB of "weights" (16 H^2 floats at H = 128)
    float4 *buf; float *out;
    (void)hipMalloc(&buf, bytes); (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int passes = 200;
    for (int blocks : {256, 128, 64, 32, 16}) {
        for (int unroll : {4, 8}) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (unroll == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(blocks), dim3(1024), 0, 0, buf, bytes / 16, passes, out);
                else hipLaunchKernelGGL(stream_kernel<8>, dim3(blocks), dim3(1024), 0, 0, buf, bytes / 16, passes, out);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double per_block = (double)bytes * passes;
            printf("%3d workgroups x 1024 threads, %d x 16 B in flight per thread: %7.3f ms, %6.2f us per pass, %5.1f B/cycle per active CU (2.4 GHz), %6.2f TB/s total\n",
                   blocks, unroll, ms, ms * 1e3 / passes, per_block / (ms * 1e-3) / 2.4e9, per_block * blocks / ms / 1e9);
        }
    }
    return 0;
}
