// Microbenchmark: bytes per cycle per CU that global loads deliver from an L2-resident buffer, by access pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2_read_bw.hip -o gpurun_out/micro/l2_read_bw && gpurun_out/micro/l2_read_bw
#include <hip/hip_runtime.h>
#include <cstdio>

// Each thread issues UNROLL independent 16-byte loads per iteration.  A wave's 64 lanes cover 64 / lanes_per_row rows
// (lanes of a row read consecutive 16-byte chunks), like a GEMM panel load; iteration `it` moves along the rows.
template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const float4 *buf, size_t n4, int iters, int lanes_per_row, size_t row_f4,
                                                  float *out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane / lanes_per_row, c = lane % lanes_per_row, rows_per_wave = 64 / lanes_per_row;
    float acc = 0.f;
    const size_t row = ((size_t)blockIdx.x * 4 + wave) * rows_per_wave * UNROLL + r;
    for (int it = 0; it < iters; ++it) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t at = ((row + (size_t)u * rows_per_wave) * row_f4 + c + (size_t)it * lanes_per_row) % n4;
            v[u] = buf[at];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t bytes = 16u << 20;
    float4 *buf; float *out;
    (void)hipMalloc(&buf, bytes); (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t n4 = bytes / 16;
    struct Case { const char *name; int lanes_per_row; size_t row_f4; int blocks; };
    const Case cases[] = {
        {"64 lanes x 16 B contiguous (1 KB / instr)", 64, 64, 2048},
        {"8 lanes per 128-B row, rows 592 B apart", 8, 37, 2048},
        {"8 lanes per 128-B row, rows 512 B apart", 8, 32, 2048},
        {"8 lanes per 128-B row, rows 1600 B apart", 8, 100, 2048},
        {"8 lanes per 128-B row, rows 1600 B apart, 1 WG/CU", 8, 100, 256},
        {"64 lanes contiguous, 1 WG/CU", 64, 64, 256},
    };
    for (const Case &c : cases) {
        const int iters = 64;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(read_kernel<8>, dim3(c.blocks), dim3(256), 0, 0, buf, n4, iters, c.lanes_per_row, c.row_f4, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double total = (double)c.blocks * 256 * 8 * iters * 16;
        printf("%-52s %6.1f MB in %7.3f ms = %6.2f TB/s = %5.1f B/cycle/CU (2.4 GHz, 256 CUs)\n", c.name, total / 1e6, ms,
               total / ms / 1e9, total / (ms * 1e-3) / 2.4e9 / 256);
    }
    return 0;
}
