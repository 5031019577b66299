// Microbenchmark: what rate do the matrix cores sustain on v_mfma_f32_16x16x4_f32 with operands in registers?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.f) out[0] = s;
}

template <int NACC>
void run(int blocks, int iters, const char *label) {
    float *out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2048.0 * NACC * iters * 4.0 * blocks;
    printf("%-28s blocks=%5d iters=%6d  %8.3f ms  %7.1f TFLOP/s\n", label, blocks, iters, ms, flop / ms / 1e9);
}

int main() {
    run<8>(256, 20000, "1 wave/SIMD, 8 acc, short");
    run<8>(256 * 2, 20000, "2 waves/SIMD, 8 acc");
    run<8>(256 * 4, 20000, "4 waves/SIMD, 8 acc");
    run<2>(256 * 4, 80000, "4 waves/SIMD, 2 acc");
    run<16>(256 * 2, 10000, "2 waves/SIMD, 16 acc");
    run<8>(256 * 4, 400000, "4 waves/SIMD, 8 acc, 100+ ms");
    return 0;
}
