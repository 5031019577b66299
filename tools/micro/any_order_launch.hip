// Does hipExtAnyOrderLaunch (an AQL packet without the barrier bit) let two kernels of ONE stream overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/any_order_launch.hip -o /tmp/any_order && /tmp/any_order
// Two single-workgroup spin kernels of ~100 us each: launched normally they take ~200 us back to back; if the flag is
// honoured the second starts while the first runs and the pair takes ~100 us.  A third, normal, launch behind them
// must still wait for BOTH (checked through a flag each spin kernel sets on exit).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(long long cycles, int *done, int slot) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) atomicExch(&done[slot], 1);
}
__global__ void check(const int *done, int *seen) { if (threadIdx.x == 0) *seen = done[0] + 2 * done[1]; }

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    int *done, *seen;
    OK(hipMalloc(&done, 8)); OK(hipMalloc(&seen, 4));
    hipStream_t st; OK(hipStreamCreate(&st));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    const long long cycles = 10000;        // wall_clock64 ticks at 100 MHz: 100 us
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f; int s = -1;
        for (int rep = 0; rep < 5; ++rep) {
            OK(hipMemsetAsync(done, 0, 8, st));
            OK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, cycles, done, 0);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, cycles, done, 1);
            else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cycles, done, 1);
            hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, st, done, seen);
            OK(hipEventRecord(e1, st));
            OK(hipStreamSynchronize(st));
            float ms; OK(hipEventElapsedTime(&ms, e0, e1));
            OK(hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost));
            if (ms < best) best = ms;
        }
        printf("%s second launch: pair + check = %.1f us, the check saw done = %d (3 = both spin kernels finished first)\n",
               mode ? "any-order" : "in-order ", best * 1e3f, s);
    }
    return 0;
}
