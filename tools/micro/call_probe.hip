// Development aid (round 6): do scratch memory and a noinline, noreturn device function that reads the kernel's argument
// segment, uses dynamic LDS and barriers work from a 512-thread kernel on this stack?   hipcc --offload-arch=gfx950 -O3
//   ./call_probe scratch | call
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

struct Args { int n; const float *in; float *out; int flag; };
extern __shared__ __attribute__((aligned(16))) float smem[];
using KernelArguments = const __attribute__((address_space(4))) void *;

__device__ __forceinline__ const Args &uniform_arguments(KernelArguments args) {
    const uint64_t p = (uint64_t)args;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return *(const Args *)(const __attribute__((address_space(4))) Args *)(((uint64_t)hi << 32) | lo);   // (constant address space: scalar loads)
}

__device__ __attribute__((noinline, noreturn)) void again(KernelArguments ka) {
    const Args &a = uniform_arguments(ka);
    const int tid = threadIdx.x, b = blockIdx.x;
    smem[tid] = a.in[b * 512 + tid] * 2.f;
    // (callee-saved vector and scalar registers written: the prologue saves them in scratch, as the decoder's function does)
    asm volatile("v_mov_b32 v40, 0\n\tv_mov_b32 v127, 0\n\tv_mov_b32 v255, 0\n\ts_mov_b32 s40, 0\n\ts_mov_b32 s99, 0"
                 ::: "v40", "v127", "v255", "s40", "s99");
    __syncthreads();
    a.out[b * 512 + tid] = smem[511 - tid] + 1000.f;
    __builtin_amdgcn_endpgm();
}

__global__ __launch_bounds__(512) void call_kernel(Args a) {
    const int tid = threadIdx.x, b = blockIdx.x;
    smem[tid] = a.in[b * 512 + tid];
    __syncthreads();
    if (a.flag) again((KernelArguments)__builtin_amdgcn_kernarg_segment_ptr());
    a.out[b * 512 + tid] = smem[511 - tid];
}

__global__ __launch_bounds__(512) void scratch_kernel(Args a) {
    const int tid = threadIdx.x, b = blockIdx.x;
    float loc[64];
    for (int i = 0; i < 64; ++i) loc[i] = a.in[b * 512 + ((tid + i) & 511)];
    float s = 0.f;
    for (int i = 0; i < 64; ++i) s += loc[(i * (a.flag + 3)) & 63];      // dynamic index: the array lives in scratch
    a.out[b * 512 + tid] = s;
}

int main(int argc, char **argv) {
    const int B = 256, N = B * 512;
    std::vector<float> h(N), o(N);
    for (int i = 0; i < N; ++i) h[i] = (float)(i % 977);
    float *in, *out;
    hipMalloc(&in, N * 4); hipMalloc(&out, N * 4);
    hipMemcpy(in, h.data(), N * 4, hipMemcpyHostToDevice);
    Args a{N, in, out, 1};
    const bool call = argc > 1 && !strcmp(argv[1], "call");
    if (call) {
        hipFuncSetAttribute((const void *)call_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int flag = 0; flag < 2; ++flag) {
            a.flag = flag;
            hipLaunchKernelGGL(call_kernel, dim3(B), dim3(512), 150 * 1024, 0, a);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(o.data(), out, N * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < N; ++i) {
                const int b = i / 512, t = i % 512;
                const float want = flag ? h[b * 512 + 511 - t] * 2.f + 1000.f : h[b * 512 + 511 - t];
                bad += o[i] != want;
            }
            printf("call flag=%d: %s, %d wrong\n", flag, hipGetErrorString(e), bad);
        }
    } else {
        hipLaunchKernelGGL(scratch_kernel, dim3(B), dim3(512), 0, 0, a);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), out, N * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < N; ++i) {
            const int b = i / 512, t = i % 512;
            float s = 0.f;
            for (int k = 0; k < 64; ++k) s += h[b * 512 + ((t + ((k * 4) & 63)) & 511)];
            bad += o[i] != s;
        }
        printf("scratch: %s, %d wrong\n", hipGetErrorString(e), bad);
    }
    return 0;
}
