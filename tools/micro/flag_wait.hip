// Microbenchmark + correctness probe: a dependency INSIDE one launch across the eight XCDs of gfx950 (each has its own L2).
// Producer workgroups (low block indices, so they are dispatched first) write an "image"; consumer workgroups (high block
// indices) wait for a counter and then read the whole image — the shape of "the step prologue builds the convolution
// weight image, the world encoder's workgroups of the same launch read it".  Stale copies of the image from the previous
// launch sit in the consumers' L2s, so the variants show which release / acquire forms are CORRECT (mismatches = 0 over
// many launches) and what they cost against two launches.
//   store: 0 plain stores + __threadfence() by every producer thread   1 sc1 (written-through) stores + vmcnt(0)
//   acquire: 0 none (expected to read stale lines)   1 __threadfence() by every consumer thread after the wait
//            2 consumers read the image with sc1 loads (no fence)   3 thread 0's __threadfence(), then a barrier
//            4 thread 0's ACQUIRE-only fence (buffer_inv, no write-back), then a barrier   5 every thread's acquire fence
//   hipcc --offload-arch=gfx950 -O3 tools/micro/flag_wait.hip -o gpurun_out/micro/fw && gpurun_out/micro/fw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int kImg = 77 * 1024;          // floats
constexpr int NP = 308;                  // producer workgroups (256 floats each)
constexpr int NF = 1024;                 // filler workgroups writing other data (the rest of the prologue): 8 MB
constexpr int kSlice = 8192;            // floats of the image a consumer reads
constexpr int NC = 1024;                 // consumer workgroups (512 threads)

__device__ __forceinline__ float value_of(int i, int epoch) { return (float)(i % 977) + 1000.f * (float)(epoch % 89); }

template <int STORE, int ACQ>
__global__ __launch_bounds__(512) void fused(float *img, float *other, unsigned *counter, unsigned *errors, int epoch,
                                             unsigned target, unsigned long long *stamps) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b == 0 && tid == 0) stamps[0] = wall_clock64();
    if (b < NP) {
        if (tid < 256) {
            const int i = b * 256 + tid;
            if (i < kImg) {
                if (STORE == 0) img[i] = value_of(i, epoch);
                else __hip_atomic_store(img + i, value_of(i, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (STORE == 0) __threadfence();
            else __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the written-through stores are acknowledged
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (b < NP + NF) {                                          // filler: 8 KB per workgroup of unrelated stores
        float4 *o = reinterpret_cast<float4 *>(other) + (size_t)(b - NP) * 512;
        o[tid] = float4{(float)tid, (float)epoch, 0.f, 1.f};
        return;
    }
    // consumer
    if (tid == 0) {
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    if (ACQ == 1) __threadfence();
    if (ACQ == 3) { if (tid == 0) __threadfence(); __syncthreads(); }      // one thread's acquire, then the barrier
    if (ACQ == 4) { if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __syncthreads(); }   // invalidate only
    if (ACQ == 5) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    unsigned bad = 0;
    for (int k = tid; k < kSlice; k += 512) {
        const int i = (k + (b - NP - NF) * 1237) % kImg;
        float v;
        if (ACQ == 2) v = __hip_atomic_load(img + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v = img[i];
        bad += v != value_of(i, epoch);
    }
    if (bad) atomicAdd(errors, bad);
    __syncthreads();
    if (b == NP + NF + NC - 1 && tid == 0) stamps[1] = wall_clock64();
}

__global__ void produce_only(float *img, float *other, int epoch, unsigned long long *stamps) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b == 0 && tid == 0) stamps[0] = wall_clock64();
    if (b < NP) { const int i = b * 256 + tid; if (i < kImg) img[i] = value_of(i, epoch); return; }
    float4 *o = reinterpret_cast<float4 *>(other) + (size_t)(b - NP) * 512;
    o[tid] = float4{(float)tid, (float)epoch, 0.f, 1.f};
    o[tid + 256] = float4{(float)tid, (float)epoch, 0.f, 1.f};
}
__global__ __launch_bounds__(512) void consume_only(const float *img, unsigned *errors, int epoch, unsigned long long *stamps) {
    const int tid = threadIdx.x;
    unsigned bad = 0;
    for (int k = tid; k < kSlice; k += 512) { const int i = (k + (int)blockIdx.x * 1237) % kImg; bad += img[i] != value_of(i, epoch); }
    if (bad) atomicAdd(errors, bad);
    __syncthreads();
    if (blockIdx.x == NC - 1 && tid == 0) stamps[1] = wall_clock64();
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int STORE, int ACQ>
int run(const char *name, float *img, float *other, unsigned *counter, unsigned *errors, unsigned long long *stamps, hipStream_t s) {
    CK(hipMemset(counter, 0, 4)); CK(hipMemset(errors, 0, 4));
    std::vector<double> us;
    const int epochs = 300;
    for (int e = 0; e < epochs; ++e) {
        hipLaunchKernelGGL((fused<STORE, ACQ>), dim3(NP + NF + NC), dim3(512), 0, s, img, other, counter, errors, e,
                           (unsigned)NP * (e + 1), stamps);
        CK(hipStreamSynchronize(s));
        unsigned long long h[2];
        CK(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
        us.push_back(((double)h[1] - (double)h[0]) / 100.0);
    }
    unsigned bad = 0;
    CK(hipMemcpy(&bad, errors, 4, hipMemcpyDeviceToHost));
    std::sort(us.begin(), us.end());
    printf("%-58s mismatches %10u   first start -> last consumer's exit: median %6.1f us  min %6.1f\n", name, bad, us[epochs / 2], us[0]);
    return 0;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *img, *other; unsigned *counter, *errors; unsigned long long *stamps;
    CK(hipMalloc(&img, kImg * 4)); CK(hipMalloc(&other, (size_t)NF * 512 * 16)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&errors, 4));
    CK(hipMalloc(&stamps, 16));
    {   // reference: two launches
        CK(hipMemset(errors, 0, 4));
        std::vector<double> us;
        for (int e = 0; e < 300; ++e) {
            hipLaunchKernelGGL(produce_only, dim3(NP + NF), dim3(256), 0, s, img, other, e, stamps);
            hipLaunchKernelGGL(consume_only, dim3(NC), dim3(512), 0, s, img, errors, e, stamps);
            CK(hipStreamSynchronize(s));
            unsigned long long h[2];
            CK(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
            us.push_back(((double)h[1] - (double)h[0]) / 100.0);
        }
        unsigned bad = 0;
        CK(hipMemcpy(&bad, errors, 4, hipMemcpyDeviceToHost));
        std::sort(us.begin(), us.end());
        printf("%-58s mismatches %10u   first start -> last consumer's exit: median %6.1f us  min %6.1f\n", "two launches", bad, us[150], us[0]);
    }
    if (run<0, 0>("one launch: plain stores + fence | no acquire", img, other, counter, errors, stamps, s)) return 1;
    if (run<0, 1>("one launch: plain stores + fence | fence after the wait", img, other, counter, errors, stamps, s)) return 1;
    if (run<1, 1>("one launch: sc1 stores + vmcnt(0) | fence after the wait", img, other, counter, errors, stamps, s)) return 1;
    if (run<1, 0>("one launch: sc1 stores + vmcnt(0) | no acquire", img, other, counter, errors, stamps, s)) return 1;
    if (run<1, 2>("one launch: sc1 stores + vmcnt(0) | sc1 loads", img, other, counter, errors, stamps, s)) return 1;
    if (run<0, 2>("one launch: plain stores + fence | sc1 loads", img, other, counter, errors, stamps, s)) return 1;
    if (run<1, 3>("one launch: sc1 stores + vmcnt(0) | one thread's fence", img, other, counter, errors, stamps, s)) return 1;
    if (run<0, 3>("one launch: plain stores + fence | one thread's fence", img, other, counter, errors, stamps, s)) return 1;
    if (run<1, 4>("one launch: sc1 stores + vmcnt(0) | one thread's acquire fence", img, other, counter, errors, stamps, s)) return 1;
    if (run<1, 5>("one launch: sc1 stores + vmcnt(0) | every thread's acquire fence", img, other, counter, errors, stamps, s)) return 1;
    return 0;
}
