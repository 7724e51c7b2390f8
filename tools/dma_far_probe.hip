// How fast does ONE CU's LDS DMA pull 153 KB images that are NOT in its L2 (round 5, gnn_fine.hip)?  256 workgroups of 512 threads,
// each walks its own images of a 2.4 GB buffer (global_load_lds_dwordx4, 16 bytes a lane).
//   whole : all 19 loads a wave of an image in flight, s_waitcnt vmcnt(0), next image        (the first version's "fill")
//   ring L: 18.5 KB k-steps, three loads a wave each, k-step j + L issued before waiting for k-step j (vmcnt(3 L))  - no compute at all:
//           the rate a k-step ring can sustain at lookahead L, i.e. latency / L per step
// build: hipcc --offload-arch=gfx950 -O3 tools/dma_far_probe.hip -o /tmp/dma_far_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int IMG = 153120, KS = 18560;
__global__ void __launch_bounds__(512, 1) whole_kernel(const char* __restrict__ buf, int images, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.f;
    for (int i = blockIdx.x; i < images; i += gridDim.x) {
        const char* src = buf + (size_t)i * IMG;
        for (int base = wave * 64; base < IMG / 16; base += 512)
            if (base + lane < IMG / 16)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(base + lane) * 16),
                                                 (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        s += *(const float*)(lds + lane * 4);
        __syncthreads();
    }
    if (s == 12345.f) out[0] = s;
}
template <int L>
__global__ void __launch_bounds__(512, 1) ring_kernel(const char* __restrict__ buf, int images, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.f;
    const int per = (images - blockIdx.x + gridDim.x - 1) / gridDim.x, steps = per * 8;       // eight full k-steps per image
    auto issue = [&](int n) {
        const int img = blockIdx.x + (n >> 3) * gridDim.x, ks = n & 7;
        const char* src = buf + (size_t)img * IMG + (size_t)ks * KS + (size_t)(wave * 145 + lane) * 16;
        char* d = lds + (n % (L + 1)) * KS + wave * (145 * 16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1024), (__attribute__((address_space(3))) void*)(d + 1024), 16, 0, 0);
        if (lane < 17)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2048), (__attribute__((address_space(3))) void*)(d + 2048), 16, 0, 0);
    };
    for (int n = 0; n < L && n < steps; ++n) issue(n);
    for (int n = 0; n < steps; ++n) {
        if (n + L < steps) issue(n + L);
        // at most the loads of the L k-steps issued after k-step n may be outstanding
        if (L == 1) __builtin_amdgcn_s_waitcnt(0x0F73);
        else if (L == 2) __builtin_amdgcn_s_waitcnt(0x0F76);
        else if (L == 3) __builtin_amdgcn_s_waitcnt(0x0F79);
        else if (L == 4) __builtin_amdgcn_s_waitcnt(0x0F7C);
        else __builtin_amdgcn_s_waitcnt(0x0F7F);       // L == 5: vmcnt(15)
        __syncthreads();
        s += *(const float*)(lds + (n % (L + 1)) * KS + lane * 4);
    }
    if (s == 12345.f) out[0] = s;
}
template <typename F>
static void timed(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-26s %8.3f ms  %6.1f GB/s per CU  aggregate %5.2f TB/s  (%.2f us per 153 KB image per CU)\n", name, best, bytes / 256 / (best * 1e-3) / 1e9,
           bytes / (best * 1e-3) / 1e12, best * 1e3 / (bytes / 256 / IMG));
}
int main() {
    const int images = 16384;                       // 2.5 GB: nothing stays in the 256 MB Infinity Cache
    char* buf; float* out;
    if (hipMalloc((void**)&buf, (size_t)images * IMG) != hipSuccess || hipMalloc((void**)&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, (size_t)images * IMG);
    hipFuncSetAttribute((const void*)whole_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    timed("whole image, then wait", (double)images * IMG, [&] { hipLaunchKernelGGL(whole_kernel, dim3(256), dim3(512), IMG + 32, 0, buf, images, out); });
#define RING(L) hipFuncSetAttribute((const void*)ring_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    timed("ring, lookahead " #L, (double)images * 8 * KS, [&] { hipLaunchKernelGGL(ring_kernel<L>, dim3(256), dim3(512), (L + 1) * KS, 0, buf, images, out); })
    RING(1); RING(2); RING(3); RING(4); RING(5);
    // the same from a 100 MB buffer (Infinity-Cache resident after the first pass)
    const int small = 640;
    timed("whole image, 98 MB set", (double)small * IMG * 8, [&] { for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(whole_kernel, dim3(256), dim3(512), IMG + 32, 0, buf, small, out); });
    timed("ring 2, 98 MB set", (double)small * 8 * KS * 8, [&] { for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(ring_kernel<2>, dim3(256), dim3(512), 3 * KS, 0, buf, small, out); });
    timed("ring 4, 98 MB set", (double)small * 8 * KS * 8, [&] { for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(ring_kernel<4>, dim3(256), dim3(512), 5 * KS, 0, buf, small, out); });
    return 0;
}
