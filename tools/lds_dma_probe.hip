// does global_load_lds_dwordx4 reach LDS offsets beyond 64 KB / 128 KB on gfx950, and is a 153 120-byte image copied intact?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int BYTES = 153120, PIECES = BYTES / 16;
__global__ void __launch_bounds__(512, 1) k(const f4v* __restrict__ src, f4v* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int r = 0; r < (PIECES + 511) / 512; ++r) {
        const int base = r * 512 + wave * 64;          // wave-uniform piece index
        if (base + lane < PIECES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + lane),
                                             (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = t; i < PIECES; i += 512) dst[i] = *(const f4v*)(lds + i * 16);
}
int main() {
    std::vector<float> h(PIECES * 4), o(PIECES * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    f4v *s, *d;
    hipMalloc((void**)&s, BYTES); hipMalloc((void**)&d, BYTES);
    hipMemcpy(s, h.data(), BYTES, hipMemcpyHostToDevice);
    hipMemset(d, 0, BYTES);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(4), dim3(512), BYTES + 32, 0, s, d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o.data(), d, BYTES, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < h.size(); ++i) if (o[i] != h[i]) { if (!bad) first = i; ++bad; }
    printf("lds dma probe: %s, mismatches %zu (first at float %zu)\n", hipGetErrorString(e), bad, first);
    return bad != 0;
}
