// WRITE_SIZE calibration by store pattern (rocprofv3 --pmc WRITE_SIZE -- ./write_patterns): the store-side twin of fetch_patterns.hip.
// Every kernel writes a buffer of known size exactly once (1.68 GB = the fine level's 19 995 x 145 x 145 floats, far beyond the 256 MB
// Infinity Cache), so raw WRITE_SIZE x 1024 / bytes is the pattern's factor - the independent backing of the x0.93 that
// tools/pmc_step.py derives from the cost build's known byte count (round-5 verdict, weak 9).  Patterns: a = float4 per lane,
// b = float2 per lane, c = float per lane (all linear), d = the cost build's MFMA-layout stores (a 16 x 16 tile's accumulators:
// four consecutive rows of one column per lane, 145-float row pitch), e = [264,145]-style rows of 145 floats, 64 + 64 + 17 per wave
// (the gathers' output).
// build: hipcc --offload-arch=gfx950 -O3 tools/write_patterns.hip -o /tmp/write_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int N_ = 145;
__global__ void wpat_a(float4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void wpat_b(float2* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float2(1.f, (float)i);
}
__global__ void wpat_c(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (float)i;
}
// one workgroup per 145 x 145 problem, 16 x 16 output tiles in MFMA accumulator layout: lane (k = lane / 16, col = lane % 16) holds rows
// 4 k .. 4 k + 3 of its column
__global__ void __launch_bounds__(256) wpat_d(float* __restrict__ Z) {
    float* Zp = Z + (size_t)blockIdx.x * (N_ * N_);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 15, k = lane >> 4;
    for (int t = wave; t < 100; t += 4) {                        // 10 x 10 tiles of 16 x 16 (the last ones ragged)
        const int r0 = (t / 10) * 16 + 4 * k, c = (t % 10) * 16 + col;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r0 + r < N_ && c < N_) Zp[(r0 + r) * N_ + c] = (float)(t + r);
    }
}
// rows of 145 floats written by a wave as 64 + 64 + 17 (the descriptor gathers' stores); 4 waves take rows w, w + 4, ...
__global__ void __launch_bounds__(256) wpat_e(float* __restrict__ Z, size_t rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (size_t row = (size_t)blockIdx.x * 4 + wave; row < rows; row += (size_t)gridDim.x * 4) {
        float* p = Z + row * N_;
        p[lane] = 1.f; p[lane + 64] = 2.f;
        if (lane < 17) p[lane + 128] = 3.f;
    }
}
__global__ void flush_kernel(const float4* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}
int main() {
    const size_t P = 19995, n = P * N_ * N_;
    float *z, *flush, *out;
    if (hipMalloc(&z, n * 4) || hipMalloc(&flush, (size_t)1 << 30) || hipMalloc(&out, 4)) return 1;
    hipMemset(flush, 0, (size_t)1 << 30);
    printf("bytes per pattern kernel: %zu (flush_kernel reads 1 GiB between them)\n", n * 4);
    for (int rep = 0; rep < 2; ++rep) {
        flush_kernel<<<4096, 256>>>((const float4*)flush, ((size_t)1 << 30) / 16, out);
        wpat_a<<<8192, 256>>>((float4*)z, n / 4);
        flush_kernel<<<4096, 256>>>((const float4*)flush, ((size_t)1 << 30) / 16, out);
        wpat_b<<<8192, 256>>>((float2*)z, n / 2);
        flush_kernel<<<4096, 256>>>((const float4*)flush, ((size_t)1 << 30) / 16, out);
        wpat_c<<<8192, 256>>>(z, n);
        flush_kernel<<<4096, 256>>>((const float4*)flush, ((size_t)1 << 30) / 16, out);
        wpat_d<<<(unsigned)P, 256>>>(z);
        flush_kernel<<<4096, 256>>>((const float4*)flush, ((size_t)1 << 30) / 16, out);
        wpat_e<<<16384, 256>>>(z, P * N_);
    }
    return hipDeviceSynchronize() != hipSuccess;
}
