// FETCH_SIZE calibration by access pattern (rocprofv3 --pmc FETCH_SIZE -- ./fetch_patterns): every kernel reads a buffer of
// known size exactly once (1.68 GB, far beyond the 256 MB Infinity Cache, flushed between kernels by reading a second buffer),
// so raw FETCH_SIZE / bytes is the pattern's factor.  Round 3 applied the x2.00 of a 16-byte streaming kernel to
// sinkhorn_blk145_kernel, whose lanes read 36-byte runs at a 36-byte stride (16 + 16 + 4 bytes, 4-byte aligned), and reported
// 1.48x "over-fetch"; its raw FETCH_SIZE equalled the algorithmic bytes.  Patterns: a = float4 per lane, b = float2 per lane,
// c = float per lane (all linear), d = the 9x9 register-block reads of sinkhorn_blk145_kernel (same address arithmetic),
// e = the expansion's row reads of the same matrices (one 145-float row per 16 lanes).
// build: hipcc --offload-arch=gfx950 -O3 tools/fetch_patterns.hip -o /tmp/fetch_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int N_ = 145, BS = 9;
typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));
__global__ void pat_a(const float4* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}
__global__ void flush_kernel(const float4* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}
__global__ void pat_b(const float2* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float2 v = p[i]; s += v.x + v.y; }
    if (s == 12345.f) out[0] = s;
}
__global__ void pat_c(const float* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += p[i];
    if (s == 12345.f) out[0] = s;
}
__global__ void __launch_bounds__(256) pat_d(const float* __restrict__ Z, float* out) {      // one workgroup per 145 x 145 problem
    const int t = threadIdx.x, J = t & 15, I = t >> 4;
    const float* Zp = Z + (size_t)blockIdx.x * (N_ * N_);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < BS; ++r) {
        const float* row = Zp + (BS * I + r) * N_ + BS * J;
        const f4a x0 = *reinterpret_cast<const f4a*>(row), x1 = *reinterpret_cast<const f4a*>(row + 4);
        s += x0.x + x0.y + x0.z + x0.w + x1.x + x1.y + x1.z + x1.w + row[8];
    }
    if (J < BS) s += Zp[(BS * I + J) * N_ + 144];
    if (I < BS) s += Zp[144 * N_ + BS * J + I];
    if (t == 0) s += Zp[144 * N_ + 144];
    if (s == 12345.f) out[0] = s;
}
__global__ void __launch_bounds__(256) pat_e(const float* __restrict__ Z, size_t rows, float* out) {   // 16 lanes per matrix row
    const size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    if (row >= rows) return;
    const float* p = Z + row * N_;
    float s = 0.f;
    for (int j = l; j < N_; j += 16) s += p[j];
    if (s == 12345.f) out[0] = s;
}
int main() {
    const size_t P = 19995, n = P * N_ * N_;                      // the bench's fine level: 1.68 GB
    float *z, *flush, *out;
    hipMalloc(&z, n * 4); hipMalloc(&flush, (size_t)1 << 30); hipMalloc(&out, 64);
    hipMemset(z, 0, n * 4); hipMemset(flush, 0, (size_t)1 << 30);
    auto fl = [&] { flush_kernel<<<4096, 256>>>((const float4*)flush, ((size_t)1 << 30) / 16, out); };
    for (int rep = 0; rep < 2; ++rep) {
        fl(); pat_a<<<8192, 256>>>((const float4*)z, n / 4, out);
        fl(); pat_b<<<8192, 256>>>((const float2*)z, n / 2, out);
        fl(); pat_c<<<8192, 256>>>(z, n, out);
        fl(); pat_d<<<(unsigned)P, 256>>>(z, out);
        fl(); pat_e<<<(unsigned)((P * N_ + 15) / 16), 256>>>(z, P * N_, out);
    }
    hipDeviceSynchronize();
    printf("bytes per pattern kernel: %zu (flush_kernel reads 1 GiB between them)\n", n * 4);
    return 0;
}
