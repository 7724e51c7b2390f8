// Store / load pattern microbenchmark (diagnostic binary, not part of the product):
//   hipcc --offload-arch=gfx950 -O3 tools/store_patterns.hip -o /tmp/store_patterns && /tmp/store_patterns
// How fast can [b][128][65] fp32 (863 MB) be WRITTEN with (a) the MFMA-layout epilogue of the 1x1-convolution kernels
// (per instruction: 2 channel rows x 32 consecutive tokens), (b) one row run of 64 tokens per instruction, (c-f)
// linear 16 / 16-misaligned / 8 / 4 byte stores; and 20 736 x [145][145] read / written in sinkhorn_blk145_kernel's
// 9 x 9 block layout (g) or linearly per workgroup (h).  Result (profiles/r02_store_patterns.txt): a store
// instruction that does not cover whole 128-byte lines runs at a quarter of the linear rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256) pat_a(float* y, int M, int n, long cols) {   // tile 128 rows x 64 cols, as conv_lean_kernel
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, kg = lane >> 5;
    const long tiles_j = (cols + 63) / 64;
    const int i0 = (int)(blockIdx.x / tiles_j) * 128;
    const long j0 = (blockIdx.x % tiles_j) * 64;
    for (int tj = 0; tj < 2; ++tj) {
        if (j0 + 32 * tj + li >= cols) continue;
        const unsigned cg = (unsigned)(j0 + 32 * tj + li), b = cg / (unsigned)n, tk = cg - b * (unsigned)n;
        for (int e = 0; e < 16; ++e) {
            const int row = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * kg;
            if (row < M) y[((long)b * M + row) * n + tk] = (float)(e + lane);
        }
    }
}
__global__ void __launch_bounds__(256) pat_b(float* y, int M, int n, long cols) {   // same tile, one row x 64 tokens per instruction
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long tiles_j = (cols + 63) / 64;
    const int i0 = (int)(blockIdx.x / tiles_j) * 128;
    const long j0 = (blockIdx.x % tiles_j) * 64;
    if (j0 + lane >= cols) return;
    const unsigned cg = (unsigned)(j0 + lane), b = cg / (unsigned)n, tk = cg - b * (unsigned)n;
    for (int e = 0; e < 32; ++e) {
        const int row = i0 + 32 * wave + e;
        if (row < M) y[((long)b * M + row) * n + tk] = (float)(e + lane);
    }
}
__global__ void __launch_bounds__(256) pat_c(float4* y, long n4) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) y[i] = float4{1.f, 2.f, 3.f, 4.f};
}
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void __launch_bounds__(256) pat_d(float* y, long n4) {        // linear 16-byte stores, base misaligned by 4 bytes
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) *reinterpret_cast<f4u*>(y + 1 + 4 * i) = f4u{1.f, 2.f, 3.f, 4.f};
}
__global__ void __launch_bounds__(256) pat_e(float* y, long n2) {        // linear 8-byte stores
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) *reinterpret_cast<float2*>(y + 2 * i) = float2{1.f, 2.f};
}
__global__ void __launch_bounds__(256) pat_f(float* y, long n1) {        // linear 4-byte stores
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n1) y[i] = 1.f;
}
__global__ void __launch_bounds__(256) pat_g(float* out, const float* in, int mode) {    // sinkhorn_blk145_kernel's block layout: 145 x 145 per workgroup
    const int t = threadIdx.x, J = t & 15, I = t >> 4;
    float* Op = out + (long)blockIdx.x * 145 * 145;
    const float* Ip = in + (long)blockIdx.x * 145 * 145;
    float acc = 0.f;
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) {
            const int e = (9 * I + r) * 145 + 9 * J + c;
            if (mode & 1) acc += Ip[e];
            if (mode & 2) Op[e] = (float)e + acc;
        }
    if (mode == 1 && acc == 12345.f) Op[0] = acc;
}
__global__ void __launch_bounds__(256) pat_g4(float* out, const float* in, int mode) {   // the block layout with 16 + 16 + 4 byte accesses per block row
    typedef float v4 __attribute__((ext_vector_type(4), aligned(4)));
    const int t = threadIdx.x, J = t & 15, I = t >> 4;
    float* Op = out + (long)blockIdx.x * 145 * 145;
    const float* Ip = in + (long)blockIdx.x * 145 * 145;
    float acc = 0.f;
    for (int r = 0; r < 9; ++r) {
        const int e = (9 * I + r) * 145 + 9 * J;
        if (mode & 1) {
            const v4 a = *reinterpret_cast<const v4*>(Ip + e), b = *reinterpret_cast<const v4*>(Ip + e + 4);
            acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + Ip[e + 8];
        }
        if (mode & 2) {
            *reinterpret_cast<v4*>(Op + e) = v4{acc, 1.f, 2.f, 3.f};
            *reinterpret_cast<v4*>(Op + e + 4) = v4{acc, 1.f, 2.f, 3.f};
            Op[e + 8] = acc;
        }
    }
    if (mode == 1 && acc == 12345.f) Op[0] = acc;
}
__global__ void __launch_bounds__(256) pat_h(float* out, const float* in, int mode) {    // the same bytes, linear per workgroup
    float* Op = out + (long)blockIdx.x * 145 * 145;
    const float* Ip = in + (long)blockIdx.x * 145 * 145;
    float acc = 0.f;
    for (int e = threadIdx.x; e < 145 * 145; e += 256) {
        if (mode & 1) acc += Ip[e];
        if (mode & 2) Op[e] = (float)e + acc;
    }
    if (mode == 1 && acc == 12345.f) Op[0] = acc;
}
int main() {
    const int M = 128, n = 65; const long b = 25920, cols = b * n, total = (long)b * M * n;
    float* y; hipMalloc(&y, total * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto f) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 10; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-40s %.3f ms  %.0f GB/s\n", name, ms, total * 4.0 / ms / 1e6);
    };
    const unsigned g = (unsigned)((cols + 63) / 64);
    run("a: 2 rows x 32 tokens per instruction", [&] { hipLaunchKernelGGL(pat_a, dim3(g), dim3(256), 0, 0, y, M, n, cols); });
    run("b: 1 row x 64 tokens per instruction", [&] { hipLaunchKernelGGL(pat_b, dim3(g), dim3(256), 0, 0, y, M, n, cols); });
    run("c: linear float4", [&] { hipLaunchKernelGGL(pat_c, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, 0, (float4*)y, total / 4); });
    run("d: linear 16-byte, base + 4 bytes", [&] { hipLaunchKernelGGL(pat_d, dim3((unsigned)((total / 4 - 1 + 255) / 256)), dim3(256), 0, 0, y, total / 4 - 1); });
    run("e: linear 8-byte", [&] { hipLaunchKernelGGL(pat_e, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, 0, y, total / 2); });
    run("f: linear 4-byte", [&] { hipLaunchKernelGGL(pat_f, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, y, total); });
    {
        const long B = 20736, tot = B * 145 * 145;
        float *zi, *zo; hipMalloc(&zi, tot * 4); hipMalloc(&zo, tot * 4); hipMemset(zi, 0, tot * 4);
        auto run2 = [&](const char* name, auto f) {
            f(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("%-44s %.3f ms  (1.74 GB each way)\n", name, ms);
        };
        run2("g1: blk145 block layout, loads only", [&] { hipLaunchKernelGGL(pat_g, dim3((unsigned)B), dim3(256), 0, 0, zo, zi, 1); });
        run2("g2: blk145 block layout, stores only", [&] { hipLaunchKernelGGL(pat_g, dim3((unsigned)B), dim3(256), 0, 0, zo, zi, 2); });
        run2("g3: blk145 block layout, both", [&] { hipLaunchKernelGGL(pat_g, dim3((unsigned)B), dim3(256), 0, 0, zo, zi, 3); });
        run2("g4-1: block layout, 16+16+4 B, loads only", [&] { hipLaunchKernelGGL(pat_g4, dim3((unsigned)B), dim3(256), 0, 0, zo, zi, 1); });
        run2("g4-2: block layout, 16+16+4 B, stores only", [&] { hipLaunchKernelGGL(pat_g4, dim3((unsigned)B), dim3(256), 0, 0, zo, zi, 2); });
        run2("g4-3: block layout, 16+16+4 B, both", [&] { hipLaunchKernelGGL(pat_g4, dim3((unsigned)B), dim3(256), 0, 0, zo, zi, 3); });
        run2("h1: linear per workgroup, loads only", [&] { hipLaunchKernelGGL(pat_h, dim3((unsigned)B), dim3(256), 0, 0, zo, zi, 1); });
        run2("h2: linear per workgroup, stores only", [&] { hipLaunchKernelGGL(pat_h, dim3((unsigned)B), dim3(256), 0, 0, zo, zi, 2); });
        run2("h3: linear per workgroup, both", [&] { hipLaunchKernelGGL(pat_h, dim3((unsigned)B), dim3(256), 0, 0, zo, zi, 3); });
    }
    return 0;
}
