// Issue rate of three f16 matrix instructions on gfx950, one wave per SIMD and two: cycles per instruction (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o /tmp/mfma_rate_probe && /tmp/mfma_rate_probe
// Question behind it (csrc/gnn_fine.hip): would the ragged ninth k-step (channels 256..263) be cheaper on a K = 16 instruction?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// RANDOM: four operand fragments of hashed bits (finite fp16 values), a different pair for every instruction - the toggling a real
// product sees; else constant small integers
template <int KIND, bool RANDOM>
__global__ void __launch_bounds__(1024) probe_data(long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    h8v ar[4], br[4];
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 8; ++e) {
            unsigned h = (unsigned)(threadIdx.x * 8 + e) * 2654435761u + (unsigned)k * 40503u + 12345u;
            h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
            const float va = RANDOM ? (float)((int)(h & 0xffff) - 32768) / 4096.0f : (float)((e + k) & 7);
            const float vb = RANDOM ? (float)((int)(h >> 16) - 32768) / 4096.0f : (float)((e * 3 + k) & 7);
            ar[k][e] = (_Float16)va; br[k][e] = (_Float16)vb;
        }
    f4v c[8];
    for (int k = 0; k < 8; ++k) c[k] = f4v{0, 0, 0, 0};
    f16v d[4];
    for (int k = 0; k < 4; ++k) d[k] = f16v{};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[k & 3], br[(k + (k >> 2)) & 3], c[k], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[k], br[(k + 1) & 3], d[k], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float acc = 0.f;
    for (int k = 0; k < 8; ++k) acc += c[k].x;
    for (int k = 0; k < 4; ++k) acc += d[k][3];
    sink[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int KIND>
__global__ void __launch_bounds__(1024) probe(long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    h8v a8 = {(_Float16)lane, 1, 2, 3, 4, 5, 6, 7}, b8 = {1, (_Float16)lane, 3, 4, 5, 6, 7, 8};
    h4v a4 = {(_Float16)lane, 1, 2, 3}, b4 = {1, (_Float16)lane, 3, 4};
    f4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f16v d0 = {}, d1 = {}, d2 = {}, d3 = {};
    f4v c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {            // v_mfma_f32_16x16x32_f16
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c3, 0, 0, 0);
        } else if (KIND == 1) {     // v_mfma_f32_16x16x16_f16
            c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c3, 0, 0, 0);
        } else if (KIND == 2) {     // v_mfma_f32_32x32x16_f16, two accumulators
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d1, 0, 0, 0);
        } else if (KIND == 3) {     // v_mfma_f32_32x32x16_f16, four accumulators
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d3, 0, 0, 0);
        } else {                    // v_mfma_f32_16x16x32_f16, eight accumulators (two iterations' worth: counted as 4 below, so halve)
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c7, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 1024 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + c4.x + c5.y + c6.z + c7.w + d0[0] + d1[5] + d2[3] + d3[7];
}

int main() {
    long long* out; float* sink;
    hipMalloc(&out, 256 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    const int iters = 4096;
    const char* names[5] = {"v_mfma_f32_16x16x32_f16 (4 acc)", "v_mfma_f32_16x16x16_f16 (4 acc)", "v_mfma_f32_32x32x16_f16 (2 acc)", "v_mfma_f32_32x32x16_f16 (4 acc)", "v_mfma_f32_16x16x32_f16 (8 acc)"};
    const int per_iter[5] = {4, 4, 4, 4, 8};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512, 1024}) {
        for (int kind = 0; kind < 5; ++kind) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
                if (kind == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
                if (kind == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
                if (kind == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
                if (kind == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
            printf("%-34s %d waves per SIMD: %.1f cycles per instruction per wave (%.1f per SIMD)\n", names[kind], threads / 256, s / 256 / (iters * (double)per_iter[kind]),
                   s / 256 / (iters * (double)per_iter[kind]) / (threads / 256));
            {
                const double flop_per = kind == 1 ? 2.0 * 16 * 16 * 16 : (kind == 2 || kind == 3) ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
                const double total = 256.0 * (threads / 64) * iters * per_iter[kind] * flop_per;
                printf("    wall %.3f ms -> %.0f TFLOP/s; shader clock by s_memtime over the wall time: %.0f MHz\n", ms, total / (ms * 1e-3) / 1e12, s / 256 / (ms * 1e-3) / 1e6);
            }
        }
    }
    printf("operand data (one wave per SIMD = the pipe's own rate, 256 workgroups):\n");
    for (int kind = 0; kind < 2; ++kind)
        for (int rnd = 0; rnd < 2; ++rnd) {
            float ms = 0.f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                if (kind == 0 && !rnd) hipLaunchKernelGGL((probe_data<0, false>), dim3(256), dim3(256), 0, 0, out, sink, iters * 4);
                if (kind == 0 && rnd) hipLaunchKernelGGL((probe_data<0, true>), dim3(256), dim3(256), 0, 0, out, sink, iters * 4);
                if (kind == 1 && !rnd) hipLaunchKernelGGL((probe_data<1, false>), dim3(256), dim3(256), 0, 0, out, sink, iters * 4);
                if (kind == 1 && rnd) hipLaunchKernelGGL((probe_data<1, true>), dim3(256), dim3(256), 0, 0, out, sink, iters * 4);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
            const double n = kind == 0 ? 8.0 : 4.0, flop = kind == 0 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
            printf("  %-26s %-8s operands: %.0f TFLOP/s, %.1f cycles per instruction, shader clock %.0f MHz\n", kind == 0 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_32x32x16_f16",
                   rnd ? "random" : "constant", 256.0 * 4 * iters * 4 * n * flop / (ms * 1e-3) / 1e12, s / 256 / (iters * 4 * n), s / 256 / (ms * 1e-3) / 1e6);
        }
    return 0;
}
