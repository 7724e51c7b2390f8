// Stand-alone reproducer for hazard (b) of the round-4 verdict: csrc/cost65_device.hpp:205-215, csrc/mfma_tile.hpp:193-196 and
// csrc/gnn.hip keep `s_nop 15; s_nop 15` + scheduling barriers between a block of v_mfma_f32_32x32x16_f16 and the VALU code that
// REWRITES the registers those MFMAs read as A / B operands (the next chunk's fp32 -> fp16 hi / lo conversions), on the claim that
// hipcc (ROCm 7.2, gfx950) interleaves the two and that the wait states it inserts do not cover that write-after-read.
//
// The kernel below is that loop and nothing else: per iteration eight operand registers are made by VALU conversions of
// iteration-dependent data, twelve MFMAs (the cost build's three passes on a 2 x 2 block of accumulators) read them, and the next
// iteration's conversions overwrite them at once.  GUARD = 1 inserts the production fence, GUARD = 0 leaves the scheduler alone.
// Every variant runs at 1, 2 and 3 waves per SIMD (the matrix pipe queues differently) and is compared, element by element, with
// (i) the same loop run with a full pipeline drain after every MFMA block (GUARD = 2: s_nop 15 x 8 + a dependent read of every
// accumulator - slow, certainly safe) and (ii) the exact integer result (the data are small integers: every product and sum is
// exact in fp16 / fp32).
// Output: one JSON line.  build: hipcc --offload-arch=gfx950 -O3 tools/mfma_war_repro.hip -o /tmp/mfma_war_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ h8v conv(int lane, int it, int which) {
    // small integers (|v| <= 7), different per lane, iteration and operand: products <= 49, sums over 16 x T stay exact in fp32
    h8v r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int v = ((lane * 7 + it * 13 + which * 5 + e * 3) % 15) - 7;
        r[e] = (_Float16)(float)v;                        // v_cvt: the VALU write into the operand register
    }
    return r;
}

template <int GUARD>
__global__ void __launch_bounds__(256) war_kernel(float* __restrict__ out, int iters, int lds_pad) {
    extern __shared__ char pad[];                         // occupancy control: more LDS per workgroup = fewer waves per SIMD
    (void)pad; (void)lds_pad;
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    f16v c00, c01, c10, c11;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c00[i] = 0.f; c01[i] = 0.f; c10[i] = 0.f; c11[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        const h8v Aeh = conv(lane, it, 0), Ael = conv(lane, it, 1), Aoh = conv(lane, it, 2), Aol = conv(lane, it, 3);
        const h8v Beh = conv(lane, it, 4), Bel = conv(lane, it, 5), Boh = conv(lane, it, 6), Bol = conv(lane, it, 7);
        if (GUARD) __builtin_amdgcn_sched_barrier(0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ael, Beh, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ael, Boh, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aol, Beh, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aol, Boh, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aeh, Bel, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aeh, Bol, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aoh, Bel, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aoh, Bol, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aeh, Beh, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aeh, Boh, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aoh, Beh, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aoh, Boh, c11, 0, 0, 0);
        if (GUARD == 1) {                                  // the production fence (cost65_device.hpp)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            asm volatile("" :: "v"(Aeh), "v"(Ael), "v"(Aoh), "v"(Aol), "v"(Beh), "v"(Bel), "v"(Boh), "v"(Bol));
            __builtin_amdgcn_sched_barrier(0);
        } else if (GUARD == 2) {                           // certainly safe: drain the matrix pipe
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            asm volatile("" :: "v"(Aeh), "v"(Ael), "v"(Aoh), "v"(Aol), "v"(Beh), "v"(Bel), "v"(Boh), "v"(Bol), "v"(c00), "v"(c01), "v"(c10), "v"(c11));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* o = out + ((size_t)gw * 64 + lane) * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[i] = c00[i]; o[16 + i] = c01[i]; o[32 + i] = c10[i]; o[48 + i] = c11[i]; }
}

// exact reference of one wave's accumulators on the host (32x32x16: A lane l holds row l % 32, k = 8 (l / 32) .. + 7; same for B
// with the column; C lane l, register i: column l % 32, row 8 (i / 4) + 4 (l / 32) ... = (i / 4) * 8 + (l / 32) * 4 + i % 4)
static int opval(int lane, int it, int which, int e) { return ((lane * 7 + it * 13 + which * 5 + e * 3) % 15) - 7; }
static void reference(int iters, std::vector<float>& ref) {
    ref.assign(64 * 64, 0.f);
    std::vector<double> acc(4 * 32 * 32, 0.0);          // c00, c01, c10, c11 as [row][col]
    const int pa[12] = {1, 1, 3, 3, 0, 0, 2, 2, 0, 0, 2, 2}, pb[12] = {4, 6, 4, 6, 5, 7, 5, 7, 4, 6, 4, 6}, pc[12] = {0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3};
    for (int it = 0; it < iters; ++it)
        for (int m = 0; m < 12; ++m)
            for (int r = 0; r < 32; ++r)
                for (int c = 0; c < 32; ++c) {
                    double s = 0;
                    for (int k = 0; k < 16; ++k) s += (double)opval(r + 32 * (k / 8), it, pa[m], k % 8) * opval(c + 32 * (k / 8), it, pb[m], k % 8);
                    acc[(pc[m] * 32 + r) * 32 + c] += s;
                }
    for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < 16; ++i) {
                const int row = (i / 4) * 8 + (l / 32) * 4 + (i % 4), col = l % 32;
                ref[l * 64 + q * 16 + i] = (float)acc[(q * 32 + row) * 32 + col];
            }
}

template <int GUARD>
static long run(int iters, int waves_per_simd, const std::vector<float>& ref, float* dout, std::vector<float>& host) {
    // 256 threads = 4 waves per workgroup = 1 per SIMD; waves_per_simd workgroups per CU fit when each takes 160 KB / waves_per_simd
    const int lds = waves_per_simd == 1 ? 100 * 1024 : waves_per_simd == 2 ? 70 * 1024 : 48 * 1024;
    hipFuncSetAttribute((const void*)war_kernel<GUARD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256 * waves_per_simd * 4;
    long bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(dout, 0, (size_t)blocks * 4 * 64 * 64 * 4);
        hipLaunchKernelGGL(war_kernel<GUARD>, dim3(blocks), dim3(256), lds, 0, dout, iters, lds);
        if (hipDeviceSynchronize() != hipSuccess) return -1;
        hipMemcpy(host.data(), dout, (size_t)blocks * 4 * 64 * 64 * 4, hipMemcpyDeviceToHost);
        for (size_t w = 0; w < (size_t)blocks * 4; ++w)
            for (int k = 0; k < 64 * 64; ++k) bad += host[w * 4096 + k] != ref[k];
    }
    return bad;
}

int main() {
    const int iters = 96;
    std::vector<float> ref;
    reference(iters, ref);
    const size_t max_out = (size_t)256 * 3 * 4 * 4 * 64 * 64;
    float* dout;
    if (hipMalloc((void**)&dout, max_out * 4) != hipSuccess) { printf("{\"error\": \"alloc\"}\n"); return 1; }
    std::vector<float> host(max_out);
    printf("{\"kernel\": \"12 x v_mfma_f32_32x32x16_f16 per iteration, operands rewritten by VALU conversions right behind them\", \"iters\": %d, \"launches_per_cell\": 3, \"wrong_elements\": {", iters);
    for (int wps = 1; wps <= 3; ++wps) {
        const long g0 = run<0>(iters, wps, ref, dout, host), g1 = run<1>(iters, wps, ref, dout, host), g2 = run<2>(iters, wps, ref, dout, host);
        printf("%s\"waves_per_simd_%d\": {\"no_fence\": %ld, \"production_fence\": %ld, \"drained\": %ld}", wps > 1 ? ", " : "", wps, g0, g1, g2);
    }
    printf("}}\n");
    return 0;
}
