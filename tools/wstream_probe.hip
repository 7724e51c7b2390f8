// How fast can every CU stream the SAME L2-resident weight block into registers?  (round 5: the fused fine-level GNN layer streams
// 2.5 MB of packed weight fragments per problem through each CU; round 4 measured ~24 bytes a clock per CU inside conv_pk_kernel.)
// One workgroup per CU (256), WAVES waves each; a wave reads its own 1 / WAVES slice of the block as 1 KB fragments (16 bytes a
// lane), DEPTH loads in flight, the whole block REPS times.  SHARED = 1: every wave reads the whole block (what a token-stationary
// layer would do: the CU's L1 absorbs the repeats or it does not).  LDSDMA = 1: global -> LDS direct (no registers).
// build: hipcc --offload-arch=gfx950 -O3 tools/wstream_probe.hip -o /tmp/wstream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f4v* gptr;

template <int WAVES, int DEPTH, bool SHARED, bool NT>
__global__ void __launch_bounds__(WAVES * 64, 1) stream_kernel(const f4v* __restrict__ w, int frags, int reps, float* out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = SHARED ? frags : frags / WAVES, f0 = SHARED ? 0 : wave * per;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        for (int f = 0; f < per; f += DEPTH) {
            f4v v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const f4v* p = w + (size_t)(f0 + ((f + d + (SHARED ? wave * 7 : 0)) % per)) * 64 + lane;
                v[d] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d];
        }
    }
    if (acc.x == 12345.f) out[0] = acc.y;
}

// global -> LDS direct, 16 bytes a lane (gfx950: global_load_lds_dwordx4); a ring of DEPTH 1 KB slots per wave
template <int WAVES, int DEPTH>
__global__ void __launch_bounds__(WAVES * 64, 1) stream_lds_kernel(const f4v* __restrict__ w, int frags, int reps, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = frags / WAVES, f0 = wave * per;
    float s = 0.f;
    for (int r = 0; r < reps; ++r) {
        for (int f = 0; f < per; f += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const f4v* p = w + (size_t)(f0 + f + d) * 64 + lane;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(lds + (wave * DEPTH + d) * 1024), 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0070);       // vmcnt(0)
            s += *(const float*)(lds + (wave * DEPTH) * 1024 + lane * 4);
        }
    }
    if (s == 12345.f) out[0] = s;
}

template <typename F>
static void timed(const char* name, size_t bytes_per_cu, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    const double per_cu = (double)bytes_per_cu / (best * 1e-3);
    printf("%-52s %8.3f ms  %7.1f GB/s per CU = %5.1f B/clk @2.4 GHz (%5.1f @1.95)  aggregate %6.2f TB/s\n", name, best, per_cu / 1e9,
           per_cu / 2.4e9, per_cu / 1.95e9, per_cu * 256 / 1e12);
}

int main() {
    const size_t bytes = (size_t)2560 * 1024;          // 2.5 MB: the fine-level layer's packed weights
    const int frags = (int)(bytes / 1024), reps = 32;
    f4v* buf; float* out;
    if (hipMalloc((void**)&buf, bytes) != hipSuccess || hipMalloc((void**)&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    const size_t own = bytes * reps;
#define RUN(W, D, S, N) timed("waves " #W " depth " #D " shared " #S " nt " #N, S ? own * W : own, [&] { \
        hipLaunchKernelGGL((stream_kernel<W, D, S, N>), dim3(256), dim3(W * 64), 0, 0, buf, frags, reps, out); })
    RUN(8, 2, false, false);
    RUN(8, 4, false, false);
    RUN(8, 8, false, false);
    RUN(8, 16, false, false);
    RUN(16, 4, false, false);
    RUN(16, 8, false, false);
    RUN(4, 8, false, false);
    RUN(4, 16, false, false);
    RUN(8, 8, false, true);
    RUN(8, 4, true, false);
    RUN(8, 8, true, false);
    hipFuncSetAttribute((const void*)stream_lds_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)stream_lds_kernel<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)stream_lds_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    timed("global -> LDS direct, waves 8 depth 4", own, [&] { hipLaunchKernelGGL((stream_lds_kernel<8, 4>), dim3(256), dim3(512), 65536, 0, buf, frags, reps, out); });
    timed("global -> LDS direct, waves 8 depth 8", own, [&] { hipLaunchKernelGGL((stream_lds_kernel<8, 8>), dim3(256), dim3(512), 65536, 0, buf, frags, reps, out); });
    timed("global -> LDS direct, waves 4 depth 8", own, [&] { hipLaunchKernelGGL((stream_lds_kernel<4, 8>), dim3(256), dim3(256), 65536, 0, buf, frags, reps, out); });
    // the same with a block 16x larger than the L2 share (40 MB: Infinity Cache)
    const size_t big = (size_t)40 << 20;
    f4v* buf2;
    if (hipMalloc((void**)&buf2, big) == hipSuccess) {
        hipMemset(buf2, 0, big);
        const int frags2 = (int)(big / 1024);
        timed("waves 8 depth 8 own slice, 40 MB block", big * 4, [&] { hipLaunchKernelGGL((stream_kernel<8, 8, false, false>), dim3(256), dim3(512), 0, 0, buf2, frags2, 4, out); });
    }
    return 0;
}
