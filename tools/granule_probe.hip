// What does a partial read of a 128-byte line cost?  (round 4: the two NCHW gathers read 32-byte runs - a window row, a pooled tap
// pair - out of 128-byte lines; their PMC traffic is 1.5-2.7x their algorithmic bytes.)  Every kernel reads RUN = 32 bytes (eight
// lanes x 4 bytes, as the gathers do) at every STRIDE bytes of a 4 GiB span (16x the Infinity Cache), plain or non-temporal;
// reported: ms, useful GB/s, span GB/s.  If span GB/s of stride 128 exceeds the ~6.3 TB/s streaming rate, the memory side fetches
// less than the line.
// build: hipcc --offload-arch=gfx950 -O3 tools/granule_probe.hip -o /tmp/granule_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int STRIDE, int RUNL, bool NT>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ p, size_t sectors, float* out) {
    // RUNL lanes x 4 bytes per run
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    const size_t per = nthreads / RUNL;               // runs visited per sweep of the grid
    const size_t run0 = tid / RUNL;
    const int l = (int)(tid % RUNL);
    float s = 0.f;
    for (size_t r = run0; r < sectors; r += per * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t rr = r + (size_t)u * per;
            const float* q = p + (rr < sectors ? rr : run0) * (STRIDE / 4) + l;
            v[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    if (s == 12345.f) out[0] = s;
}
template <int STRIDE, int RUNL, bool NT>
static void run(const char* name, const float* buf, size_t span, float* out) {
    const size_t sectors = span / STRIDE;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<STRIDE, RUNL, NT>), dim3(256 * 8), dim3(256), 0, 0, buf, sectors, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    const double useful = (double)sectors * RUNL * 4;
    printf("%-34s stride %4d run %3d B %s: %8.3f ms  useful %7.1f GB/s  span %8.1f GB/s\n", name, STRIDE, RUNL * 4, NT ? "nt   " : "plain",
           best, useful / best / 1e6, (double)span / best / 1e6);
}
int main() {
    const size_t span = (size_t)4 << 30;
    float *buf, *out;
    if (hipMalloc((void**)&buf, span) != hipSuccess || hipMalloc((void**)&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, span);
    run<32, 8, false>("dense (every 32-byte run)", buf, span, out);
    run<32, 8, true>("dense (every 32-byte run)", buf, span, out);
    run<64, 8, false>("32 of every 64", buf, span, out);
    run<64, 8, true>("32 of every 64", buf, span, out);
    run<128, 8, false>("32 of every 128", buf, span, out);
    run<128, 8, true>("32 of every 128", buf, span, out);
    run<256, 8, false>("32 of every 256", buf, span, out);
    run<256, 8, true>("32 of every 256", buf, span, out);
    run<128, 16, false>("64 of every 128", buf, span, out);
    run<128, 16, true>("64 of every 128", buf, span, out);
    run<256, 16, false>("64 of every 256", buf, span, out);
    run<256, 32, false>("128 of every 256", buf, span, out);
    run<512, 32, false>("128 of every 512", buf, span, out);
    return 0;
}
