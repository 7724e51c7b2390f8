// Issue cost of the instruction kinds the Sinkhorn sweeps are made of, on gfx950 (diagnostic binary, not
// part of the product):  hipcc --offload-arch=gfx950 -O2 tools/valu_cost.hip -o tools/valu_cost.bin
// Every kernel runs K iterations of a 64-instruction block of ONE kind (or a 32/32 mix) on 8 independent
// register sets, with W waves resident per SIMD (grid = 1024 * W single-wave workgroups); reported:
// shader cycles (s_memtime) per instruction per SIMD = wave cycles / (64 K) / W, and the wall-clock rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void __launch_bounds__(64) kern(float* out, int K, long long* cyc) {
    const int lane = threadIdx.x;
    f2v p[8], q[8];
    float s[8], t[8];
    f4 m[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int i = 0; i < 8; ++i) {
        p[i] = f2v{1.0f + lane * 1e-3f + i, 0.5f + i};
        q[i] = f2v{1.0f - 1e-6f * i, 1.0f - 2e-6f * i};
        s[i] = 1.0f + lane + i;
        t[i] = 0.999f + 1e-6f * i;
    }
    float one = 1.0f;
    asm volatile("" : "+v"(one));
    __shared__ float ldsbuf[1024];
    ldsbuf[lane] = lane;
    const unsigned ldsaddr = (unsigned)(lane * 16);
    f4 m4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned long long mask = 0x5555aaaa3333ccccull;
    asm volatile("" : "+s"(mask));
    const long long t0 = clock64();
    for (int it = 0; it < K; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if (MODE == 1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(s[i]) : "v"(t[i]));
                REP8(X)
#undef X
            } else if (MODE == 2) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(t[i]));
                REP8(X)
#undef X
            } else if (MODE == 3) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(s[i]) : "v"(t[i]));
                REP8(X)
#undef X
            } else if (MODE == 4) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(t[i]));
                REP8(X)
#undef X
            } else if (MODE == 5) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
                REP8(X)
#undef X
            } else if (MODE == 6) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[i]), "+v"(t[i]));
                REP8(X)
#undef X
            } else if (MODE == 7) {      // 4 pk_fma + 4 v_add alternating
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n\tv_add_f32 %1, %1, %3" : "+v"(p[i]), "+v"(s[i]) : "v"(q[i]), "v"(t[i]));
                X(0) X(1) X(2) X(3)
#undef X
            } else if (MODE == 8) {      // 4 pk_fma + 4 dpp adds alternating
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n\tv_add_f32_dpp %1, %3, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(p[i]), "+v"(s[i]) : "v"(q[i]), "v"(t[i]));
                X(0) X(1) X(2) X(3)
#undef X
            } else if (MODE == 9) {
#define X(i) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)" : "+v"(s[i]));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (MODE == 10) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %2, %3, %1"
                             : "+v"(m[0]), "+v"(m[1]) : "v"(s[0]), "v"(one));
            } else if (MODE == 11) {     // 7 pk_fma + 1 mfma 16x16x4
                asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n\tv_pk_fma_f32 %1, %1, %8, %1\n\tv_pk_fma_f32 %2, %2, %8, %2\n\t"
                             "v_pk_fma_f32 %3, %3, %8, %3\n\tv_pk_fma_f32 %4, %4, %8, %4\n\tv_pk_fma_f32 %5, %5, %8, %5\n\t"
                             "v_pk_fma_f32 %6, %6, %8, %6\n\tv_mfma_f32_16x16x4_f32 %7, %9, %10, %7"
                             : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(m[0])
                             : "v"(q[0]), "v"(s[0]), "v"(one));
            } else if (MODE == 12) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if (MODE == 13) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(t[i]));
                REP8(X)
#undef X
            } else if (MODE == 14) {     // 4 pk_fma + 4 ds_swizzle (LDS pipe beside the VALU)
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n\tds_swizzle_b32 %1, %1 offset:swizzle(SWAP,1)" : "+v"(p[i]), "+v"(s[i]) : "v"(q[i]));
                X(0) X(1) X(2) X(3)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (MODE == 17) {     // VOP3 select on an SGPR-pair mask (what hipcc emits for lane-pattern selects)
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(s[i]) : "v"(t[i]), "s"(mask));
                REP8(X)
#undef X
            } else if (MODE == 18) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(s[i]) : "v"(t[i]));
                REP8(X)
#undef X
            } else if (MODE == 19) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
                REP8(X)
#undef X
            } else if (MODE == 20) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s[i]) : "v"(t[i]));
                REP8(X)
#undef X
            } else if (MODE == 21) {     // 4 pk_fma + 4 plain selects
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(p[i]), "+v"(s[i]) : "v"(q[i]), "v"(t[i]));
                X(0) X(1) X(2) X(3)
#undef X
            } else if (MODE == 22) {
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(s[i]), "+v"(t[i]));
                REP8(X)
#undef X
            } else if (MODE == 23) {     // LDS broadcast-free b128 reads beside nothing
#define X(i) asm volatile("ds_read_b128 %0, %1" : "=v"(m4[i & 1]) : "v"(ldsaddr));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (MODE == 15) {     // dependent chain of v_pk_fma (latency)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n\tv_pk_fma_f32 %0, %0, %1, %0\n\tv_pk_fma_f32 %0, %0, %1, %0\n\t"
                             "v_pk_fma_f32 %0, %0, %1, %0\n\tv_pk_fma_f32 %0, %0, %1, %0\n\tv_pk_fma_f32 %0, %0, %1, %0\n\t"
                             "v_pk_fma_f32 %0, %0, %1, %0\n\tv_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[0]) : "v"(q[0]));
            } else if (MODE == 16) {     // dependent chain of dpp adds (latency incl. the required wait states)
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1" : "+v"(s[0]));
            }
        }
    }
    const long long t1 = clock64();
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y + s[i] + t[i];
    acc += m[0].x + m[1].y + m4[0].x + m4[1].y;
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, float* out, long long* cyc, int K) {
    for (int W : {1, 2, 3, 4, 8}) {
        const int grid = 1024 * W;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern<MODE>, dim3(grid), dim3(64), 0, 0, out, K / 8, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern<MODE>, dim3(grid), dim3(64), 0, 0, out, K, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(grid);
        hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= grid;
        const double n = (double)K * per_iter;
        printf("%-34s W=%d  %7.2f memtime-ticks/instr/wave  %6.2f ticks/instr/SIMD   wall %8.3f ms -> %6.2f ns/instr/SIMD\n", name, W,
               mean / n, mean / n / W, ms, ms * 1e6 / n / W);
    }
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&cyc, 8192 * 8);
    const int K = 4000;
    run<0>("v_pk_fma_f32", 64, out, cyc, K);
    run<1>("v_fma_f32", 64, out, cyc, K);
    run<2>("v_add_f32", 64, out, cyc, K);
    run<13>("v_mul_f32", 64, out, cyc, K);
    run<12>("v_pk_add_f32", 64, out, cyc, K);
    run<3>("v_add_f32_dpp", 64, out, cyc, K);
    run<4>("v_cndmask_b32", 64, out, cyc, K);
    run<5>("v_rcp_f32", 64, out, cyc, K);
    run<6>("v_permlane32_swap_b32", 64, out, cyc, K);
    run<7>("pk_fma + v_add 1:1", 64, out, cyc, K);
    run<8>("pk_fma + dpp add 1:1", 64, out, cyc, K);
    run<9>("ds_swizzle_b32", 64, out, cyc, K);
    run<14>("pk_fma + ds_swizzle 1:1", 64, out, cyc, K);
    run<10>("v_mfma_f32_16x16x4_f32", 64, out, cyc, K);
    run<11>("7 pk_fma + 1 mfma16x16x4", 64, out, cyc, K);
    run<17>("v_cndmask_b32_e64 (sgpr mask)", 64, out, cyc, K);
    run<18>("v_mov_b32_dpp", 64, out, cyc, K);
    run<19>("v_exp_f32", 64, out, cyc, K);
    run<20>("v_max_f32", 64, out, cyc, K);
    run<21>("pk_fma + v_cndmask 1:1", 64, out, cyc, K);
    run<22>("v_permlane16_swap_b32", 64, out, cyc, K);
    run<23>("ds_read_b128", 64, out, cyc, K);
    run<15>("pk_fma dependent chain", 64, out, cyc, K);
    run<16>("dpp add dependent chain (+s_nop 1)", 64, out, cyc, K);
    return 0;
}
