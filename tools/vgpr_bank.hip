// v_pk_fma_f32 / v_fma_f32 issue cost against the VGPR numbers of its three sources (register-bank conflicts),
// gfx950.  Diagnostic binary: hipcc --offload-arch=gfx950 -O2 tools/vgpr_bank.hip -o tools/vgpr_bank.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

// 8 independent accumulators A0..A7; operand registers are given as text
#define BODY8(A, K, V)                                                        \
    "v_pk_fma_f32 v[" A "+0:" A "+1], v[" K "+0:" K "+1], v[" V ":" V "+1], v[" A "+0:" A "+1]\n\t"

template <int MODE>
__global__ void __launch_bounds__(64) kern(float* out, int K, long long* cyc) {
    const long long t0 = clock64();
    for (int it = 0; it < K; ++it) {
        if (MODE == 0)        // acc {0,1}+even spacing, K {2,3}, vec {0,1}: acc and vec share banks
            asm volatile(
                "v_pk_fma_f32 v[8:9], v[42:43], v[100:101], v[8:9]\n\tv_pk_fma_f32 v[12:13], v[46:47], v[100:101], v[12:13]\n\t"
                "v_pk_fma_f32 v[16:17], v[50:51], v[100:101], v[16:17]\n\tv_pk_fma_f32 v[20:21], v[54:55], v[100:101], v[20:21]\n\t"
                "v_pk_fma_f32 v[24:25], v[58:59], v[100:101], v[24:25]\n\tv_pk_fma_f32 v[28:29], v[62:63], v[100:101], v[28:29]\n\t"
                "v_pk_fma_f32 v[32:33], v[66:67], v[100:101], v[32:33]\n\tv_pk_fma_f32 v[36:37], v[70:71], v[100:101], v[36:37]\n\t" ::: "memory", "v103", "s21");
        else if (MODE == 1)   // acc {0,1}, K {2,3}, vec {2,3}: K and vec share banks
            asm volatile(
                "v_pk_fma_f32 v[8:9], v[42:43], v[102:103], v[8:9]\n\tv_pk_fma_f32 v[12:13], v[46:47], v[102:103], v[12:13]\n\t"
                "v_pk_fma_f32 v[16:17], v[50:51], v[102:103], v[16:17]\n\tv_pk_fma_f32 v[20:21], v[54:55], v[102:103], v[20:21]\n\t"
                "v_pk_fma_f32 v[24:25], v[58:59], v[102:103], v[24:25]\n\tv_pk_fma_f32 v[28:29], v[62:63], v[102:103], v[28:29]\n\t"
                "v_pk_fma_f32 v[32:33], v[66:67], v[102:103], v[32:33]\n\tv_pk_fma_f32 v[36:37], v[70:71], v[102:103], v[36:37]\n\t" ::: "memory", "v103", "s21");
        else if (MODE == 2)   // all three in banks {0,1}
            asm volatile(
                "v_pk_fma_f32 v[8:9], v[40:41], v[100:101], v[8:9]\n\tv_pk_fma_f32 v[12:13], v[44:45], v[100:101], v[12:13]\n\t"
                "v_pk_fma_f32 v[16:17], v[48:49], v[100:101], v[16:17]\n\tv_pk_fma_f32 v[20:21], v[52:53], v[100:101], v[20:21]\n\t"
                "v_pk_fma_f32 v[24:25], v[56:57], v[100:101], v[24:25]\n\tv_pk_fma_f32 v[28:29], v[60:61], v[100:101], v[28:29]\n\t"
                "v_pk_fma_f32 v[32:33], v[64:65], v[100:101], v[32:33]\n\tv_pk_fma_f32 v[36:37], v[68:69], v[100:101], v[36:37]\n\t" ::: "memory", "v103", "s21");
        else if (MODE == 3)   // two distinct operands only (the form tools/valu_cost.hip times)
            asm volatile(
                "v_pk_fma_f32 v[8:9], v[8:9], v[100:101], v[8:9]\n\tv_pk_fma_f32 v[12:13], v[12:13], v[100:101], v[12:13]\n\t"
                "v_pk_fma_f32 v[16:17], v[16:17], v[100:101], v[16:17]\n\tv_pk_fma_f32 v[20:21], v[20:21], v[100:101], v[20:21]\n\t"
                "v_pk_fma_f32 v[24:25], v[24:25], v[100:101], v[24:25]\n\tv_pk_fma_f32 v[28:29], v[28:29], v[100:101], v[28:29]\n\t"
                "v_pk_fma_f32 v[32:33], v[32:33], v[100:101], v[32:33]\n\tv_pk_fma_f32 v[36:37], v[36:37], v[100:101], v[36:37]\n\t" ::: "memory", "v103", "s21");
        else if (MODE == 4)   // scalar fma, three distinct registers in banks 0, 1, 2
            asm volatile(
                "v_fma_f32 v8, v41, v102, v8\n\tv_fma_f32 v12, v45, v102, v12\n\tv_fma_f32 v16, v49, v102, v16\n\tv_fma_f32 v20, v53, v102, v20\n\t"
                "v_fma_f32 v24, v57, v102, v24\n\tv_fma_f32 v28, v61, v102, v28\n\tv_fma_f32 v32, v65, v102, v32\n\tv_fma_f32 v36, v69, v102, v36\n\t" ::: "memory", "v103", "s21");
        else if (MODE == 5)   // scalar fma, all three in bank 0
            asm volatile(
                "v_fma_f32 v8, v40, v100, v8\n\tv_fma_f32 v12, v44, v100, v12\n\tv_fma_f32 v16, v48, v100, v16\n\tv_fma_f32 v20, v52, v100, v20\n\t"
                "v_fma_f32 v24, v56, v100, v24\n\tv_fma_f32 v28, v60, v100, v28\n\tv_fma_f32 v32, v64, v100, v32\n\tv_fma_f32 v36, v68, v100, v36\n\t" ::: "memory", "v103", "s21");
        else if (MODE == 6)   // v_fmac_f32 (VOP2: dst = src2), K bank 1, vec bank 2, acc bank 0
            asm volatile(
                "v_fmac_f32 v8, v41, v102\n\tv_fmac_f32 v12, v45, v102\n\tv_fmac_f32 v16, v49, v102\n\tv_fmac_f32 v20, v53, v102\n\t"
                "v_fmac_f32 v24, v57, v102\n\tv_fmac_f32 v28, v61, v102\n\tv_fmac_f32 v32, v65, v102\n\tv_fmac_f32 v36, v69, v102\n\t" ::: "memory", "v103", "s21");
        else if (MODE == 7)   // pk_fma with the vector operand in SGPRs (constant bus), acc {0,1}, K {2,3}
            asm volatile(
                "v_pk_fma_f32 v[8:9], v[42:43], s[20:21], v[8:9]\n\tv_pk_fma_f32 v[12:13], v[46:47], s[20:21], v[12:13]\n\t"
                "v_pk_fma_f32 v[16:17], v[50:51], s[20:21], v[16:17]\n\tv_pk_fma_f32 v[20:21], v[54:55], s[20:21], v[20:21]\n\t"
                "v_pk_fma_f32 v[24:25], v[58:59], s[20:21], v[24:25]\n\tv_pk_fma_f32 v[28:29], v[62:63], s[20:21], v[28:29]\n\t"
                "v_pk_fma_f32 v[32:33], v[66:67], s[20:21], v[32:33]\n\tv_pk_fma_f32 v[36:37], v[70:71], s[20:21], v[36:37]\n\t" ::: "memory", "v103", "s21");
        else if (MODE == 8)   // pk_mul-free form: acc {0,1}, K {2,3}, vec {0,1} but op_sel broadcast of one half
            asm volatile(
                "v_pk_fma_f32 v[8:9], v[42:43], v[100:101], v[8:9] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[12:13], v[46:47], v[100:101], v[12:13] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 v[16:17], v[50:51], v[100:101], v[16:17] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[20:21], v[54:55], v[100:101], v[20:21] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 v[24:25], v[58:59], v[100:101], v[24:25] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[28:29], v[62:63], v[100:101], v[28:29] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 v[32:33], v[66:67], v[100:101], v[32:33] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[36:37], v[70:71], v[100:101], v[36:37] op_sel_hi:[1,0,1]\n\t" ::: "memory", "v103", "s21");
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (K < 0) out[0] = 1.f;
}

// the asm above names v8..v103 / s20..s21 without declaring them: reserve them by making the kernel look big
template <int MODE>
void run(const char* name, float* out, long long* cyc, int K) {
    for (int W : {1, 2, 4}) {
        const int grid = 1024 * W;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kern<MODE>, dim3(grid), dim3(64), 0, 0, out, K / 8, cyc);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern<MODE>, dim3(grid), dim3(64), 0, 0, out, K, cyc);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s W=%d  wall %8.3f ms -> %6.2f ns/instr/SIMD\n", name, W, ms, ms * 1e6 / ((double)K * 8) / W);
    }
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1024); (void)hipMalloc(&cyc, 8192 * 8);
    const int K = 40000;
    run<3>("pk_fma a,a,v,a (2 distinct operands)", out, cyc, K);
    run<0>("pk_fma acc{0,1} K{2,3} vec{0,1}", out, cyc, K);
    run<1>("pk_fma acc{0,1} K{2,3} vec{2,3}", out, cyc, K);
    run<2>("pk_fma acc{0,1} K{0,1} vec{0,1}", out, cyc, K);
    run<8>("pk_fma acc{0,1} K{2,3} vec{0,1} op_sel bcast", out, cyc, K);
    run<7>("pk_fma acc{0,1} K{2,3} vec SGPR pair", out, cyc, K);
    run<4>("v_fma banks 0,1,2", out, cyc, K);
    run<5>("v_fma banks 0,0,0", out, cyc, K);
    run<6>("v_fmac banks acc0 K1 vec2", out, cyc, K);
    return 0;
}
