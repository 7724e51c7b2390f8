// Device-side body of ThirdLayer.Compute_result + the match label for ONE 65x65 problem, executed
// by one wave (models/third_layer.py:161-170,184-217).  `Sp` may point to global memory or LDS and
// holds the plan (or the log-plan when input_is_log) with row stride 65.  Shared by
// compute_result_kernel (third.hip) and the fused third-level kernel (sinkhorn.hip).
#pragma once
#include "common.hpp"

namespace pats {

struct ComputeResultOut {
    float* mk0;            // [P,16,2]
    float* mk1;            // [P,16,2]
    float* wl_raw;         // [P,16] or null
    float* label;          // [P*16,2]
    uint8_t* ifm;          // [P,16]
    int* count;            // global count of whole_loss >= 1e-2, or null
};

// arguments of the register-block fused third-level kernel (third_fused.hip)
struct Fused65Args {
    const float* d0;
    const float* d1;
    int D;
    int64_t P;
    const float* ns;             // [P,64] target areas
    const float* one;
    int iters, linear;
    const float* scale_x;
    const float* scale_y;
    const int64_t* p_s;
    const int64_t* p_t;
    int outdoor;
    ComputeResultOut cr;
    int stagger;
    unsigned long long* fallbacks;   // guard-trip counter or null
    int scan;                        // third_fused_kernel: re-solve only the problems flagged THIRD_REDO
    const int64_t* P_dev;            // throughput mode: the launch covers the capacity P, *P_dev problems exist (or null)
    int fingerprint;                 // libpats_amd_diag.so only: leave a fingerprint of the score matrix in label[p*16][1]
    int lds_poison_on;               // libpats_amd_diag.so only: fill the workgroup's LDS with lds_poison first
    unsigned lds_poison;
    int reverse_blocks;              // libpats_amd_diag.so only: workgroup b solves problem P - 1 - b (round-6 first-launch experiment)
};
// number of problems that exist: min(capacity, device-side count)
__device__ __forceinline__ int64_t live_problems(const Fused65Args& g) {
    if (!g.P_dev) return g.P;
    const int64_t n = *g.P_dev;
    return n < g.P ? n : g.P;
}
constexpr uint8_t THIRD_REDO = 0xEE; // if_matching1[p*16] of a problem the log-domain kernel must redo
int launch_third_fused(const Fused65Args& g, hipStream_t st);

// row-level (16-lane) all-reduces: 4 DPP steps, no LDS traffic
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<DPP_QUAD_XOR1>(v);
    v += dpp_f<DPP_QUAD_XOR2>(v);
    v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f<DPP_ROW_MIRROR>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ void argmax_step(float& v, int& i) {
    const float ov = dpp_f<CTRL>(v);
    const int oi = dpp_i<CTRL>(i);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }     // first index wins ties
}
__device__ __forceinline__ void row16_argmax(float& v, int& i) {
    argmax_step<DPP_QUAD_XOR1>(v, i);
    argmax_step<DPP_QUAD_XOR2>(v, i);
    argmax_step<DPP_ROW_HALF_MIRROR>(v, i);
    argmax_step<DPP_ROW_MIRROR>(v, i);
}

// One wave, one problem.  The 16 centre rows are processed four at a time: each 16-lane DPP row
// of the wave owns one centre row (lane t of the group holds targets t, t+16, t+32, t+48), so the
// argmax / sums are 4-step row-level DPP reductions and the 5x5 taps are two per lane, gathered
// from the plan by address.  sx / sy are indexable [64] scale vectors (global or LDS).
// `compact_stride` == 0: Sp is the full 65 x 65 matrix (row stride 65); > 0: Sp holds only the 16
// centre rows, row q at Sp + q * compact_stride; < 0: see below.
__device__ __forceinline__ void compute_result_problem(const float* Sp, int input_is_log, int64_t p,
                                                       const float* sx, const float* sy, float ps0,
                                                       float ps1, float pt0, float pt1, int outdoor,
                                                       const ComputeResultOut& o, int lane,
                                                       int compact_stride = 0, bool scale_is_area = false) {
    constexpr int W = 8, T = 5, NN = 65;
    const int grp = lane >> 4, t = lane & 15;
    int local_count = 0;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const int q = pass * 4 + grp;
        const int qy = q / 4 + 2, qx = q % 4 + 2;                    // [:, 2:6, 2:6]  (:186,188)
        // compact_stride > 0: row q at q * stride; < 0: the four block rows 2..5 with all eight local rows each,
        // row (qy - 2) * 8 + qx at stride -compact_stride (third_fused3.hip)
        const float* row = compact_stride > 0 ? Sp + q * compact_stride
                         : compact_stride < 0 ? Sp + ((qy - 2) * 8 + qx) * (-compact_stride)
                                              : Sp + (qy * W + qx) * NN;
        float x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = row[t + 16 * k];
            if (input_is_log) x[k] = expf(x[k]);
        }
        float xd = row[64];                                           // dustbin column
        if (input_is_log) xd = expf(xd);
        // argmax over the 64 real columns (:188); the all-column argmax of (row + 1e-8) (:167-168)
        // hits the dustbin only if it is strictly larger than every real entry
        float bv = x[0];
        int bi = t;
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (x[k] > bv) { bv = x[k]; bi = t + 16 * k; }
        row16_argmax(bv, bi);
        const int max0 = bi;
        const bool matching = !((xd + 1e-8f) > (bv + 1e-8f));
        const float rowsum = row16_sum((x[0] + x[1]) + (x[2] + x[3])) + xd;
        const int mx = max0 % W, my = max0 / W;
        // 5x5 taps, two per lane (tap ids t and t + 16)
        float wpx = 0.f, wpy = 0.f, sumx = 0.f, sumy = 0.f, unfold = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tap = t + 16 * h;
            if (tap < T * T) {
                const int tx = tap % T, ty = tap / T;
                const int ux = mx + tx - 2, uy = my + ty - 2;         // index3 on the pad-2 map (:189-191)
                const bool inside = ux >= 0 && ux < W && uy >= 0 && uy < W;
                const int src = inside ? uy * W + ux : 0;
                float sbv = row[src];
                if (input_is_log) sbv = expf(sbv);
                sbv = inside ? sbv : 0.0f;                            // ZeroPad2d(2)           (:185)
                // scale_is_area: sx = sy = the OT's target areas and scale_x = scale_y = sqrt(area + 1e-8) is formed here
                // (third_layer.py:153-154) instead of by the caller
                float scx = inside ? sx[src] : 1e-2f;                 // ConstantPad2d(2, 1e-2) (:195-196)
                float scy = inside ? sy[src] : 1e-2f;
                if (scale_is_area && inside) scx = scy = sqrtf(scx + 1e-8f);
                const float root = sqrtf(sbv + 1e-7f);
                const float fx = root / scx, fy = root / scy;         // :197-198
                wpx += fx * ((float)tx * 2.0f - (float)(T - 1));      // meshgrid * 2 - (T - 1)  (:199)
                wpy += fy * ((float)ty * 2.0f - (float)(T - 1));
                sumx += fx;
                sumy += fy;
                unfold += sbv;
            }
        }
        wpx = row16_sum(wpx); wpy = row16_sum(wpy);
        sumx = row16_sum(sumx); sumy = row16_sum(sumy); unfold = row16_sum(unfold);
        if (t == 0) {
            const int64_t oo = (p * 16 + q) * 2;
            const float m1x = wpx / sumx + ((float)mx + 0.5f - (float)W / 2) * 2.0f;   // :206
            const float m1y = wpy / sumy + ((float)my + 0.5f - (float)W / 2) * 2.0f;   // :207
            o.mk1[oo + 0] = m1x + pt0;                                                  // :208
            o.mk1[oo + 1] = m1y + pt1;
            o.mk0[oo + 0] = ps0 + (float)(q % 4) * 2.0f - 3.0f;                         // :209-210
            o.mk0[oo + 1] = ps1 + (float)(q / 4) * 2.0f - 3.0f;
            const float wl = rowsum - unfold;                                           // :213
            if (o.wl_raw) o.wl_raw[p * 16 + q] = wl;
            if (wl >= 1e-2f) local_count += 1;
            o.ifm[p * 16 + q] = matching ? 1 : 0;
            float l0 = 1e8f;                                                            // :161
            if (!outdoor) {
                const bool select = (q == 5 || q == 15 || q == 7 || q == 13);           // :163-166
                l0 = select ? l0 : -10.0f;
            } else {
                l0 = matching ? l0 : -10.0f;                                            // :169-170
            }
            o.label[oo + 0] = l0;
            o.label[oo + 1] = 1e8f;
        }
    }
    if (o.count) {
        // lanes with t == 0 (4 per wave) hold the counts
        int c = local_count;
        c += __shfl_xor(c, 16);
        c += __shfl_xor(c, 32);
        if (lane == 0 && c) atomicAdd(o.count, c);
    }
}

}  // namespace pats
