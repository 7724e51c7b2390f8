// Fine-level local expectation for PATS on gfx950.
//
// Replaces ThirdLayer.Compute_result (models/third_layer.py:184-217) and the match-label rule
// (models/third_layer.py:161-170): six gathers + two einsums over [P,16,25] index tensors in the
// reference; here one wave owns one 65x65 problem and walks its 16 centre rows (the 4x4 centre of
// the 8x8 source window).  Per row: argmax over the 64 real targets (first index on ties), a 5x5
// neighbourhood on the zero-padded (pad 2) 8x8 target map with weights sqrt(P + 1e-7) / scale
// (scale padded with 1e-2), expectation -> sub-pixel target; label from the argmax over all 65
// columns hitting the dustbin (outdoor) or a fixed cell pattern (indoor).
// kornia.create_meshgrid (un-vendored, kornia==0.5.5) is only an index grid here: (x, y) integer
// coordinates - restated inline.
#include "common.hpp"

namespace pats {

__global__ void __launch_bounds__(256)
compute_result_kernel(const float* __restrict__ scores, int input_is_log, int64_t P,
                      const float* __restrict__ scale_x, const float* __restrict__ scale_y,
                      const int64_t* __restrict__ p_s, const int64_t* __restrict__ p_t, int outdoor,
                      float* __restrict__ mk0, float* __restrict__ mk1, float* __restrict__ wl_raw,
                      float* __restrict__ label, uint8_t* __restrict__ ifm,
                      int* __restrict__ count) {
    constexpr int W = 8, T = 5, NN = 65;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 4 + wave;
    if (p >= P) return;
    const float* Sp = scores + p * (int64_t)NN * NN;
    const float sxl = scale_x[p * 64 + lane], syl = scale_y[p * 64 + lane];
    const float ps0 = (float)p_s[p * 2], ps1 = (float)p_s[p * 2 + 1];
    const float pt0 = (float)p_t[p * 2], pt1 = (float)p_t[p * 2 + 1];
    int local_count = 0;
    for (int q = 0; q < 16; ++q) {
        const int qy = q / 4 + 2, qx = q % 4 + 2;                    // [:, 2:6, 2:6]  (:186,188)
        const float* row = Sp + (int64_t)(qy * W + qx) * NN;
        float x = row[lane];
        float xd = row[64];                                           // dustbin column, uniform
        if (input_is_log) { x = expf(x); xd = expf(xd); }
        // argmax over the 64 real columns (:188) and over all 65 of (row + 1e-8) (:167-168)
        float bv = x; int bi = lane;
        wave_argmax(bv, bi);
        const int max0 = bi;
        float av = x + 1e-8f; int ai = lane;
        wave_argmax(av, ai);
        const bool matching = !((xd + 1e-8f) > av);                   // dustbin wins only if strictly larger
        const float rowsum = wave_sum(x) + xd;
        const int mx = max0 % W, my = max0 / W;
        // 5x5 taps: lane t < 25 owns tap (tx, ty)
        float fx = 0.f, fy = 0.f, posx = 0.f, posy = 0.f, sbv = 0.f;
        {
            const int t = lane < T * T ? lane : 0;
            const int tx = t % T, ty = t / T;
            const int ux = mx + tx - 2, uy = my + ty - 2;             // index3 on the pad-2 map (:189-191)
            const bool inside = ux >= 0 && ux < W && uy >= 0 && uy < W;
            const int src = inside ? uy * W + ux : 0;
            const float sb_in = __shfl(x, src);
            const float scx_in = __shfl(sxl, src), scy_in = __shfl(syl, src);
            if (lane < T * T) {
                sbv = inside ? sb_in : 0.0f;                          // ZeroPad2d(2)           (:185)
                const float scx = inside ? scx_in : 1e-2f;            // ConstantPad2d(2, 1e-2) (:195-196)
                const float scy = inside ? scy_in : 1e-2f;
                const float root = sqrtf(sbv + 1e-7f);
                fx = root / scx;                                      // :197-198
                fy = root / scy;
                posx = (float)tx * 2.0f - (float)(T - 1);             // meshgrid * 2 - (T - 1)  (:199)
                posy = (float)ty * 2.0f - (float)(T - 1);
            }
        }
        const float wpx = wave_sum(fx * posx), wpy = wave_sum(fy * posy);
        const float sumx = wave_sum(fx), sumy = wave_sum(fy), unfold = wave_sum(sbv);
        if (lane == 0) {
            const int64_t o = (p * 16 + q) * 2;
            const float m1x = wpx / sumx + ((float)mx + 0.5f - (float)W / 2) * 2.0f;   // :206
            const float m1y = wpy / sumy + ((float)my + 0.5f - (float)W / 2) * 2.0f;   // :207
            mk1[o + 0] = m1x + pt0;                                                     // :208
            mk1[o + 1] = m1y + pt1;
            mk0[o + 0] = ps0 + (float)(q % 4) * 2.0f - 3.0f;                            // :209-210
            mk0[o + 1] = ps1 + (float)(q / 4) * 2.0f - 3.0f;
            const float wl = rowsum - unfold;                                           // :213
            if (wl_raw) wl_raw[p * 16 + q] = wl;
            if (wl >= 1e-2f) local_count += 1;
            ifm[p * 16 + q] = matching ? 1 : 0;
            float l0 = 1e8f;                                                            // :161
            if (!outdoor) {
                const bool select = (q == 5 || q == 15 || q == 7 || q == 13);           // :163-166
                l0 = select ? l0 : -10.0f;
            } else {
                l0 = matching ? l0 : -10.0f;                                            // :169-170
            }
            label[o + 0] = l0;
            label[o + 1] = 1e8f;
        }
    }
    if (lane == 0 && count && local_count) atomicAdd(count, local_count);
}

// whole_loss = where(wl >= 1e-2, wl, 0) / (count + 10) / 10        (:215)
__global__ void __launch_bounds__(256)
whole_loss_finish_kernel(float* __restrict__ wl, int64_t n, const int* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float denom = (float)(*count) + 10.0f;
    const float x = wl[i];
    wl[i] = (x >= 1e-2f ? x : 0.0f) / denom / 10.0f;
}

}  // namespace pats

using namespace pats;

extern "C" int pats_compute_result_f32(const float* scores, int input_is_log, int64_t P,
                                       const float* scale_x, const float* scale_y,
                                       const int64_t* p_s, const int64_t* p_t, int outdoor,
                                       float* mkpts0_f, float* mkpts1_f, float* whole_loss,
                                       float* label, uint8_t* if_matching1, pats_stream_t stream) {
    PATS_REQUIRE(P >= 0, "compute_result: bad shape");
    if (P == 0) return PATS_OK;
    PATS_REQUIRE(scores && scale_x && scale_y && p_s && p_t && mkpts0_f && mkpts1_f && label &&
                     if_matching1, "compute_result: null pointer");
    hipStream_t st = as_stream(stream);
    // whole_loss needs one cross-problem count (:215): a 4-byte stream-ordered allocation owned
    // by this call (no device synchronisation).
    int* count = nullptr;
    if (whole_loss) {
        if (hipMallocAsync((void**)&count, sizeof(int), st) != hipSuccess) {
            set_error("compute_result: hipMallocAsync failed");
            return PATS_ERR_LAUNCH;
        }
        hipMemsetAsync(count, 0, sizeof(int), st);
    }
    hipLaunchKernelGGL(compute_result_kernel, dim3((unsigned)ceil_div(P, 4)), dim3(256), 0, st, scores,
                       input_is_log, P, scale_x, scale_y, p_s, p_t, outdoor, mkpts0_f, mkpts1_f,
                       whole_loss, label, if_matching1, count);
    int rc = check_launch("compute_result_kernel");
    if (whole_loss && rc == PATS_OK) {
        hipLaunchKernelGGL(whole_loss_finish_kernel, dim3((unsigned)ceil_div(P * 16, 256)), dim3(256),
                           0, st, whole_loss, P * 16, count);
        rc = check_launch("whole_loss_finish_kernel");
    }
    if (count) hipFreeAsync(count, st);
    return rc;
}
