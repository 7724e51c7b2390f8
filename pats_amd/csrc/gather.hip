// Descriptor gathers feeding the fine / third-level cost builds, for PATS on gfx950.
//
//   fine level   SecondLayer.forward, models/second_layer.py:71-86: AvgPool2d(2,1,1) on the two
//                high-resolution maps, sample the 12x12 grid at ((pos + 0.5) * stride) for strides
//                4, 2, 1, concat 64 + 64 + 128 channels, prepend the 8-channel "title", append the
//                dustbin feature column  ->  desc [2, B, 264, 145]
//   third level  ThirdLayer.forward, models/third_layer.py:121-146: round the coarse points to the
//                4-px lattice, gather the 8x8 window of the padded 52x52 half-resolution map around
//                each point in both crops, add the keypoint encoding, append the dustbin feature
//                -> feat_unfold [P, 128, 65] x 2 (the layout the MFMA cost build consumes)
//
// Both are pure index arithmetic + copies (HBM-bound on the output write).  Output rows are written
// by consecutive lanes; source windows are short contiguous runs of a feature-map row (L2-resident).
#include "common.hpp"

namespace pats {

// ---- fine level ------------------------------------------------------------------------------
// grid: (2*B, 264 / 8); block 256 = 8 channels x 32 lanes... simpler: one workgroup per (s*B + b),
// threads stride over the 264 x 145 outputs of that descriptor block (consecutive lanes ->
// consecutive points of one channel).
__global__ void __launch_bounds__(256)
fine_desc_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                 const float* __restrict__ f2, const float* __restrict__ title,
                 const float* __restrict__ rubbish, int64_t B, float* __restrict__ desc) {
    const int64_t n = blockIdx.x;              // s * B + b : index into the stacked maps (left crops first)
    const int64_t b = n % B;
    float* o = desc + n * 264 * 145;
    for (int e = threadIdx.x; e < 264 * 145; e += 256) {
        const int ch = e / 145, p = e - ch * 145;
        float v;
        if (p == 144) {
            v = rubbish[b * 264 + ch];                                   // second_layer.py:83,85
        } else if (ch < 8) {
            v = title[b * 8 + ch];                                       // :82,84
        } else {
            const int r = p / 12, c = p - r * 12;                        // positions (k // 12, k % 12)
            if (ch < 72) {            // map 0: [.,64,48,48], avgpool -> 49x49, index (4r+2, 4c+2)   :73-79
                const float* m = f0 + (n * 64 + (ch - 8)) * 48 * 48;
                const int y = 4 * r + 1, x = 4 * c + 1;                  // window rows y..y+1, cols x..x+1
                v = (((m[y * 48 + x] + m[y * 48 + x + 1]) + m[(y + 1) * 48 + x]) + m[(y + 1) * 48 + x + 1]) / 4.0f;
            } else if (ch < 136) {    // map 1: [.,64,24,24], avgpool -> 25x25, index (2r+1, 2c+1)
                const float* m = f1 + (n * 64 + (ch - 72)) * 24 * 24;
                const int y = 2 * r, x = 2 * c;
                v = (((m[y * 24 + x] + m[y * 24 + x + 1]) + m[(y + 1) * 24 + x]) + m[(y + 1) * 24 + x + 1]) / 4.0f;
            } else {                  // map 2: [.,128,12,12], no pooling, index (r, c)
                v = f2[(n * 128 + (ch - 136)) * 144 + p];
            }
        }
        o[e] = v;
    }
}

// ---- third level -----------------------------------------------------------------------------
__device__ __forceinline__ long long round_half_even_div(float x, float d) {
    return (long long)rintf(x / d);      // torch.round = round half to even = rintf in the default mode
}

// one workgroup (256 threads = 4 waves) per point; wave w handles channels w, w+4, ...; lane = window cell
__global__ void __launch_bounds__(256)
third_desc_kernel(const float* __restrict__ ff0, const float* __restrict__ ff1,
                  const float* __restrict__ mk0, const float* __restrict__ mk1,
                  const int64_t* __restrict__ b_ids, const float* __restrict__ kenc,
                  const float* __restrict__ rubbish, int64_t P, int64_t B,
                  float* __restrict__ out0, float* __restrict__ out1, int64_t* __restrict__ ps_out,
                  int64_t* __restrict__ pt_out, const int64_t* __restrict__ P_dev) {
    constexpr int W = 8, M = 52, C = 128;
    const int64_t p = blockIdx.x;
    if (P_dev && p >= *P_dev) return;        // throughput mode: the launch covers the capacity, the count is on the device
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = b_ids[p];
    // mkpts0_c = round(mkpts0_c / 4) * 4                                    third_layer.py:124
    const long long s0 = round_half_even_div(mk0[p * 2 + 0], 4.0f) * 4, s1 = round_half_even_div(mk0[p * 2 + 1], 4.0f) * 4;
    // mkpts1_c clamped to [0, 96] then rounded the same way                  :128-130
    float t0 = mk1[p * 2 + 0], t1 = mk1[p * 2 + 1];
    t0 = t0 >= 96.f ? 96.f : t0; t1 = t1 >= 96.f ? 96.f : t1;
    t0 = t0 <= 0.f ? 0.f : t0;   t1 = t1 <= 0.f ? 0.f : t1;
    const long long q0 = round_half_even_div(t0, 4.0f) * 4, q1 = round_half_even_div(t1, 4.0f) * 4;
    if (threadIdx.x == 0) {
        if (ps_out) { ps_out[p * 2] = s0; ps_out[p * 2 + 1] = s1; }
        if (pt_out) { pt_out[p * 2] = q0; pt_out[p * 2 + 1] = q1; }
    }
    const int wx = lane % W, wy = lane / W;
    auto fdiv2 = [](long long v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };      // python floor division
    // x = mk[:,0] // 2 + wx - W/2 + 2 ; y = mk[:,1] // 2 + wy - W/2 + 2     :125-126,131-132
    long long x0 = fdiv2(s0) + wx - W / 2 + 2, y0 = fdiv2(s1) + wy - W / 2 + 2;
    long long x1 = fdiv2(q0) + wx - W / 2 + 2, y1 = fdiv2(q1) + wy - W / 2 + 2;
    long long i0 = b * M * M + y0 * M + x0, i1 = b * M * M + y1 * M + x1;    // :127,133 (rows of the NHWC view)
    const long long lim = B * M * M - 1;
    i0 = i0 < 0 ? 0 : (i0 > lim ? lim : i0);      // memory safety (torch.gather would raise out of range)
    i1 = i1 < 0 ? 0 : (i1 > lim ? lim : i1);
    // dustbin feature: rubbish[b, :, y2*12 + x2], x2 = round(mk0x / 8), y2 = round(mk0y / 8)   :141-144
    long long x2 = round_half_even_div((float)s0, 8.0f), y2 = round_half_even_div((float)s1, 8.0f);
    long long i2 = y2 * 12 + x2;
    i2 = i2 < 0 ? 0 : (i2 > 143 ? 143 : i2);
    // NHWC row index -> (batch, y, x) of the NCHW map
    const long long bb0 = i0 / (M * M), r0 = i0 - bb0 * (M * M);
    const long long bb1 = i1 / (M * M), r1 = i1 - bb1 * (M * M);
    float* o0 = out0 + p * C * 65;
    float* o1 = out1 + p * C * 65;
    for (int ch = wave; ch < C; ch += 4) {
        const float ke = kenc[ch * 64 + lane];                                   // + self.kenc(kpts)   :139-140
        o0[ch * 65 + lane] = ff0[(bb0 * C + ch) * (M * M) + r0] + ke;
        o1[ch * 65 + lane] = ff1[(bb1 * C + ch) * (M * M) + r1] + ke;
        if (lane == 0) {
            const float rb = rubbish[(b * C + ch) * 144 + i2];
            o0[ch * 65 + 64] = rb;                                               // :145-146
            o1[ch * 65 + 64] = rb;
        }
    }
}

}  // namespace pats

using namespace pats;

extern "C" int pats_fine_descriptors_f32(const float* feat0, const float* feat1, const float* feat2,
                                         const float* title, const float* rubbish, int64_t B,
                                         float* desc, pats_stream_t stream) {
    PATS_REQUIRE(B >= 0, "fine_descriptors: bad shape");
    if (B == 0) return PATS_OK;
    PATS_REQUIRE(feat0 && feat1 && feat2 && title && rubbish && desc, "fine_descriptors: null pointer");
    hipLaunchKernelGGL(fine_desc_kernel, dim3((unsigned)(2 * B)), dim3(256), 0, as_stream(stream), feat0, feat1,
                       feat2, title, rubbish, B, desc);
    return check_launch("fine_desc_kernel");
}

extern "C" int pats_third_descriptors_f32(const float* feat_f0, const float* feat_f1,
                                          const float* mkpts0_c, const float* mkpts1_c,
                                          const int64_t* b_ids, const float* kenc, const float* rubbish,
                                          int64_t P, int64_t B, float* out0, float* out1,
                                          int64_t* p_s_out, int64_t* p_t_out, pats_stream_t stream) {
    PATS_REQUIRE(P >= 0 && B > 0, "third_descriptors: bad shape");
    if (P == 0) return PATS_OK;
    PATS_REQUIRE(feat_f0 && feat_f1 && mkpts0_c && mkpts1_c && b_ids && kenc && rubbish && out0 && out1,
                 "third_descriptors: null pointer");
    hipLaunchKernelGGL(third_desc_kernel, dim3((unsigned)P), dim3(256), 0, as_stream(stream), feat_f0, feat_f1,
                       mkpts0_c, mkpts1_c, b_ids, kenc, rubbish, P, B, out0, out1, p_s_out, p_t_out, (const int64_t*)nullptr);
    return check_launch("third_desc_kernel");
}

extern "C" int pats_third_descriptors_counted_f32(const float* feat_f0, const float* feat_f1,
                                                  const float* mkpts0_c, const float* mkpts1_c,
                                                  const int64_t* b_ids, const float* kenc, const float* rubbish,
                                                  int64_t P_cap, const int64_t* P_dev, int64_t B, float* out0, float* out1,
                                                  int64_t* p_s_out, int64_t* p_t_out, pats_stream_t stream) {
    PATS_REQUIRE(P_cap >= 0 && B > 0, "third_descriptors_counted: bad shape");
    if (P_cap == 0) return PATS_OK;
    PATS_REQUIRE(P_dev && feat_f0 && feat_f1 && mkpts0_c && mkpts1_c && b_ids && kenc && rubbish && out0 && out1,
                 "third_descriptors_counted: null pointer");
    hipLaunchKernelGGL(third_desc_kernel, dim3((unsigned)P_cap), dim3(256), 0, as_stream(stream), feat_f0, feat_f1,
                       mkpts0_c, mkpts1_c, b_ids, kenc, rubbish, P_cap, B, out0, out1, p_s_out, p_t_out, P_dev);
    return check_launch("third_desc_kernel");
}
