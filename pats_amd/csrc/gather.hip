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
// One 256-thread workgroup per stacked image n = s * B + b (left crops first).  Wave w takes channels w, w + 4, ...;
// a lane owns the points l, l + 64, l + 128 (< 145) of every channel it visits, so the source offsets of the three maps are
// computed ONCE per lane (no division in the channel loop) and a channel's 145 outputs leave as three coalesced stores.
// The channel loop is split by source (title / map 0 / map 1 / map 2): wave-uniform, branch-free bodies.
__global__ void __launch_bounds__(256)
fine_desc_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                 const float* __restrict__ f2, const float* __restrict__ title,
                 const float* __restrict__ rubbish, int64_t B, float* __restrict__ desc) {
    const int64_t n = blockIdx.x;              // s * B + b : index into the stacked maps
    const int64_t b = n % B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* o = desc + n * 264 * 145;
    int pt[3], off0[3], off1[3];
    bool live[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int p = lane + 64 * g;
        live[g] = p < 145;
        pt[g] = p < 144 ? p : 0;                               // positions (k // 12, k % 12); the dustbin column reads nothing
        const int r = pt[g] / 12, c = pt[g] - r * 12;
        off0[g] = (4 * r + 1) * 48 + 4 * c + 1;                // map 0: avgpool(2,1,1) -> 49x49, sample (4r+2, 4c+2)   :73-79
        off1[g] = (2 * r) * 24 + 2 * c;                        // map 1: avgpool -> 25x25, sample (2r+1, 2c+1)
    }
    const bool dust = lane == 16;                              // group 2 of lane 16 is point 144: the dustbin feature column
    auto put = [&](int ch, float v0, float v1, float v2) {
        float* row = o + ch * 145;
        row[lane] = v0;
        row[lane + 64] = v1;
        if (live[2]) row[lane + 128] = dust ? rubbish[b * 264 + ch] : v2;           // second_layer.py:83,85
    };
    for (int ch = wave; ch < 8; ch += 4) {                     // the 8-channel "title"                         :82,84
        const float v = title[b * 8 + ch];
        put(ch, v, v, v);
    }
#pragma unroll 2
    for (int ch = 8 + wave; ch < 72; ch += 4) {                // map 0: [.,64,48,48]
        const float* m = f0 + (n * 64 + (ch - 8)) * 48 * 48;
        float v[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float* q = m + off0[g];
            v[g] = (((q[0] + q[1]) + q[48]) + q[49]) / 4.0f;
        }
        put(ch, v[0], v[1], v[2]);
    }
#pragma unroll 2
    for (int ch = 72 + wave; ch < 136; ch += 4) {              // map 1: [.,64,24,24]
        const float* m = f1 + (n * 64 + (ch - 72)) * 24 * 24;
        float v[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float* q = m + off1[g];
            v[g] = (((q[0] + q[1]) + q[24]) + q[25]) / 4.0f;
        }
        put(ch, v[0], v[1], v[2]);
    }
#pragma unroll 4
    for (int ch = 136 + wave; ch < 264; ch += 4) {             // map 2: [.,128,12,12], no pooling, sample (r, c)
        const float* m = f2 + (n * 128 + (ch - 136)) * 144;
        put(ch, m[pt[0]], m[pt[1]], m[pt[2]]);
    }
}

// ---- third level -----------------------------------------------------------------------------
__device__ __forceinline__ long long round_half_even_div(float x, float d) {
    return (long long)rintf(x / d);      // torch.round = round half to even = rintf in the default mode
}

// one workgroup (256 threads = 4 waves) per point; wave w handles channels 32w .. 32w+31, eight at a time (sixteen
// window loads in flight, then sixteen stores); lane = window cell.  The dustbin feature column is written once per wave
// by 32 lanes (one channel each) instead of by lane 0 inside the channel loop.
__global__ void __launch_bounds__(256)
third_desc_kernel(const float* __restrict__ ff0, const float* __restrict__ ff1,
                  const float* __restrict__ mk0, const float* __restrict__ mk1,
                  const int64_t* __restrict__ b_ids, const float* __restrict__ kenc,
                  const float* __restrict__ rubbish, int64_t P, int64_t B,
                  float* __restrict__ out0, float* __restrict__ out1, int64_t* __restrict__ ps_out,
                  int64_t* __restrict__ pt_out, const int64_t* __restrict__ P_dev) {
    constexpr int W = 8, M = 52, C = 128;
    // throughput mode: the launch covers the capacity, the count is on the device.
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), and the points of one fine
    // row - consecutive p - read overlapping windows of the same two maps (neighbouring cells are 2 map pixels apart, a
    // window row is 32 bytes of a 128-byte line): workgroup k takes point (k % 8) * ceil(P / 8) + k / 8, so that
    // neighbours in p run on the SAME XCD one after the other and meet in its L2 instead of each fetching its own lines.
    int64_t live = P;
    if (P_dev) { const int64_t n = *P_dev; live = n < P ? n : P; }
    const int64_t per = (live + 7) >> 3, p = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per || p >= live) return;
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
    const float* src0 = ff0 + bb0 * C * (M * M) + r0;
    const float* src1 = ff1 + bb1 * C * (M * M) + r1;
#pragma unroll 1
    for (int c0 = 32 * wave; c0 < 32 * wave + 32; c0 += 8) {
        float a[8], c[8], ke[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[k] = src0[(int64_t)(c0 + k) * (M * M)];
            c[k] = src1[(int64_t)(c0 + k) * (M * M)];
            ke[k] = kenc[(c0 + k) * 64 + lane];                                  // + self.kenc(kpts)   :139-140
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o0[(c0 + k) * 65 + lane] = a[k] + ke[k];
            o1[(c0 + k) * 65 + lane] = c[k] + ke[k];
        }
    }
    if (lane < 32) {
        const int ch = 32 * wave + lane;
        const float rb = rubbish[(b * C + ch) * 144 + i2];
        o0[ch * 65 + 64] = rb;                                                   // :145-146
        o1[ch * 65 + 64] = rb;
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
    hipLaunchKernelGGL(third_desc_kernel, dim3((unsigned)((P + 7) / 8 * 8)), dim3(256), 0, as_stream(stream), feat_f0, feat_f1,
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
    hipLaunchKernelGGL(third_desc_kernel, dim3((unsigned)((P_cap + 7) / 8 * 8)), dim3(256), 0, as_stream(stream), feat_f0, feat_f1,
                       mkpts0_c, mkpts1_c, b_ids, kenc, rubbish, P_cap, B, out0, out1, p_s_out, p_t_out, P_dev);
    return check_launch("third_desc_kernel");
}
