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

#include <cstdlib>

namespace pats {

// Cache policy of the gathers' accesses (round 4).  tools/granule_probe.hip -> profiles/r04_granule_probe.txt: the memory side always
// moves whole 128-byte lines (32 bytes of every 128 cost what the whole line costs: the NCHW kernels' over-fetch is the layout's),
// and NON-TEMPORAL loads of data used once stream 9 % faster than plain ones (6.28 against 5.74 TB/s dense).  In the kernels
// (tools/gather_nt_ab.sh -> profiles/r04_gather_nt_ab.txt, inside the bench's steps): the CHANNELS-LAST kernels, whose outputs leave
// as whole lines of a linear 16-byte stream, gain 6 % / 9 % from non-temporal STORES (4.10 -> 3.85, 2.94 -> 2.67 ms; non-temporal
// loads alone change nothing there); the NCHW kernels LOSE with either - their taps re-visit lines through the L1 the policy
// bypasses (6.2 -> 6.7, 4.9 -> 6.8 ms with such loads) and their 4-byte stores into rows of 145 / 65 floats end lines partially
// (-> 7.3, 8.0 ms with such stores).  POL bit 0: map reads non-temporal; bit 1: output stores non-temporal.  Defaults: 0 on NCHW
// maps, 3 on channels-last maps; PATS_GATHER_NT = 0..3 overrides both (read once per process).
template <int POL, typename T>
__device__ __forceinline__ T ldm(const T* p) {
    if (POL & 1) return __builtin_nontemporal_load(p);
    return *p;
}
template <int POL, typename T>
__device__ __forceinline__ void stm(T* p, T v) {
    if (POL & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// two neighbouring floats as ONE 8-byte access at 4-byte alignment (gfx950 global memory takes unaligned dwordx2; the pooled taps of
// map 0 start at an odd element): half the load instructions of the pooling loops
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
template <int POL>
__device__ __forceinline__ f2u ldm2(const float* p) {
    const f2u* q = reinterpret_cast<const f2u*>(p);
    if (POL & 1) return __builtin_nontemporal_load(q);
    return *q;
}

// ---- fine level ------------------------------------------------------------------------------
// One 256-thread workgroup per stacked image n = s * B + b (left crops first).  Wave w takes channels w, w + 4, ...;
// a lane owns the points l, l + 64, l + 128 (< 145) of every channel it visits, so the source offsets of the three maps are
// computed ONCE per lane (no division in the channel loop) and a channel's 145 outputs leave as three coalesced stores.
// The channel loop is split by source (title / map 0 / map 1 / map 2): wave-uniform, branch-free bodies.
template <int POL>
__global__ void __launch_bounds__(256)
fine_desc_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                 const float* __restrict__ f2, const float* __restrict__ title,
                 const float* __restrict__ rubbish, int64_t B, float* __restrict__ desc,
                 const int64_t* __restrict__ B_live) {
    const int64_t n = blockIdx.x;              // s * B + b : index into the stacked maps
    const int64_t b = n % B;
    if (B_live && b >= *B_live) return;        // counted launch: rows past the device-side total are padding
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
        stm<POL>(row + lane, v0);
        stm<POL>(row + lane + 64, v1);
        if (live[2]) stm<POL>(row + lane + 128, dust ? rubbish[b * 264 + ch] : v2);           // second_layer.py:83,85
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
            const f2u a = ldm2<POL>(q), c2 = ldm2<POL>(q + 48);
            v[g] = (((a.x + a.y) + c2.x) + c2.y) / 4.0f;
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
            const f2u a = ldm2<POL>(q), c2 = ldm2<POL>(q + 24);
            v[g] = (((a.x + a.y) + c2.x) + c2.y) / 4.0f;
        }
        put(ch, v[0], v[1], v[2]);
    }
#pragma unroll 4
    for (int ch = 136 + wave; ch < 264; ch += 4) {             // map 2: [.,128,12,12], no pooling, sample (r, c)
        const float* m = f2 + (n * 128 + (ch - 136)) * 144;
        put(ch, ldm<POL>(m + pt[0]), ldm<POL>(m + pt[1]), ldm<POL>(m + pt[2]));
    }
}

// ---- third level -----------------------------------------------------------------------------
__device__ __forceinline__ long long round_half_even_div(float x, float d) {
    return (long long)rintf(x / d);      // torch.round = round half to even = rintf in the default mode
}

// one workgroup (256 threads = 4 waves) per point; wave w handles channels 32w .. 32w+31, eight at a time (sixteen
// window loads in flight, then sixteen stores); lane = window cell.  The dustbin feature column is written once per wave
// by 32 lanes (one channel each) instead of by lane 0 inside the channel loop.
template <int POL>
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
    // as a row of the flattened [B*144, 128] view, like the reference: cell 11 of a patch (round(92 / 8) = 12) reads the NEXT
    // patch's feature (PATS itself never sends the border ring here: second_layer.py:140-149,176-184)
    long long x2 = round_half_even_div((float)s0, 8.0f), y2 = round_half_even_div((float)s1, 8.0f);
    long long i2 = b * 144 + y2 * 12 + x2;
    i2 = i2 < 0 ? 0 : (i2 > B * 144 - 1 ? B * 144 - 1 : i2);      // memory safety (torch.gather would raise out of range)
    const long long bb2 = i2 / 144;
    i2 -= bb2 * 144;
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
            a[k] = ldm<POL>(src0 + (int64_t)(c0 + k) * (M * M));
            c[k] = ldm<POL>(src1 + (int64_t)(c0 + k) * (M * M));
            ke[k] = kenc[(c0 + k) * 64 + lane];                                  // + self.kenc(kpts)   :139-140
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            stm<POL>(o0 + (c0 + k) * 65 + lane, a[k] + ke[k]);
            stm<POL>(o1 + (c0 + k) * 65 + lane, c[k] + ke[k]);
        }
    }
    if (lane < 32) {
        const int ch = 32 * wave + lane;
        const float rb = rubbish[(bb2 * C + ch) * 144 + i2];
        o0[ch * 65 + 64] = rb;                                                   // :145-146
        o1[ch * 65 + 64] = rb;
    }
}


// ---- the same two gathers on CHANNELS-LAST maps ----------------------------------------------------------------------
// The reference gathers from `feat.permute(0, 2, 3, 1).reshape(-1, C)` (third_layer.py:139-140) and samples single pixels
// of AvgPool'd maps (second_layer.py:73-79): per-PIXEL reads of all channels.  On the NCHW tensors a torch conv emits by
// default a pixel's channels lie H*W*4 bytes apart - a third-level window row is 32 bytes of every 208-byte map row (2.75
// 64-byte HBM granules fetched per 32 bytes used, measured), the stride-4 samples of the 48x48 map touch half of its
// granules for a sixteenth of its pixels.  A backbone run in torch.channels_last (MIOpen's native layout; the logical
// shape stays [B,C,H,W]) puts a pixel's channels in ONE contiguous run of 256 / 512 bytes: every granule fetched is
// used in full.  Lanes then run over channels while the outputs want lanes over nodes ([C, nodes] rows for the MFMA cost
// builds), so the tile turns through LDS and leaves as one linear, 16-byte-vectorised copy.  Same values, same
// operation order per element as the NCHW kernels: bit-identical outputs (tests/test_gpu_parity.py).

// one workgroup per (point, side): 64 pixels x 128 channels = 32 KB in, [128, 65] out
template <int POL>
__global__ void __launch_bounds__(256)
third_desc_nhwc_kernel(const float* __restrict__ ff0, const float* __restrict__ ff1,
                       const float* __restrict__ mk0, const float* __restrict__ mk1,
                       const int64_t* __restrict__ b_ids, const float* __restrict__ kenc,
                       const float* __restrict__ rubbish, int64_t P, int64_t B,
                       float* __restrict__ out0, float* __restrict__ out1, int64_t* __restrict__ ps_out,
                       int64_t* __restrict__ pt_out, const int64_t* __restrict__ P_dev) {
    constexpr int W = 8, M = 52, C = 128, NT = 65;
    __shared__ __attribute__((aligned(16))) float tile[C * NT];
    int64_t live = P;
    if (P_dev) { const int64_t n = *P_dev; live = n < P ? n : P; }
    // XCD-aware order as in third_desc_kernel; the two sides of a point follow each other on the same XCD (they share
    // the dustbin feature's lines)
    const unsigned k = blockIdx.x >> 3;
    const int side = k & 1;
    const int64_t per = (live + 7) >> 3, p = (int64_t)(blockIdx.x & 7) * per + (k >> 1);
    if ((int64_t)(k >> 1) >= per || p >= live) return;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t b = b_ids[p];
    const long long s0 = round_half_even_div(mk0[p * 2 + 0], 4.0f) * 4, s1 = round_half_even_div(mk0[p * 2 + 1], 4.0f) * 4;   // :124
    float t0 = mk1[p * 2 + 0], t1 = mk1[p * 2 + 1];                                                                          // :128-130
    t0 = t0 >= 96.f ? 96.f : t0; t1 = t1 >= 96.f ? 96.f : t1;
    t0 = t0 <= 0.f ? 0.f : t0;   t1 = t1 <= 0.f ? 0.f : t1;
    const long long q0 = round_half_even_div(t0, 4.0f) * 4, q1 = round_half_even_div(t1, 4.0f) * 4;
    if (t == 0 && side == 0) {
        if (ps_out) { ps_out[p * 2] = s0; ps_out[p * 2 + 1] = s1; }
        if (pt_out) { pt_out[p * 2] = q0; pt_out[p * 2 + 1] = q1; }
    }
    auto fdiv2 = [](long long v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };      // python floor division
    // row of the NHWC view of window cell (0, 0): b M M + (y // 2 - W/2 + 2) M + (x // 2 - W/2 + 2)      :125-127,131-133
    const long long i00 = b * M * M + (fdiv2(side ? q1 : s1) - W / 2 + 2) * M + (fdiv2(side ? q0 : s0) - W / 2 + 2);
    const long long lim = B * M * M - 1;
    const float* __restrict__ ff = side ? ff1 : ff0;
    // dustbin feature: rubbish[b, :, y2*12 + x2]                                                           :141-144
    long long i2 = b * 144 + round_half_even_div((float)s1, 8.0f) * 12 + round_half_even_div((float)s0, 8.0f);
    i2 = i2 < 0 ? 0 : (i2 > B * 144 - 1 ? B * 144 - 1 : i2);      // a row of the flattened [B*144, 128] view, see third_desc_kernel
    const long long bb2 = i2 / 144;
    i2 -= bb2 * 144;
    float rb = 0.f;
    if (t < C) rb = rubbish[(bb2 * C + t) * 144 + i2];
    // wave w brings window cells 16w .. 16w+15, lane l channels l and l + 64 of each: 32 loads of 256 contiguous bytes in flight
    float v0[16], v1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        long long i = i00 + (2 * wave + (j >> 3)) * M + (j & 7);
        i = i < 0 ? 0 : (i > lim ? lim : i);          // memory safety (torch.gather would raise out of range)
        const float* src = ff + i * C;
        v0[j] = ldm<POL>(src + lane);
        v1[j] = ldm<POL>(src + lane + 64);
    }
    // + self.kenc(kpts) rides on the way out: element e = c * 65 + n of the output takes kenc[c, n]       :139-140
    float ke[33];
#pragma unroll
    for (int r = 0; r < 33; ++r) {
        const int e = t + 256 * r, c = e / NT, n = e - c * NT;
        ke[r] = kenc[(e < C * NT && n < 64) ? c * 64 + n : 0];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        tile[lane * NT + 16 * wave + j] = v0[j];
        tile[(lane + 64) * NT + 16 * wave + j] = v1[j];
    }
    if (t < C) tile[t * NT + 64] = rb;                                                                      // :145-146
    wg_barrier();
    float* o = (side ? out1 : out0) + p * C * NT;
#pragma unroll
    for (int r = 0; r < 33; ++r) {
        const int e = t + 256 * r, c = e / NT, n = e - c * NT;
        if (e < C * NT) stm<POL>(o + e, n < 64 ? tile[e] + ke[r] : tile[e]);
    }
}

// one workgroup per stacked image; the 264 output channels leave in four 64-channel tiles (map 0, map 1, the two halves of
// map 2), each gathered with 16-byte loads - lane = (node % 4, four channels) - pooled in registers, turned in LDS and
// copied out as float4.  `cpp` = channels per pixel of the map, `ch0` = first of the 64 channels this pass takes.
template <int TAPS, int POL>
__device__ __forceinline__ void fine_tile_pass(const float* __restrict__ img, int cpp, int ch0, int rowpix, int step, int first,
                                               float* tile, float* __restrict__ o, const float* __restrict__ rub, int t) {
    constexpr int NP = 145;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = t & 63, wave = t >> 6, cg = lane & 15, sub = lane >> 4;
    constexpr int GR = TAPS == 1 ? 9 : 3;          // node groups per round: 9 or 12 loads of 16 bytes in flight per lane
    float dust = 0.f;
    if (t < 64) dust = rub[t];                                                                   // second_layer.py:83,85
#pragma unroll 1
    for (int g0 = 0; g0 < 9; g0 += GR) {
        f4 q[GR][TAPS];
#pragma unroll
        for (int g = 0; g < GR; ++g) {
            const int nd = 16 * (g0 + g) + 4 * wave + sub, r = nd / 12, c = nd - 12 * r;       // positions (k // 12, k % 12)
            const float* px = img + ((step * r + first) * rowpix + step * c + first) * cpp + ch0 + 4 * cg;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
                q[g][tap] = ldm<POL>(reinterpret_cast<const f4*>(px + ((tap >> 1) * rowpix + (tap & 1)) * cpp));
        }
#pragma unroll
        for (int g = 0; g < GR; ++g) {
            const int nd = 16 * (g0 + g) + 4 * wave + sub;
            f4 v = q[g][0];
            if (TAPS == 4) v = (((q[g][0] + q[g][1]) + q[g][2]) + q[g][3]) / 4.0f;              // AvgPool2d(2, 1, 1)  :73-79
            tile[(4 * cg + 0) * NP + nd] = v.x;
            tile[(4 * cg + 1) * NP + nd] = v.y;
            tile[(4 * cg + 2) * NP + nd] = v.z;
            tile[(4 * cg + 3) * NP + nd] = v.w;
        }
    }
    if (t < 64) tile[t * NP + 144] = dust;
    wg_barrier();
    const f4* src = reinterpret_cast<const f4*>(tile);
    f4* dst = reinterpret_cast<f4*>(o);
    for (int e = t; e < 64 * NP / 4; e += 256) stm<POL>(dst + e, src[e]);
    wg_barrier();
}

template <int POL>
__global__ void __launch_bounds__(256)
fine_desc_nhwc_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                      const float* __restrict__ f2, const float* __restrict__ title,
                      const float* __restrict__ rubbish, int64_t B, float* __restrict__ desc,
                      const int64_t* __restrict__ B_live) {
    constexpr int NP = 145;
    __shared__ __attribute__((aligned(16))) float tile[64 * NP];
    const int64_t n = blockIdx.x;              // s * B + b : index into the stacked maps
    const int64_t b = n % B;
    if (B_live && b >= *B_live) return;        // counted launch: rows past the device-side total are padding
    const int t = threadIdx.x;
    float* o = desc + n * 264 * NP;
    const float* rub = rubbish + b * 264;
    for (int e = t; e < 8 * NP; e += 256) {                    // the 8-channel "title"                         :82,84
        const int ch = e / NP, p = e - ch * NP;
        o[e] = p == 144 ? rub[ch] : title[b * 8 + ch];
    }
    // map 0 [.,48,48,64]: avgpool(2,1,1) -> 49x49, sample (4r+2, 4c+2) = mean of pixels (4r+1.., 4c+1..)
    fine_tile_pass<4, POL>(f0 + n * 48 * 48 * 64, 64, 0, 48, 4, 1, tile, o + 8 * NP, rub + 8, t);
    // map 1 [.,24,24,64]: avgpool -> 25x25, sample (2r+1, 2c+1) = mean of pixels (2r.., 2c..)
    fine_tile_pass<4, POL>(f1 + n * 24 * 24 * 64, 64, 0, 24, 2, 0, tile, o + 72 * NP, rub + 72, t);
    // map 2 [.,12,12,128]: no pooling, sample (r, c)
    fine_tile_pass<1, POL>(f2 + n * 144 * 128, 128, 0, 12, 1, 0, tile, o + 136 * NP, rub + 136, t);
    fine_tile_pass<1, POL>(f2 + n * 144 * 128, 128, 64, 12, 1, 0, tile, o + 200 * NP, rub + 200, t);
}

// PATS_GATHER_NT = 0..3 (see ldm / stm above), read once per process; without it 0 on NCHW maps, 3 on channels-last maps
static int gather_policy(bool channels_last) {
    static const int pol = [] { const char* e = env_switch("PATS_GATHER_NT"); return e ? (atoi(e) & 3) : -1; }();
    return pol >= 0 ? pol : (channels_last ? 3 : 0);
}
#define GATHER_LAUNCH(KERNEL, NHWC, grid, ...)                                                                                \
    switch (gather_policy(NHWC)) {                                                                                           \
        case 0: hipLaunchKernelGGL((KERNEL<0>), grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break;                  \
        case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break;                  \
        case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break;                  \
        default: hipLaunchKernelGGL((KERNEL<3>), grid, dim3(256), 0, as_stream(stream), __VA_ARGS__); break;                 \
    }

}  // namespace pats

using namespace pats;

extern "C" int pats_fine_descriptors_f32(const float* feat0, const float* feat1, const float* feat2,
                                         const float* title, const float* rubbish, int64_t B,
                                         float* desc, pats_stream_t stream) {
    PATS_REQUIRE(B >= 0, "fine_descriptors: bad shape");
    if (B == 0) return PATS_OK;
    PATS_REQUIRE(feat0 && feat1 && feat2 && title && rubbish && desc, "fine_descriptors: null pointer");
    GATHER_LAUNCH(fine_desc_kernel, false, dim3((unsigned)(2 * B)), feat0, feat1,
                       feat2, title, rubbish, B, desc, (const int64_t*)nullptr);
    return check_launch("fine_desc_kernel");
}

// a15 launched over a CAPACITY of B_cap rows with the number of rows in use on the device (throughput mode: the fine level's
// row table): workgroups of rows >= *B_dev return at once, their desc blocks are left untouched.
extern "C" int pats_fine_descriptors_counted_f32(const float* feat0, const float* feat1, const float* feat2,
                                                 const float* title, const float* rubbish, int64_t B_cap,
                                                 const int64_t* B_dev, int channels_last, float* desc, pats_stream_t stream) {
    PATS_REQUIRE(B_cap >= 0, "fine_descriptors_counted: bad shape");
    if (B_cap == 0) return PATS_OK;
    PATS_REQUIRE(B_dev && feat0 && feat1 && feat2 && title && rubbish && desc, "fine_descriptors_counted: null pointer");
    if (channels_last) {
        PATS_REQUIRE(((uintptr_t)feat0 | (uintptr_t)feat1 | (uintptr_t)feat2 | (uintptr_t)desc) % 16 == 0,
                     "fine_descriptors_counted: channels-last maps and desc must be 16-byte aligned");
        GATHER_LAUNCH(fine_desc_nhwc_kernel, true, dim3((unsigned)(2 * B_cap)), feat0, feat1,
                           feat2, title, rubbish, B_cap, desc, B_dev);
        return check_launch("fine_desc_nhwc_kernel");
    }
    GATHER_LAUNCH(fine_desc_kernel, false, dim3((unsigned)(2 * B_cap)), feat0, feat1,
                       feat2, title, rubbish, B_cap, desc, B_dev);
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
    GATHER_LAUNCH(third_desc_kernel, false, dim3((unsigned)((P + 7) / 8 * 8)), feat_f0, feat_f1,
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
    GATHER_LAUNCH(third_desc_kernel, false, dim3((unsigned)((P_cap + 7) / 8 * 8)), feat_f0, feat_f1,
                       mkpts0_c, mkpts1_c, b_ids, kenc, rubbish, P_cap, B, out0, out1, p_s_out, p_t_out, P_dev);
    return check_launch("third_desc_kernel");
}

extern "C" int pats_fine_descriptors_nhwc_f32(const float* feat0, const float* feat1, const float* feat2,
                                              const float* title, const float* rubbish, int64_t B,
                                              float* desc, pats_stream_t stream) {
    PATS_REQUIRE(B >= 0, "fine_descriptors_nhwc: bad shape");
    if (B == 0) return PATS_OK;
    PATS_REQUIRE(feat0 && feat1 && feat2 && title && rubbish && desc, "fine_descriptors_nhwc: null pointer");
    PATS_REQUIRE(((uintptr_t)feat0 | (uintptr_t)feat1 | (uintptr_t)feat2 | (uintptr_t)desc) % 16 == 0,
                 "fine_descriptors_nhwc: maps and desc must be 16-byte aligned");
    GATHER_LAUNCH(fine_desc_nhwc_kernel, true, dim3((unsigned)(2 * B)), feat0, feat1,
                       feat2, title, rubbish, B, desc, (const int64_t*)nullptr);
    return check_launch("fine_desc_nhwc_kernel");
}

extern "C" int pats_third_descriptors_nhwc_f32(const float* feat_f0, const float* feat_f1,
                                               const float* mkpts0_c, const float* mkpts1_c,
                                               const int64_t* b_ids, const float* kenc, const float* rubbish,
                                               int64_t P_cap, const int64_t* P_dev, int64_t B, float* out0, float* out1,
                                               int64_t* p_s_out, int64_t* p_t_out, pats_stream_t stream) {
    PATS_REQUIRE(P_cap >= 0 && B > 0, "third_descriptors_nhwc: bad shape");
    if (P_cap == 0) return PATS_OK;
    PATS_REQUIRE(feat_f0 && feat_f1 && mkpts0_c && mkpts1_c && b_ids && kenc && rubbish && out0 && out1,
                 "third_descriptors_nhwc: null pointer");
    GATHER_LAUNCH(third_desc_nhwc_kernel, true, dim3((unsigned)((P_cap + 7) / 8 * 16)), feat_f0,
                       feat_f1, mkpts0_c, mkpts1_c, b_ids, kenc, rubbish, P_cap, B, out0, out1, p_s_out, p_t_out, P_dev);
    return check_launch("third_desc_nhwc_kernel");
}
