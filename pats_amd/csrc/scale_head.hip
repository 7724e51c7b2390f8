// The scale head: what turns the target descriptors into the OT problem's column marginals `ns`.
//
//   first_layer.py:39-40,106-107    scale = exp(sigmoid(scalex_proj(mdesc1 as [b,448,15,20])) * ln256 - ln256 / 2)
//   second_layer.py:33-36,92-98     scale_x, scale_y the same from two heads on [B,264,12,12]; scale = scale_x * scale_y
//   third_layer.py:88-89,151-152    scale the same from scale_proj on [P,128,8,8]
// with *_proj = nn.Conv2d(C, 1, kernel_size=3, padding=1): a 3 x 3 stencil over the descriptor grid, all C channels
// into ONE output channel per head.  The descriptors are the [b, C, ld] tensors the cost build takes (ld = h*w, or
// h*w + 1 with the dustbin feature column, which the heads do not see: `mdesc1[:, :, :-1]`).
//
// One workgroup per problem, and the stencil turned inside out: out[p] = sum_k sum_c w[c][k] x[c][p + off_k] is computed
// as nine TAP PLANES T_k[q] = sum_c w[c][k] x[c][q] - one coalesced global load of x[c][q] per channel and cell, nine
// FMAs against wave-uniform (scalar-loaded) weights, no LDS in the loop - and one nine-way gather of the planes
// through LDS at the end (out-of-grid taps read a zero slot).  The four waves of the workgroup split the CHANNELS (c mod 4),
// each covering the whole grid with 1 / 3 / 5 / 8 cells per lane, and add their planes in LDS in a fixed order.
// HBM-bound on the descriptor read (33 KB per third-level problem for 74 k multiply-adds); no MFMA - nine output
// rows would use 9 of a tile's 32.
#include "common.hpp"

namespace pats {

namespace {

constexpr int SH_THREADS = 256;
constexpr float LN256 = 5.545177444479562f, HALF_LN256 = 2.772588722239781f;      // math.log(256.0), / 2, as fp32

template <int HEADS, int CELLS>
__global__ void __launch_bounds__(SH_THREADS)
scale_head_kernel(const float* __restrict__ x, int C, int ld, int h, int w, const float* __restrict__ weight,
                  const float* __restrict__ bias, float* __restrict__ out, float* __restrict__ per_head) {
    extern __shared__ float sm[];                    // [HEADS * 9][hw + 1]; slot hw of a plane = the zero the border taps read
    const int hw = h * w, row = hw + 1;
    const int64_t b = blockIdx.x;
    const float* xb = x + b * (int64_t)C * ld;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // wave wv takes the channels c = wv (mod 4); its lanes cover the whole grid, CELLS cells each (lane, lane + 64, ...)
    float T[CELLS][HEADS * 9];
    bool in[CELLS];
    const float* xq[CELLS];
#pragma unroll
    for (int q = 0; q < CELLS; ++q) {
        in[q] = lane + 64 * q < hw;
        xq[q] = xb + (in[q] ? lane + 64 * q : 0);
#pragma unroll
        for (int k = 0; k < HEADS * 9; ++k) T[q][k] = 0.f;
    }
    for (int c = wave; c < C; c += SH_THREADS / 64) {
        float v[CELLS];
#pragma unroll
        for (int q = 0; q < CELLS; ++q) v[q] = in[q] ? xq[q][(int64_t)c * ld] : 0.f;
        const float* wc = weight + (int64_t)c * 9;                // wave-uniform: scalar loads
#pragma unroll
        for (int hd = 0; hd < HEADS; ++hd)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float wk = wc[(int64_t)hd * C * 9 + k];
#pragma unroll
                for (int q = 0; q < CELLS; ++q) T[q][hd * 9 + k] = fmaf(wk, v[q], T[q][hd * 9 + k]);
            }
    }
    // the four waves' planes are added in LDS one wave after the other (a fixed order)
    for (int wv = 0; wv < SH_THREADS / 64; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int q = 0; q < CELLS; ++q)
                if (in[q])
#pragma unroll
                    for (int k = 0; k < HEADS * 9; ++k) {
                        float* slot = &sm[k * row + lane + 64 * q];
                        *slot = wv == 0 ? T[q][k] : *slot + T[q][k];
                    }
        }
        if (wv == 0 && t < HEADS * 9) sm[t * row + hw] = 0.f;
        wg_barrier();
    }
    for (int p = t; p < hw; p += SH_THREADS) {
        const int py = p / w, px = p - py * w;
        float s = 1.0f;
#pragma unroll
        for (int hd = 0; hd < HEADS; ++hd) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int yy = py + k / 3 - 1, xx = px + k % 3 - 1;
                const int slot = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? yy * w + xx : hw;
                v += sm[(hd * 9 + k) * row + slot];
            }
            v += bias[hd];
            const float sig = 1.0f / (1.0f + expf(-v));                  // nn.Sigmoid
            const float e = expf(sig * LN256 - HALF_LN256);
            s = hd == 0 ? e : s * e;                                      // scale_x * scale_y (second_layer.py:98)
            if (per_head) per_head[(b * HEADS + hd) * hw + p] = e;        // scale_x, scale_y on their own (est_position takes them)
        }
        out[b * hw + p] = s;
    }
}

}  // namespace

}  // namespace pats

using namespace pats;

extern "C" int pats_scale_head_f32(const float* x, int64_t batch, int C, int ld, int h, int w, const float* weight,
                                   const float* bias, int heads, float* out, float* per_head, pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && C > 0 && h > 0 && w > 0 && ld >= h * w, "scale_head: bad shape");
    PATS_REQUIRE(h * w <= 2 * SH_THREADS, "scale_head: grid of %d cells exceeds %d", h * w, 2 * SH_THREADS);
    PATS_REQUIRE(heads == 1 || heads == 2, "scale_head: one head (first / third layer) or two (second layer: x, y)");
    PATS_REQUIRE(batch < (1ll << 31), "scale_head: batch too large");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(x && weight && bias && out, "scale_head: null pointer");
    const int hw = h * w;
    const size_t lds = (size_t)heads * 9 * (hw + 1) * sizeof(float);        // at most 2 heads x 9 planes x 513 floats = 36.9 KB
    const dim3 grid((unsigned)batch), block(SH_THREADS);
    hipStream_t st = as_stream(stream);
#define PATS_SH_LAUNCH(HD, CL) hipLaunchKernelGGL((scale_head_kernel<HD, CL>), grid, block, lds, st, x, C, ld, h, w, weight, bias, out, per_head)
    if (heads == 1) {
        if (hw <= 64) PATS_SH_LAUNCH(1, 1); else if (hw <= 192) PATS_SH_LAUNCH(1, 3); else if (hw <= 320) PATS_SH_LAUNCH(1, 5); else PATS_SH_LAUNCH(1, 8);
    } else {
        if (hw <= 64) PATS_SH_LAUNCH(2, 1); else if (hw <= 192) PATS_SH_LAUNCH(2, 3); else if (hw <= 320) PATS_SH_LAUNCH(2, 5); else PATS_SH_LAUNCH(2, 8);
    }
#undef PATS_SH_LAUNCH
    return check_launch("scale_head_kernel");
}
