// AttentionalPropagation of the SuperGlue-style GNN layers on gfx950.
//   models/modules.py:91-117:
//     MultiHeadedAttention.forward:  q, k, v = proj[i](x) (Conv1d k=1) viewed [b, dim, heads, n];
//                                    x, _ = attention(q, k, v);  merge(x.view(b, dim*heads, n))          (:100-105)
//     AttentionalPropagation.forward: message = attn(x, source, source)
//                                     mlp(cat([x, message], dim=1))   with mlp = Conv1d(2C,2C) BN ReLU Conv1d(2C,C)   (:107-117, MLP :57-69)
//     AttentionalGNN.forward:         desc = desc + delta                                                    (:131-133)
//
// A Conv1d with kernel 1 over [b, C_in, n] is the product W [C_out x C_in] . X_b [C_in x n] for every b, i.e. the
// channel-major descriptor layout of the cost build with the weights as the shared (batch-stride 0) operand: the
// same 160 x 160 MFMA tile (mfma_tile.hpp: fp32 operands as fp16 hi + lo pairs, three exact-product passes, fp32
// accumulation, in-kernel fp32 redo for operands beyond the fp16 range; PATS_COST_F32=1 selects the fp32 MFMA
// throughout), generalised in three ways so that the layer needs no glue kernels:
//   * the 160 tile columns run over the flattened (batch, token) axis - a tile spans several problems, so 65-token
//     problems fill the tile instead of wasting 60 % of it;
//   * the reduction dimension may come from TWO tensors (x | message): `cat` is never materialised;
//   * input channels can carry an affine + ReLU applied while staging (BatchNorm in eval mode = its running
//     statistics, in train mode = batch statistics from the bn_* kernels below: PATS.eval leaves the third layer in train
//     mode, pats.py:112-120), and the epilogue adds the bias and an optional residual (desc + delta).
// The attention core in the middle is pats_attention_f32 (attention.hip).
#include "mfma_tile.hpp"

#include <cstdlib>

namespace pats {

namespace {

struct ConvArgs {
    const float* wt;           // [K0 + K1][M]: transposed weights (row = input channel)
    const float* x0;           // [batch, K0, n]
    const float* x1;           // [batch, K1, n] or null (K1 = 0)
    int K0, K1, M, n;
    int64_t cols;              // batch * n
    const float* in_scale;     // [K0 + K1] or null: x <- max(0, x * scale + shift) while staging
    const float* in_shift;
    const float* bias;         // [M] or null
    const float* residual;     // [batch, M, n] or null
    float* y;                  // [batch, M, n]
    int tile_rows;             // output rows per workgroup: 160, or 128 when M is a multiple of 128 (no fifth tile row)
};

// column addressing for mfma_tile.hpp's CmSrc: A = transposed weights at output rows i0.. (clamped), B = activations at
// flattened (batch, token) columns j0..: tile column c -> element offset of channel 0 in a [batch, Kc, n] tensor.  The
// decomposition is 32-bit (launch_conv requires batch * n < 2^31) and done once per item, not per chunk.
struct ConvCols {
    int i0, M, n, Kc;          // Kc: channels of THIS activation tensor (its batch stride is Kc * n)
    int64_t j0, cols;
    __device__ __forceinline__ int64_t a_off(int c) const { return min(i0 + c, M - 1); }
    __device__ __forceinline__ int64_t b_off(int c) const {
        const unsigned cg = (unsigned)min(j0 + c, cols - 1), b = cg / (unsigned)n, tk = cg - b * (unsigned)n;
        return (int64_t)b * Kc * n + tk;
    }
    __device__ __forceinline__ bool row_stored(int r) const { return i0 + r < M; }
    __device__ __forceinline__ bool col_stored(int c) const { return j0 + c < cols; }
};

}  // namespace

template <bool SPLIT, bool ROW4>
__global__ void __launch_bounds__(256, 2)
conv1x1_kernel(ConvArgs g) {
    __shared__ mt::Lds lds;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int64_t tiles_j = (g.cols + mt::CT - 1) / mt::CT;
    const int i0 = (int)(blockIdx.x / tiles_j) * g.tile_rows;
    const int64_t j0 = (int64_t)(blockIdx.x % tiles_j) * mt::CT;
    const int n = g.n, M = g.M;
    // output-row tiles that exist: rows i0 + 32 w .. for wave w, and the fifth tile row (i0 + 128 ..) shared by all
    // (workgroup-uniform: 128-channel layers skip it)
    const bool row4 = ROW4 && i0 + 128 < M;

    // the reduction runs over x0's channels, then over x1's (weights rows K0..): two sources, one accumulator
    const ConvCols c0{i0, M, n, g.K0, j0, g.cols}, c1{i0, M, n, g.K1, j0, g.cols};
    mt::CmSrc<ConvCols> s0(g.wt, M, g.x0, n, g.K0, c0, t, g.in_scale, g.in_shift);
    mt::CmSrc<ConvCols> s1(g.wt + (int64_t)g.K0 * M, M, g.x1, n, g.K1, c1, t, g.in_scale ? g.in_scale + g.K0 : nullptr,
                           g.in_shift ? g.in_shift + g.K0 : nullptr);
    mt::f32x16 acc[7];
    const float unscale = mt::tile<SPLIT, ROW4>(s0, g.K1 > 0 ? &s1 : nullptr, lds, acc, row4, t, wave);

    // C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    auto store_tile = [&](const mt::f32x16& cacc, int ti, int tj) {
        if (j0 + 32 * tj + li >= g.cols) return;
        const unsigned cg = (unsigned)(j0 + 32 * tj + li), b = cg / (unsigned)n, tk = cg - b * (unsigned)n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < M) {
                const int64_t o = ((int64_t)b * M + row) * n + tk;
                float v = cacc[r] * unscale;
                if (g.bias) v += g.bias[row];
                if (g.residual) v = g.residual[o] + v;
                g.y[o] = v;
            }
        }
    };
#pragma unroll
    for (int tj = 0; tj < 5; ++tj) store_tile(acc[tj], wave, tj);
    if (row4) {
        store_tile(acc[5], 4, wave);
        if (wave == 0) store_tile(acc[6], 4, 4);
    }
}

// BatchNorm1d in train mode (modules.py:66 inside MLP; the third layer's GNN runs it on batch statistics because
// PATS.eval() does not reach it, pats.py:112-120): per channel over (batch, n), biased variance, then
// scale = gamma / sqrt(var + eps), shift = beta - mean * scale.  Two passes, both in double and both in a fixed order
// (the result does not depend on scheduling): bn_partial_kernel - grid (channel, BN_SPLITS), wave w of split s sums the
// batches s * 4 + w, + 4 * BN_SPLITS, ... (a batch's row of one channel is n contiguous floats: one coalesced read per
// wave) - then bn_finish_kernel adds the BN_SPLITS partials of a channel.  The first version walked the flattened
// (batch, token) index with a 64-bit division per element from ONE workgroup per channel: 3.5 ms for the 1.7 GB of the
// third level's hidden tensor, a third of the layer.
constexpr int BN_SPLITS = 64;

__global__ void __launch_bounds__(256)
bn_partial_kernel(const float* __restrict__ h, int64_t batch, int C, int n, double* __restrict__ part) {
    __shared__ double s1[4], s2[4];
    const int c = blockIdx.x, split = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 0.0, q = 0.0;
    for (int64_t b = (int64_t)split * 4 + wave; b < batch; b += 4 * BN_SPLITS) {
        const float* row = h + (b * C + c) * (int64_t)n;
        for (int t = lane; t < n; t += 64) {
            const double x = (double)row[t];
            a += x;
            q += x * x;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {          // fixed butterfly order
        a += __shfl_xor(a, o);
        q += __shfl_xor(q, o);
    }
    if (lane == 0) { s1[wave] = a; s2[wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((int64_t)c * BN_SPLITS + split) * 2 + 0] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
        part[((int64_t)c * BN_SPLITS + split) * 2 + 1] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    }
}

__global__ void __launch_bounds__(64)
bn_finish_kernel(const double* __restrict__ part, int64_t total, int C, const float* __restrict__ gamma,
                 const float* __restrict__ beta, float eps, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, q = 0.0;
    for (int s_ = 0; s_ < BN_SPLITS; ++s_) {
        a += part[((int64_t)c * BN_SPLITS + s_) * 2 + 0];
        q += part[((int64_t)c * BN_SPLITS + s_) * 2 + 1];
    }
    const double mean = a / (double)total;
    double var = q / (double)total - mean * mean;
    if (var < 0.0) var = 0.0;
    const float sc = gamma[c] / sqrtf((float)var + eps);
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
}

static int launch_conv(const ConvArgs& g0, hipStream_t st) {
    ConvArgs g = g0;
    static const bool fp32_only = [] { const char* e = getenv("PATS_COST_F32"); return e && atoi(e) != 0; }();
    // 128-row workgroup tiles when they divide M (the 128- and 256-row products of the third level): four full tile
    // rows per workgroup and the kernel without the fifth one - 32 registers less, no spill, no 96-row remainder tile
    const bool rows128 = !fp32_only && (g.M <= 128 || g.M % 128 == 0);
    g.tile_rows = rows128 ? 128 : mt::CT;
    const int64_t tiles = (int64_t)((g.M + g.tile_rows - 1) / g.tile_rows) * ((g.cols + mt::CT - 1) / mt::CT);
    PATS_REQUIRE(tiles < (1ll << 31) && g.cols < (1ll << 31), "attentional_propagation: grid too large (split the batch)");
    const dim3 grid((unsigned)tiles), block(256);
    if (fp32_only) hipLaunchKernelGGL((conv1x1_kernel<false, true>), grid, block, 0, st, g);
    else if (rows128) hipLaunchKernelGGL((conv1x1_kernel<true, false>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((conv1x1_kernel<true, true>), grid, block, 0, st, g);
    return check_launch("conv1x1_kernel");
}

}  // namespace pats

using namespace pats;

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t pats_attentional_propagation_workspace_bytes(int64_t batch, int C, int n, int m) {
    if (batch < 0 || C <= 0 || n <= 0 || m <= 0) return 0;
    const size_t qb = al256((size_t)batch * C * n * sizeof(float)), kb = al256((size_t)batch * C * m * sizeof(float));
    // q, attention output, message [b,C,n]; k, v [b,C,m]; hidden [b,2C,n]; BN scale / shift [2C] each; BN partial sums
    return 3 * qb + 2 * kb + al256((size_t)batch * 2 * C * n * sizeof(float)) + 2 * al256((size_t)2 * C * sizeof(float)) +
           al256((size_t)2 * C * pats::BN_SPLITS * 2 * sizeof(double));
}

extern "C" int pats_attentional_propagation_f32(const float* x, const float* source, int64_t batch, int C, int heads,
                                                int n, int m, const pats_propagation_weights* w, int bn_train,
                                                float bn_eps, const float* residual, float* out, void* workspace,
                                                size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && C > 0 && heads > 0 && n > 0 && m > 0 && (C % heads) == 0,
                 "attentional_propagation: bad shape");
    PATS_REQUIRE((C % 8) == 0, "attentional_propagation: feature_dim must be a multiple of 8 (operand slabs of 8 channels)");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(x && source && w && out, "attentional_propagation: null pointer");
    PATS_REQUIRE(w->wq_t && w->bq && w->wk_t && w->bk && w->wv_t && w->bv && w->wm_t && w->bm && w->w1_t && w->b1 &&
                     w->w2_t && w->b2 && w->bn_a && w->bn_b, "attentional_propagation: null weight pointer");
    PATS_REQUIRE(workspace && workspace_bytes >= pats_attentional_propagation_workspace_bytes(batch, C, n, m),
                 "attentional_propagation: workspace too small");
    hipStream_t st = as_stream(stream);
    char* p = (char*)workspace;
    const size_t qb = al256((size_t)batch * C * n * sizeof(float)), kb = al256((size_t)batch * C * m * sizeof(float));
    float* q = (float*)p; p += qb;
    float* att = (float*)p; p += qb;
    float* msg = (float*)p; p += qb;
    float* k = (float*)p; p += kb;
    float* v = (float*)p; p += kb;
    float* hid = (float*)p; p += al256((size_t)batch * 2 * C * n * sizeof(float));
    float* bsc = (float*)p; p += al256((size_t)2 * C * sizeof(float));
    float* bsh = (float*)p; p += al256((size_t)2 * C * sizeof(float));
    double* bpart = (double*)p;
    int rc;
    // projections (modules.py:101-102)
    if ((rc = launch_conv(ConvArgs{w->wq_t, x, nullptr, C, 0, C, n, batch * n, nullptr, nullptr, w->bq, nullptr, q}, st))) return rc;
    if ((rc = launch_conv(ConvArgs{w->wk_t, source, nullptr, C, 0, C, m, batch * m, nullptr, nullptr, w->bk, nullptr, k}, st))) return rc;
    if ((rc = launch_conv(ConvArgs{w->wv_t, source, nullptr, C, 0, C, m, batch * m, nullptr, nullptr, w->bv, nullptr, v}, st))) return rc;
    // attention core (:103): the [b, C, n] projections ARE the [b, dim, heads, n] views
    if ((rc = pats_attention_f32(q, k, v, batch, C / heads, heads, n, m, att, nullptr, stream))) return rc;
    // merge (:104)
    if ((rc = launch_conv(ConvArgs{w->wm_t, att, nullptr, C, 0, C, n, batch * n, nullptr, nullptr, w->bm, nullptr, msg}, st))) return rc;
    // mlp[0] on cat([x, message]) without the cat (:116, MLP :64)
    if ((rc = launch_conv(ConvArgs{w->w1_t, x, msg, C, C, 2 * C, n, batch * n, nullptr, nullptr, w->b1, nullptr, hid}, st))) return rc;
    // mlp[1] BatchNorm1d: eval -> the caller's folded running statistics (bn_a = scale, bn_b = shift);
    //                      train -> batch statistics with bn_a = gamma, bn_b = beta
    const float *sc = w->bn_a, *sh = w->bn_b;
    if (bn_train) {
        hipLaunchKernelGGL(bn_partial_kernel, dim3((unsigned)(2 * C), BN_SPLITS), dim3(256), 0, st, hid, batch, 2 * C, n, bpart);
        hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)((2 * C + 63) / 64)), dim3(64), 0, st, bpart, batch * (int64_t)n, 2 * C,
                           w->bn_a, w->bn_b, bn_eps, bsc, bsh);
        if ((rc = check_launch("bn_stats kernels"))) return rc;
        sc = bsc; sh = bsh;
    }
    // mlp[2] ReLU + mlp[3] Conv1d(2C, C), BN affine + ReLU applied while staging; optional residual (desc + delta, :133)
    return launch_conv(ConvArgs{w->w2_t, hid, nullptr, 2 * C, 0, C, n, batch * n, sc, sh, w->b2, residual, out}, st);
}
