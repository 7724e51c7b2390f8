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

#include <algorithm>
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
    int* redo;                 // conv_lean_kernel: set to 1 by a workgroup whose outputs came out non-finite;
                               // conv1x1_kernel: if non-null, the whole launch is a no-op unless *redo != 0
    const int* gate;           // optional: every kernel of the launch is a no-op unless *gate != 0 (the composition as the
                               // fallback behind the fused layer of gnn_fused.hip)
    int64_t tiles;             // conv1x1_kernel: workgroup tiles of the product (a gated / redo launch covers them with a capped grid)
};

// column addressing for mfma_tile.hpp's CmSrc: A = transposed weights at output rows i0.. (clamped), B = activations at
// flattened (batch, token) columns j0..: tile column c -> element offset of channel 0 in a [batch, Kc, n] tensor.  The
// decomposition is 32-bit (launch_conv requires batch * n < 2^31) and done once per item, not per chunk.
struct ConvCols {
    static constexpr bool stream = false;
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
    if (g.gate && *g.gate == 0) return;
    if (g.redo && *g.redo == 0) return;              // the second, normally empty, launch behind conv_lean_kernel
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int64_t tiles_j = (g.cols + mt::CT - 1) / mt::CT;
    // (grid-stride: a plain launch has one workgroup per tile; a launch that normally leaves at its gate - the redo behind
    //  conv_lean_kernel, the composition behind a fused layer - comes with a capped grid: ~2 600 of them a step in the bench's
    //  with-GNN leg cost 5 us each as full grids of workgroups that do nothing)
    for (int64_t vb = blockIdx.x; vb < g.tiles; vb += gridDim.x) {
    const int i0 = (int)(vb / tiles_j) * g.tile_rows;
    const int64_t j0 = (int64_t)(vb % tiles_j) * mt::CT;
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
    wg_barrier();                                    // the staging LDS is free for the next tile
    }
}

// The same product for the shapes the GNN actually has - many short problems, 128..528 channels - as a LEAN tile: a
// 256-thread workgroup owns 128 output rows x 64 flattened columns, wave w the 32 rows 32 w.. of both 32-column tiles:
// 32 accumulator registers instead of 80-112, 128 registers in all, so FOUR workgroups (16 waves) share a CU where
// conv1x1_kernel fits two.  Counters and an occupancy sweep on conv1x1_kernel (one workgroup per CU: 1.5x slower than
// two) say these products wait on their operand stream, not on the matrix pipe; more waves in flight is what helps.
// fp16-split contraction as mfma_tile.hpp (same split, same fragment layout in LDS, same MFMA order), chunks of 16
// channels, three 4-channel items per thread (two of weights, one of activations), double-buffered LDS, one barrier
// per chunk, two reduction passes for the MLP's (x | message) product.  No fp32 path in here: a workgroup that finds a
// non-finite value among its outputs raises *redo, and the launch of conv1x1_kernel queued right behind this one (a
// no-op otherwise) recomputes the whole product with its own in-kernel fp32 redo.
namespace {
constexpr int LR = 128;
template <int NT>
struct __attribute__((aligned(16))) LeanLds {
    uint2 a[2][2][4][LR];            // [buffer][hi | lo][channel / 4][output row]
    uint2 b[2][2][4][32 * NT];       // [buffer][hi | lo][channel / 4][column]
};
}  // namespace

// NT = 32-column tiles per wave (2 or 4): the workgroup tile is 128 rows x 32 NT columns
template <int NT>
__global__ void __launch_bounds__(256, NT == 2 ? 4 : 3)
conv_lean_kernel(ConvArgs g) {
    constexpr int LC = 32 * NT, NB = NT / 2;         // columns per workgroup, activation items per thread
    __shared__ LeanLds<NT> lds;
    if (g.gate && *g.gate == 0) return;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, kg = lane >> 5;
    // (an XCD-aware tile order - each XCD a contiguous run of column tiles, so that the two halves of a 128-byte line
    // shared by neighbouring tiles meet in one L2 - was measured: no difference)
    const int64_t tiles_j = (g.cols + LC - 1) / LC;
    const int i0 = (int)(blockIdx.x / tiles_j) * LR;
    const int64_t j0 = (int64_t)(blockIdx.x % tiles_j) * LC;
    const int n = g.n, M = g.M;

    // staging: items 0, 1 = weights (channel quad id / 128, output row id % 128, id = t + 256 q), items 2.. = activations
    // (id = t + 256 q: channel quad id / LC, column id % LC): running pointers, advanced by 16 channels per chunk
    const int aq0 = t / LR, ar = t % LR, aq1 = aq0 + 2;
    const int arow = min(i0 + ar, M - 1);
    int bq[NB], bc[NB];
    unsigned bsq[NB], tkq[NB];                       // (batch, token) of the item's column: 32-bit, once
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int id = t + 256 * q;
        bq[q] = id / LC;
        bc[q] = id % LC;
        const unsigned cgs = (unsigned)min(j0 + bc[q], g.cols - 1);
        bsq[q] = cgs / (unsigned)n;
        tkq[q] = cgs - bsq[q] * (unsigned)n;
    }
    const float *pa0, *pa1, *pb[NB];
    float r[2 + NB][4];
    mt::f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

    auto fraga = [&](int buf, int hl, int idx) {                  // planes [quad][row]: quads 2 kg, 2 kg + 1
        const uint2 e0 = lds.a[buf][hl][2 * kg][idx], e1 = lds.a[buf][hl][2 * kg + 1][idx];
        return __builtin_bit_cast(mt::h8c, mt::u4c{e0.x, e0.y, e1.x, e1.y});
    };
    auto fragb = [&](int buf, int hl, int idx) {
        const uint2 e0 = lds.b[buf][hl][2 * kg][idx], e1 = lds.b[buf][hl][2 * kg + 1][idx];
        return __builtin_bit_cast(mt::h8c, mt::u4c{e0.x, e0.y, e1.x, e1.y});
    };

    // one reduction pass over K channels: weights rows k_w0.., activations x (K channels per batch entry)
    auto pass = [&](int k_w0, const float* x, int K, const float* scale, const float* shift) {
        pa0 = g.wt + (int64_t)(k_w0 + 4 * aq0) * M + arow;
        pa1 = g.wt + (int64_t)(k_w0 + 4 * aq1) * M + arow;
#pragma unroll
        for (int q = 0; q < NB; ++q) pb[q] = x + ((int64_t)bsq[q] * K + 4 * bq[q]) * n + tkq[q];
        auto fetch = [&](int k0) {
            if (k0 + 16 <= K) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r[0][e] = pa0[e * M];
                    r[1][e] = pa1[e * M];
#pragma unroll
                    for (int q = 0; q < NB; ++q) r[2 + q][e] = pb[q][e * n];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {          // ragged last chunk: read a valid channel, then zero
                    const int ka0 = k0 + 4 * aq0 + e, ka1 = k0 + 4 * aq1 + e;
                    const float v0 = pa0[(min(ka0, K - 1) - (ka0 - e)) * M], v1 = pa1[(min(ka1, K - 1) - (ka1 - e)) * M];
                    r[0][e] = ka0 < K ? v0 : 0.f;
                    r[1][e] = ka1 < K ? v1 : 0.f;
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        const int kb = k0 + 4 * bq[q] + e;
                        const float v2 = pb[q][(min(kb, K - 1) - (kb - e)) * n];
                        r[2 + q][e] = kb < K ? v2 : 0.f;
                    }
                }
            }
            if (scale) {
#pragma unroll
                for (int q = 0; q < NB; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kb = k0 + 4 * bq[q] + e;
                        if (kb < K) r[2 + q][e] = fmaxf(fmaf(r[2 + q][e], scale[kb], shift[kb]), 0.f);
                    }
            }
            pa0 += 16 * M; pa1 += 16 * M;
#pragma unroll
            for (int q = 0; q < NB; ++q) pb[q] += 16 * n;
        };
        auto stash = [&](int buf) {
            uint2 hi, lo;
            mt::split2(r[0][0], r[0][1], hi.x, lo.x); mt::split2(r[0][2], r[0][3], hi.y, lo.y);
            lds.a[buf][0][aq0][ar] = hi; lds.a[buf][1][aq0][ar] = lo;
            mt::split2(r[1][0], r[1][1], hi.x, lo.x); mt::split2(r[1][2], r[1][3], hi.y, lo.y);
            lds.a[buf][0][aq1][ar] = hi; lds.a[buf][1][aq1][ar] = lo;
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                mt::split2(r[2 + q][0], r[2 + q][1], hi.x, lo.x); mt::split2(r[2 + q][2], r[2 + q][3], hi.y, lo.y);
                lds.b[buf][0][bq[q]][bc[q]] = hi; lds.b[buf][1][bq[q]][bc[q]] = lo;
            }
        };
        const int nchunk = (K + 15) / 16;
        fetch(0);
        stash(0);
        wg_barrier();
        for (int c = 0; c < nchunk; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunk) fetch((c + 1) * 16);
            // phases kept apart as in mfma_tile.hpp: fragments, the MFMAs (pass-major: consecutive ones never depend on
            // each other), wait states, then the split of the next chunk
            __builtin_amdgcn_sched_barrier(0);
            const mt::h8c ah = fraga(buf, 0, 32 * wave + li), al = fraga(buf, 1, 32 * wave + li);
            mt::h8c bh[NT], bl[NT];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) { bh[tj] = fragb(buf, 0, 32 * tj + li); bl[tj] = fragb(buf, 1, 32 * tj + li); }
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[tj], acc[tj], 0, 0, 0);
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[tj], acc[tj], 0, 0, 0);
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[tj], acc[tj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            asm volatile("" :: "v"(ah), "v"(al));
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) asm volatile("" :: "v"(bh[tj]), "v"(bl[tj]));
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < nchunk) stash(buf ^ 1);
            wg_barrier();
        }
    };
    pass(0, g.x0, g.K0, g.in_scale, g.in_shift);
    if (g.K1 > 0) pass(g.K0, g.x1, g.K1, g.in_scale ? g.in_scale + g.K0 : nullptr, g.in_shift ? g.in_shift + g.K0 : nullptr);

    // C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    bool bad = false;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        if (j0 + 32 * tj + li >= g.cols) continue;
        const unsigned cg = (unsigned)(j0 + 32 * tj + li), b = cg / (unsigned)n, tk = cg - b * (unsigned)n;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * kg;
            if (row < M) {
                const int64_t o = ((int64_t)b * M + row) * n + tk;
                float v = acc[tj][e] * mt::UNSCALE;
                bad |= !(fabsf(v) <= 3.0e38f);
                if (g.bias) v += g.bias[row];
                if (g.residual) v = g.residual[o] + v;
                g.y[o] = v;
            }
        }
    }
    if (__any(bad) && lane == 0) atomicOr(g.redo, 1);
}

// WEIGHTS-STATIONARY variant of the lean tile for the third level's C x C products (128 channels in, one source, many
// column tiles).  In conv_lean_kernel two thirds of the loads in flight re-fetch the 64 KB weight matrix for every 64-column
// tile (counters: matrix pipe 11 % busy, waves waiting 71 % of their cycles).  Here ONE 1024-thread workgroup per CU - the
// same 16 waves - splits its 128 x 128 weights ONCE into LDS (hi | lo planes, 64 KB) and walks column tiles; its four
// 256-thread groups each own a 64-column tile and stream only activations, three 16-channel chunks in flight per thread
// (a ring of four register slots), through a double-buffered 4 KB LDS stage per group.  The chunk
// stream runs on across tile boundaries.  Same split, same fragment layout, same MFMA order and the same redo protocol
// as conv_lean_kernel: the results are bit-identical to it.
namespace {
constexpr int WS_LC = 64;                            // columns per group tile
}  // namespace

template <int WS_NCH, bool TWO>                     // chunks of 16 channels: K = 128 (8) or 256 (16); TWO: x | message, 128 + 128
__global__ void __launch_bounds__(1024)
conv_ws_kernel(ConvArgs g, int row_tiles) {
    constexpr int WS_KQ = 4 * WS_NCH;                // channel quads
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_raw[];
    if (g.gate && *g.gate == 0) return;
    uint2* la = reinterpret_cast<uint2*>(ws_raw);                    // [hi | lo][WS_KQ][LR]
    uint2* lb_all = la + (size_t)2 * WS_KQ * LR;                     // [group][buffer][hi | lo][4][WS_LC]
    const int t = threadIdx.x, grp = __builtin_amdgcn_readfirstlane(t >> 8), tt = t & 255;
    const int lane = tt & 63, wave = __builtin_amdgcn_readfirstlane(tt >> 6);
    const int li = lane & 31, kg = lane >> 5;
    const int n = g.n, M = g.M;
    constexpr int K = 16 * WS_NCH;
    constexpr int NCH0 = TWO ? WS_NCH / 2 : WS_NCH;  // chunks of the first source (x | message: the second follows)
    const int i0 = (int)(blockIdx.x % row_tiles) * LR;
    uint2* lb = lb_all + (size_t)grp * 2 * 2 * 4 * WS_LC;
    // ---- the weights (rows 0..127 of the layer: M <= 128), all K channels, split once --------------------------------------
    {
        const int ar = t % LR, aq = t / LR;          // 1024 threads: rows x quads {aq} of every 32-channel step
        const int arow = min(i0 + ar, M - 1);
        for (int k0 = 0; k0 < K; k0 += 32) {
            const int q = k0 / 4 + aq;
            const float* pw = g.wt + (int64_t)(4 * q) * M + arow;
            uint2 hi, lo;
            mt::split2(pw[0], pw[M], hi.x, lo.x);
            mt::split2(pw[2 * (int64_t)M], pw[3 * (int64_t)M], hi.y, lo.y);
            la[(0 * WS_KQ + q) * LR + ar] = hi;
            la[(1 * WS_KQ + q) * LR + ar] = lo;
        }
    }
    auto fraga = [&](int hl, int c, int idx) {       // chunk c: quads 4c + 2 kg, + 1
        const uint2 e0 = la[(hl * WS_KQ + 4 * c + 2 * kg) * LR + idx], e1 = la[(hl * WS_KQ + 4 * c + 2 * kg + 1) * LR + idx];
        return __builtin_bit_cast(mt::h8c, mt::u4c{e0.x, e0.y, e1.x, e1.y});
    };
    auto fragb = [&](int buf, int hl, int idx) {
        const uint2* base = lb + ((size_t)(buf * 2 + hl) * 4 + 2 * kg) * WS_LC;
        const uint2 e0 = base[idx], e1 = base[WS_LC + idx];
        return __builtin_bit_cast(mt::h8c, mt::u4c{e0.x, e0.y, e1.x, e1.y});
    };
    const int64_t tiles_j = (g.cols + WS_LC - 1) / WS_LC;
    const int64_t step = (int64_t)(gridDim.x / row_tiles) * 4;
    const int bqd = tt / WS_LC, bc = tt % WS_LC;     // this thread's activation item: channel quad bqd of every chunk, column bc
    // element offset of channel 4 bqd of this thread's column in column tile ct - the same in both sources (equal shapes);
    // 32 bits (launch_conv: batch * channels * n < 2^31), added to the wave-uniform base pointers
    auto tile_off = [&](int64_t ct) {
        const unsigned cgs = (unsigned)min(ct * WS_LC + bc, g.cols - 1), bs = cgs / (unsigned)n, tk = cgs - bs * (unsigned)n;
        return (bs * (unsigned)g.K0 + 4u * bqd) * (unsigned)n + tk;
    };
    float R[4][4];                                   // a ring of four chunks: 8 / 16 chunks per tile, so the slots line up across tiles
    auto fetch = [&](unsigned off, int c, float (&r)[4]) {
        const float* base = (TWO && c >= NCH0) ? g.x1 : g.x0;
        const int cc = (TWO && c >= NCH0) ? c - NCH0 : c;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = base[off + (unsigned)((16 * cc + e) * n)];
    };
    auto stash = [&](int buf, int c, const float (&r)[4]) {
        float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
        if (g.in_scale) {
            const int kb = 16 * c + 4 * bqd;
            r0 = fmaxf(fmaf(r0, g.in_scale[kb], g.in_shift[kb]), 0.f);
            r1 = fmaxf(fmaf(r1, g.in_scale[kb + 1], g.in_shift[kb + 1]), 0.f);
            r2 = fmaxf(fmaf(r2, g.in_scale[kb + 2], g.in_shift[kb + 2]), 0.f);
            r3 = fmaxf(fmaf(r3, g.in_scale[kb + 3], g.in_shift[kb + 3]), 0.f);
        }
        uint2 hi, lo;
        mt::split2(r0, r1, hi.x, lo.x); mt::split2(r2, r3, hi.y, lo.y);
        lb[((size_t)(buf * 2 + 0) * 4 + bqd) * WS_LC + bc] = hi;
        lb[((size_t)(buf * 2 + 1) * 4 + bqd) * WS_LC + bc] = lo;
    };
    bool bad = false;
    // every group walks the same number of steps (the barriers are workgroup-wide); a group past the end works on a
    // clamped tile and stores nothing
    const int64_t first = (int64_t)(blockIdx.x / row_tiles) * 4 + grp;
    const int64_t nsteps = (tiles_j + step - 1) / step;
    unsigned pcur = tile_off(min(first, tiles_j - 1));
    fetch(pcur, 0, R[0]); fetch(pcur, 1, R[1]); fetch(pcur, 2, R[2]);
    wg_barrier();                                 // the weights
    stash(0, 0, R[0]);
    wg_barrier();
    for (int64_t it = 0; it < nsteps; ++it) {
        const int64_t ct = first + it * step;
        const bool live = ct < tiles_j;
        const int64_t j0 = min(ct, tiles_j - 1) * WS_LC;
        const unsigned pnext = tile_off(min(ct + step, tiles_j - 1));
        mt::f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
#pragma unroll 1
        for (int c0 = 0; c0 < WS_NCH; c0 += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) {                // four chunks unrolled: the ring slots stay compile-time
            const int c = c0 + u;
            const int buf = u & 1;
            // chunk c + 3 of the stream (this tile's, or the next tile's first chunks) into the slot chunk c - 1 left
            if (c + 3 < WS_NCH) fetch(pcur, c + 3, R[(u + 3) % 4]);
            else fetch(pnext, c + 3 - WS_NCH, R[(u + 3) % 4]);
            __builtin_amdgcn_sched_barrier(0);
            const mt::h8c ah = fraga(0, c, 32 * wave + li), al = fraga(1, c, 32 * wave + li);
            mt::h8c bh[2], bl[2];
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) { bh[tj] = fragb(buf, 0, 32 * tj + li); bl[tj] = fragb(buf, 1, 32 * tj + li); }
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[tj], acc[tj], 0, 0, 0);
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[tj], acc[tj], 0, 0, 0);
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[tj], acc[tj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            asm volatile("" :: "v"(ah), "v"(al));
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) asm volatile("" :: "v"(bh[tj]), "v"(bl[tj]));
            __builtin_amdgcn_sched_barrier(0);
            // the next chunk of the stream (chunk c + 1, or chunk 0 of the next tile) into the other LDS buffer
            stash(buf ^ 1, (c + 1) % WS_NCH, R[(u + 1) % 4]);
            wg_barrier();
        }
        if (live) {
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                if (j0 + 32 * tj + li >= g.cols) continue;
                const unsigned cg = (unsigned)(j0 + 32 * tj + li), b = cg / (unsigned)n, tkk = cg - b * (unsigned)n;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * kg;
                    if (row < M) {
                        const int64_t o = ((int64_t)b * M + row) * n + tkk;
                        float v = acc[tj][e] * mt::UNSCALE;
                        bad |= !(fabsf(v) <= 3.0e38f);
                        if (g.bias) v += g.bias[row];
                        if (g.residual) v = g.residual[o] + v;
                        g.y[o] = v;
                    }
                }
            }
        }
        pcur = pnext;
    }
    if (__any(bad) && lane == 0) atomicOr(g.redo, 1);
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
bn_partial_kernel(const float* __restrict__ h, int64_t batch, int C, int n, double* __restrict__ part,
                  const int* __restrict__ gate = nullptr) {
    __shared__ double s1[4], s2[4];
    if (gate && *gate == 0) return;
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
    wg_barrier();
    if (threadIdx.x == 0) {
        part[((int64_t)c * BN_SPLITS + split) * 2 + 0] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
        part[((int64_t)c * BN_SPLITS + split) * 2 + 1] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    }
}

// The same partial sums over a channel-BLOCKED tensor [batch, C / 8, n, 8] (the hidden tensor between the packed-weights
// convolutions, csrc/conv_pk.hip): grid (C / 8, BN_SPLITS); a batch's block is n x 8 contiguous floats, thread t walks the elements
// t, t + 256, .. - always channel t & 7 - and the eight lanes' classes are folded with a fixed butterfly.
__global__ void __launch_bounds__(256)
bn_partial_blocked_kernel(const float* __restrict__ h, int64_t batch, int C, int n, double* __restrict__ part,
                          const int* __restrict__ gate = nullptr) {
    __shared__ double s1[4][8], s2[4][8];
    if (gate && *gate == 0) return;
    const int cb = blockIdx.x, split = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 0.0, q = 0.0;
    for (int64_t b = (int64_t)split * 4 + wave; b < batch; b += 4 * BN_SPLITS) {
        const float* blk = h + (b * (C >> 3) + cb) * (int64_t)n * 8;
        for (int t = lane; t < n * 8; t += 64) {
            const double x = (double)blk[t];
            a += x;
            q += x * x;
        }
    }
#pragma unroll
    for (int o = 32; o >= 8; o >>= 1) {         // fixed butterfly order over the lanes of one channel (lane & 7)
        a += __shfl_xor(a, o);
        q += __shfl_xor(q, o);
    }
    if (lane < 8) { s1[wave][lane] = a; s2[wave][lane] = q; }
    wg_barrier();
    if (threadIdx.x < 8) {
        const int c = cb * 8 + threadIdx.x, e = threadIdx.x;
        part[((int64_t)c * BN_SPLITS + split) * 2 + 0] = (s1[0][e] + s1[1][e]) + (s1[2][e] + s1[3][e]);
        part[((int64_t)c * BN_SPLITS + split) * 2 + 1] = (s2[0][e] + s2[1][e]) + (s2[2][e] + s2[3][e]);
    }
}

__global__ void __launch_bounds__(64)
bn_finish_kernel(const double* __restrict__ part, int64_t total, int C, const float* __restrict__ gamma,
                 const float* __restrict__ beta, float eps, float* __restrict__ scale, float* __restrict__ shift,
                 const int* __restrict__ gate = nullptr, int splits = BN_SPLITS) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C || (gate && *gate == 0)) return;
    double a = 0.0, q = 0.0;
    for (int s_ = 0; s_ < splits; ++s_) {
        a += part[((int64_t)c * splits + s_) * 2 + 0];
        q += part[((int64_t)c * splits + s_) * 2 + 1];
    }
    const double mean = a / (double)total;
    double var = q / (double)total - mean * mean;
    if (var < 0.0) var = 0.0;
    const float sc = gamma[c] / sqrtf((float)var + eps);
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
}

// `redo`: one int of workspace, zero on entry (conv_lean_kernel raises it; see there).  null -> conv1x1_kernel only.
int launch_conv_pk(const void* packed, const float* x0, const float* x1, int K0, int K1, int M, int n, int64_t cols,
                   const float* in_scale, const float* in_shift, const float* bias, const float* residual, float* y, int* redo,
                   const int* gate, hipStream_t st, int layout);                                            // conv_pk.hip
bool conv_pk_ready();
size_t conv_packed_bytes(int K, int M);
size_t packed_fused_bytes(int C, int heads);                                                                // gnn_fused.hip
bool gnn_fold_enabled();                                                                                    // (merge folded into mlp[0])
const float* packed_folded_bias(const void* packed, int C, int heads);

static int launch_conv(const ConvArgs& g0, int* redo, hipStream_t st, const int* gate = nullptr) {
    ConvArgs g = g0;
    g.gate = gate;
    const bool fp32_only = cost_f32_only();
    static const bool no_lean = [] { const char* e = diag_env("PATS_CONV_LEAN"); return e && atoi(e) == 0; }();    // A/B switch
    PATS_REQUIRE(g.cols < (1ll << 31), "attentional_propagation: grid too large (split the batch)");
    const bool lean = redo && !fp32_only && !no_lean && !(g.residual && g.residual == g.y);
    // weights-stationary tile: exactly 128 channels in from one source, at most 128 out, and enough column tiles for a
    // persistent grid of 4 x CUs groups to pay (the third level: 26 325 tiles)
    static const int ws_mode = [] { const char* e = env_switch("PATS_CONV_WS"); return e ? atoi(e) : 1; }();     // A/B switch: 0 off, 2 = also on small grids (tests)
    const int Kt = g.K0 + g.K1;
    const bool two = g.K1 > 0;
    // the tile needs up to the CU's whole 160 KB of LDS as dynamic shared memory: if the runtime will not grant it, the lean
    // tile computes the same bits
    // Both the attribute and the CU count belong to a DEVICE: cached per device id (one process may drive several GPUs;
    // attention.hip sets its attribute on every launch for the same reason), never process-wide.
    struct PerDevice { int state = 0; int n_cu = 256; };      // state: 0 unknown, 1 granted, -1 refused
    static PerDevice per_dev[64];
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) { (void)hipGetLastError(); dev_id = 0; }
    PerDevice& pd = per_dev[dev_id];
    if (pd.state == 0) {
        const bool ok = hipFuncSetAttribute((const void*)conv_ws_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                        hipFuncSetAttribute((const void*)conv_ws_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                        hipFuncSetAttribute((const void*)conv_ws_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        int v = 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess) (void)hipGetLastError();
        pd.n_cu = v > 0 ? v : 256;
        pd.state = ok ? 1 : -1;
    }
    const bool ws_lds_ok = pd.state == 1;
    const bool ws = lean && ws_mode != 0 && ws_lds_ok && ((Kt == 128 && !two) || (Kt == 256 && (!two || g.K0 == 128))) &&
                    g.cols * (int64_t)std::max(g.K0, 1) < (1ll << 31) && (g.cols >= 64 * 4096 || ws_mode == 2);
    if (ws) {
        const int n_cu = pd.n_cu;
        // weights of one 128-row tile (hi | lo planes, K x 512 bytes) + four groups' double-buffered 4 KB stages: 96 KB at
        // K = 128, the CU's whole 160 KB at K = 256
        const size_t lds = (size_t)2 * (Kt / 4) * LR * sizeof(uint2) + (size_t)4 * 2 * 2 * 4 * WS_LC * sizeof(uint2);
        const int row_tiles = (g.M + LR - 1) / LR;
        const int64_t tiles = (g.cols + WS_LC - 1) / WS_LC;
        int grid = (int)std::min<int64_t>(n_cu, ((tiles + 3) / 4) * row_tiles);
        grid = std::max(row_tiles, grid - grid % row_tiles);       // whole groups of row tiles
        g.redo = redo;
        if (Kt == 128) hipLaunchKernelGGL((conv_ws_kernel<8, false>), dim3((unsigned)grid), dim3(1024), lds, st, g, row_tiles);
        else if (!two) hipLaunchKernelGGL((conv_ws_kernel<16, false>), dim3((unsigned)grid), dim3(1024), lds, st, g, row_tiles);
        else hipLaunchKernelGGL((conv_ws_kernel<16, true>), dim3((unsigned)grid), dim3(1024), lds, st, g, row_tiles);
    } else if (lean) {
        static const int nt = diag_env("PATS_CONV_NT") ? atoi(diag_env("PATS_CONV_NT")) : 2;       // A/B switch: 2 or 4 column tiles
        const int lc = 32 * (nt == 4 ? 4 : 2);
        const int64_t lt = (int64_t)((g.M + LR - 1) / LR) * ((g.cols + lc - 1) / lc);
        PATS_REQUIRE(lt < (1ll << 31), "attentional_propagation: grid too large (split the batch)");
        g.redo = redo;
        if (nt == 4) hipLaunchKernelGGL(conv_lean_kernel<4>, dim3((unsigned)lt), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(conv_lean_kernel<2>, dim3((unsigned)lt), dim3(256), 0, st, g);
    } else {
        g.redo = nullptr;
    }
    // 128-row workgroup tiles when they divide M (the 128- and 256-row products of the third level): four full tile
    // rows per workgroup and the kernel without the fifth one - 32 registers less, no spill, no 96-row remainder tile
    const bool rows128 = !fp32_only && (g.M <= 128 || g.M % 128 == 0);
    g.tile_rows = rows128 ? 128 : mt::CT;
    const int64_t tiles = (int64_t)((g.M + g.tile_rows - 1) / g.tile_rows) * ((g.cols + mt::CT - 1) / mt::CT);
    PATS_REQUIRE(tiles < (1ll << 31), "attentional_propagation: grid too large (split the batch)");
    g.tiles = tiles;
    const dim3 grid((unsigned)((g.gate || g.redo) ? std::min<int64_t>(tiles, 2048) : tiles)), block(256);
    if (fp32_only) hipLaunchKernelGGL((conv1x1_kernel<false, true>), grid, block, 0, st, g);
    else if (rows128) hipLaunchKernelGGL((conv1x1_kernel<true, false>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((conv1x1_kernel<true, true>), grid, block, 0, st, g);
    return check_launch("conv1x1_kernel");
}

}  // namespace pats

using namespace pats;

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

namespace pats {
int fine_layer_supported(int C, int heads, int n, int m);          // gnn_fine.hip
size_t fine_scratch_bytes(int64_t P);
}

extern "C" size_t pats_attentional_propagation_workspace_bytes(int64_t batch, int C, int n, int m) {
    if (batch < 0 || C <= 0 || n <= 0 || m <= 0) return 0;
    const size_t qb = al256((size_t)batch * C * n * sizeof(float)), kb = al256((size_t)batch * C * m * sizeof(float));
    // q, attention output, message [b,C,n]; k, v [b,C,m]; hidden [b,2C,n]; BN scale / shift [2C] each; BN partial sums
    // (BatchNorm partial sums: BN_SPLITS per channel for the composition, one per workgroup of the fused kernel's grid - at most 512)
    // (+ the fine level's one-kernel layer: its per-workgroup scratch blocks; its descriptor images overlay q / att / msg / k)
    const size_t fine = pats::fine_layer_supported(C, 4, n, m) ? al256(pats::fine_scratch_bytes(batch)) : 0;
    return 3 * qb + 2 * kb + al256((size_t)batch * 2 * C * n * sizeof(float)) + 2 * al256((size_t)2 * C * sizeof(float)) +
           al256((size_t)2 * C * 512 * 2 * sizeof(double)) + 256 /* seven redo flags + the fused layer's flag */ + fine;
}

namespace pats {
int launch_attention(const float* query, const float* key, const float* value, int64_t batch, int dim, int heads, int n, int m,
                     float* out, float* prob, pats_stream_t stream, const int* gate);                     // attention.hip
int launch_attention145(const float* query, const float* key, const float* value, int64_t batch, int dim, int heads, int n, int m,
                        float* out, int* flag, const int* gate, hipStream_t st);                          // attention145.hip
int fused_layer_supported(int C, int heads, int n, int m);                                                // gnn_fused.hip
int launch_fused_layer(const float* x, const float* source, int64_t batch, const void* packed, const float* bn_a, const float* bn_b,
                       int bn_train, const float* residual, float* out, float* hid, int* flag, double* bn_part, int* splits_out,
                       hipStream_t st, const int64_t* live = nullptr, int64_t live_off = 0);
int launch_gnn_tail(const float* hid, int64_t batch, const void* packed, const float* scale, const float* shift, const float* residual,
                    float* out, int* flag, hipStream_t st);
// the fine level's one-kernel layer (gnn_fine.hip)
int fine_layer_supported(int C, int heads, int n, int m);
const void* packed_fine_section(const void* packed, int C, int heads);                                   // gnn_fused.hip
size_t fine_scratch_bytes(int64_t P);
size_t fine_image_bytes(int64_t P);
int launch_fine_in(const float* x, int64_t P, char* tf, hipStream_t st);
int launch_fine_out(const char* tf, int64_t P, float* y, hipStream_t st, const int64_t* live = nullptr, int64_t live_off = 0,
                    const float* add = nullptr);
int launch_fine_layer(const char* tf_x, const char* tf_s, int64_t shift, int residual, int64_t P, const void* section,
                      char* tf_out, char* tf_att, char* qkv, int* flag, const int* gate, hipStream_t st, int sets,
                      const int64_t* live, int64_t live_off);
int launch_fine_qkv(const char* tf, int64_t P, const void* section, char* qkv, int want_q, int want_kv, const int* gate, hipStream_t st, int sets,
                    const int64_t* live, int64_t live_off);
int launch_fine_attn(const char* qkv, int64_t shift, char* tf_att, int64_t P, const int* gate, hipStream_t st, int sets, const int64_t* live,
                     int64_t live_off);
int launch_fine_mlp(const char* tf_x, const char* tf_att, int residual, const void* section, char* tf_out, const void* next_section, char* qkv,
                    int64_t P, int* flag, const int* gate, hipStream_t st, int sets, const int64_t* live, int64_t live_off);
}

static int propagation_impl(const float* x, const float* source, int64_t batch, int C, int heads, int n, int m,
                            const pats_propagation_weights* w, int bn_train, float bn_eps, const float* residual, float* out,
                            void* workspace, size_t workspace_bytes, pats_stream_t stream, const void* packed,
                            const int* ext_gate = nullptr, const int64_t* live = nullptr, int64_t live_off = 0);

extern "C" int pats_attentional_propagation_f32(const float* x, const float* source, int64_t batch, int C, int heads,
                                                int n, int m, const pats_propagation_weights* w, int bn_train,
                                                float bn_eps, const float* residual, float* out, void* workspace,
                                                size_t workspace_bytes, pats_stream_t stream) {
    return propagation_impl(x, source, batch, C, heads, n, m, w, bn_train, bn_eps, residual, out, workspace, workspace_bytes, stream,
                            nullptr);
}

// The same layer with the weights additionally handed over PACKED (pats_propagation_pack_f32, once per layer): at the third
// level's shape (C = 128, 4 heads, n = m = 65) the whole layer then runs as ONE kernel (gnn_fused.hip; on batch statistics: one
// kernel up to the hidden tensor, then the statistics passes and the last convolution of the composition).  Any other shape, or
// PATS_GNN_FUSED=0, takes the composition; so does - gated on a device-side flag, its kernels return at once otherwise - a
// launch in which the fused kernel met a non-finite value (an activation beyond the fp16 range of its split operands).
extern "C" int pats_attentional_propagation_packed_f32(const float* x, const float* source, int64_t batch, int C, int heads,
                                                       int n, int m, const pats_propagation_weights* w, const void* packed,
                                                       int bn_train, float bn_eps, const float* residual, float* out,
                                                       void* workspace, size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(packed, "attentional_propagation_packed: null packed weights");
    return propagation_impl(x, source, batch, C, heads, n, m, w, bn_train, bn_eps, residual, out, workspace, workspace_bytes, stream,
                            packed);
}

// The packed layer over a CAPACITY with the problem count on the device (throughput mode: the third level's P, the fine level's row
// total): problems >= clamp(*live - live_off, 0, batch) are skipped by the one-kernel layers (third and fine level's shapes, eval
// mode); their output rows are left untouched.  Any other shape / bn_train: the count is ignored, every problem is computed.
extern "C" int pats_attentional_propagation_packed_counted_f32(const float* x, const float* source, int64_t batch, const int64_t* live,
                                                               int64_t live_off, int C, int heads, int n, int m,
                                                               const pats_propagation_weights* w, const void* packed, int bn_train,
                                                               float bn_eps, const float* residual, float* out, void* workspace,
                                                               size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(packed, "attentional_propagation_packed_counted: null packed weights");
    return propagation_impl(x, source, batch, C, heads, n, m, w, bn_train, bn_eps, residual, out, workspace, workspace_bytes, stream,
                            packed, nullptr, live, live_off);
}

static int propagation_impl(const float* x, const float* source, int64_t batch, int C, int heads, int n, int m,
                            const pats_propagation_weights* w, int bn_train, float bn_eps, const float* residual, float* out,
                            void* workspace, size_t workspace_bytes, pats_stream_t stream, const void* packed,
                            const int* ext_gate, const int64_t* live, int64_t live_off) {
    // ext_gate: run ONLY the round-2 composition, every kernel of it gated on *ext_gate (the redo chain behind a fused stack,
    // pats_attentional_gnn_packed_f32): no-ops unless that flag is raised
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
    double* bpart = (double*)p; p += al256((size_t)2 * C * 512 * 2 * sizeof(double));
    int* redo = (int*)p; p += 256;
    char* fine_scratch = p;
    // deferred overflow protocol (pats_set_gnn_redo_mode(1), host.cpp): the one-kernel / packed-weights branches raise the device's
    // sticky flag instead of a per-call one and NO gated composition is queued behind them; the caller reads the flag at its own
    // synchronisation point (pats_gnn_overflows) and repeats the work in the default mode if it is up
    const bool deferred = gnn_redo_deferred() && !ext_gate && packed;
    int* const fast_flag = deferred ? gnn_overflow_flag() : redo + 7;
    if (!deferred && fill_bytes(redo, 0, 8 * sizeof(int), st)) return PATS_ERR_LAUNCH;      // (a pats:: kernel: the steps hold no runtime fill)
    int rc;
    const int* gate = ext_gate;
    if (ext_gate) packed = nullptr;
    if (packed && !bn_train && fine_layer_supported(C, heads, n, m) && gnn_fold_enabled() && !(residual && residual == out)) {
        // The fine level (round 5, gnn_fine.hip): the whole layer in one kernel on (fp32 blocked, TF image) descriptors.  This
        // single-layer entry converts on the way in and out (pats_attentional_gnn_packed_f32 keeps a stack in that form); the
        // images and blocked copies overlay the composition's q / att / msg / k buffers, which only the gated redo would use.
        int* flag = fast_flag;
        char* tf_x = (char*)q;
        char* tf_s = source == x ? tf_x : (char*)att;
        char* tf_att = (char*)msg;
        char* tf_out = (char*)k;
        if ((rc = launch_fine_in(x, batch, tf_x, st))) return rc;
        if (source != x && (rc = launch_fine_in(source, batch, tf_s, st))) return rc;
        // the kernels add the layer's own x (what AttentionalGNN.forward does); any other residual is added by the conversion out
        rc = launch_fine_layer(tf_x, tf_s, 0, residual == x ? 1 : 0, batch, packed_fine_section(packed, C, heads), tf_out, tf_att, fine_scratch,
                               flag, nullptr, st, 1, live, live_off);
        if (rc == PATS_OK) {
            if ((rc = launch_fine_out(tf_out, batch, out, st, live, live_off, residual && residual != x ? residual : nullptr))) return rc;
            if (deferred) return PATS_OK;
            gate = flag;         // the composition below runs only if the kernel raised it
        } else if (rc != PATS_ERR_UNSUPPORTED) {
            return rc;
        }
    }
    if (packed && !gate && fused_layer_supported(C, heads, n, m) && !(residual && residual == out)) {
        // residual == out is excluded (round-4 advice): the kernel itself reads and writes each element once, but the gated redo
        // behind it would read the residual the first attempt has already overwritten - such a call takes the composition alone
        int* flag = fast_flag;
        int splits = 0;
        rc = launch_fused_layer(x, source, batch, packed, w->bn_a, w->bn_b, bn_train, residual, out, hid, flag, bpart, &splits, st, live, live_off);
        if (rc == PATS_OK) {
            if (bn_train) {      // the fused kernel stopped behind mlp[0] and left per-workgroup partial sums of the hidden tensor:
                                 // scale / shift from them, then mlp[1..3] from the hidden tensor in one kernel
                hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)((2 * C + 63) / 64)), dim3(64), 0, st, bpart, batch * (int64_t)n,
                                   2 * C, w->bn_a, w->bn_b, bn_eps, bsc, bsh, (const int*)nullptr, splits);
                if ((rc = check_launch("bn_finish_kernel"))) return rc;
                if ((rc = launch_gnn_tail(hid, batch, packed, bsc, bsh, residual, out, flag, st))) return rc;
            }
            if (deferred) return PATS_OK;
            gate = flag;         // the composition below runs only if the fused kernel raised it
        } else if (rc != PATS_ERR_UNSUPPORTED) {
            return rc;
        }
    }
    // Packed weights at any other shape (round 4): six conv_pk_kernel launches around the attention core, the message and the hidden
    // tensor - the layer's own intermediates - channel-BLOCKED ([batch, C / 8, n, 8]: 16-byte accesses both ways).  No fp32 path in
    // these kernels: a non-finite output anywhere raises ONE flag, and the round-2 composition below - gated on it, its kernels
    // return at once otherwise - redoes the layer.  (residual == out: the redo would read what the first attempt wrote.)
    static const bool no_pk = [] { const char* e = diag_env("PATS_CONV_PK"); return e && atoi(e) == 0; }();         // A/B switch
    const bool fp32_only = cost_f32_only();
    if (packed && !gate && !no_pk && !fp32_only && !(residual && residual == out) && conv_pk_ready() && batch * (int64_t)std::max(n, m) < (1ll << 31)) {
        int* flag = fast_flag;
        const char* q0 = (const char*)packed + packed_fused_bytes(C, heads);
        const size_t cc = conv_packed_bytes(C, C);
        const char* pk4 = q0 + 4 * cc;
        const char* pk5 = pk4 + conv_packed_bytes(2 * C, 2 * C);
        if ((rc = launch_conv_pk(q0, x, nullptr, C, 0, C, n, batch * n, nullptr, nullptr, w->bq, nullptr, q, flag, nullptr, st, 0))) return rc;
        if ((rc = launch_conv_pk(q0 + cc, source, nullptr, C, 0, C, m, batch * m, nullptr, nullptr, w->bk, nullptr, k, flag, nullptr, st, 0))) return rc;
        if ((rc = launch_conv_pk(q0 + 2 * cc, source, nullptr, C, 0, C, m, batch * m, nullptr, nullptr, w->bv, nullptr, v, flag, nullptr, st, 0))) return rc;
        // the attention core: at the fine level's shape with its scores in registers (attention145.hip), else the general kernel
        rc = launch_attention145(q, k, v, batch, C / heads, heads, n, m, att, flag, nullptr, st);
        if (rc == PATS_ERR_UNSUPPORTED) rc = launch_attention(q, k, v, batch, C / heads, heads, n, m, att, nullptr, stream, nullptr);
        if (rc) return rc;
        if (gnn_fold_enabled()) {
            // the merge is folded into mlp[0]'s packed weights and bias (gnn_fold_kernel): hidden = W1x x + (W1m Wm) att + b1'
            if ((rc = launch_conv_pk(pk4, x, att, C, C, 2 * C, n, batch * n, nullptr, nullptr, packed_folded_bias(packed, C, heads), nullptr, hid, flag, nullptr, st, 4))) return rc;
        } else {
            if ((rc = launch_conv_pk(q0 + 3 * cc, att, nullptr, C, 0, C, n, batch * n, nullptr, nullptr, w->bm, nullptr, msg, flag, nullptr, st, 4))) return rc;
            if ((rc = launch_conv_pk(pk4, x, msg, C, C, 2 * C, n, batch * n, nullptr, nullptr, w->b1, nullptr, hid, flag, nullptr, st, 2 | 4))) return rc;
        }
        const float *sc = w->bn_a, *sh = w->bn_b;
        if (bn_train) {
            hipLaunchKernelGGL(bn_partial_blocked_kernel, dim3((unsigned)(2 * C / 8), BN_SPLITS), dim3(256), 0, st, hid, batch, 2 * C, n, bpart,
                               (const int*)nullptr);
            hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)((2 * C + 63) / 64)), dim3(64), 0, st, bpart, batch * (int64_t)n, 2 * C,
                               w->bn_a, w->bn_b, bn_eps, bsc, bsh, (const int*)nullptr);
            if ((rc = check_launch("bn_stats kernels"))) return rc;
            sc = bsc; sh = bsh;
        }
        if ((rc = launch_conv_pk(pk5, hid, nullptr, 2 * C, 0, C, n, batch * n, sc, sh, w->b2, residual, out, flag, nullptr, st, 1))) return rc;
        if (deferred) return PATS_OK;
        gate = flag;             // the composition below runs only if one of the kernels above raised it
    }
    if (deferred && fill_bytes(redo, 0, 8 * sizeof(int), st)) return PATS_ERR_LAUNCH;     // no fast branch took the call: the composition's flags
    // (as the gated fallback behind the fused layer / the packed-weights kernels: conv1x1_kernel alone - it has the fp32 redo inside -
    //  i.e. nine empty launches per layer instead of fifteen)
    // projections (modules.py:101-102)
    if ((rc = launch_conv(ConvArgs{w->wq_t, x, nullptr, C, 0, C, n, batch * n, nullptr, nullptr, w->bq, nullptr, q}, gate ? nullptr : redo + 0, st, gate))) return rc;
    if ((rc = launch_conv(ConvArgs{w->wk_t, source, nullptr, C, 0, C, m, batch * m, nullptr, nullptr, w->bk, nullptr, k}, gate ? nullptr : redo + 1, st, gate))) return rc;
    if ((rc = launch_conv(ConvArgs{w->wv_t, source, nullptr, C, 0, C, m, batch * m, nullptr, nullptr, w->bv, nullptr, v}, gate ? nullptr : redo + 2, st, gate))) return rc;
    // attention core (:103): the [b, C, n] projections ARE the [b, dim, heads, n] views
    if ((rc = launch_attention(q, k, v, batch, C / heads, heads, n, m, att, nullptr, stream, gate))) return rc;
    // merge (:104)
    if ((rc = launch_conv(ConvArgs{w->wm_t, att, nullptr, C, 0, C, n, batch * n, nullptr, nullptr, w->bm, nullptr, msg}, gate ? nullptr : redo + 3, st, gate))) return rc;
    // mlp[0] on cat([x, message]) without the cat (:116, MLP :64)
    if ((rc = launch_conv(ConvArgs{w->w1_t, x, msg, C, C, 2 * C, n, batch * n, nullptr, nullptr, w->b1, nullptr, hid}, gate ? nullptr : redo + 4, st, gate))) return rc;
    // mlp[1] BatchNorm1d: eval -> the caller's folded running statistics (bn_a = scale, bn_b = shift);
    //                      train -> batch statistics with bn_a = gamma, bn_b = beta
    const float *sc = w->bn_a, *sh = w->bn_b;
    if (bn_train) {
        hipLaunchKernelGGL(bn_partial_kernel, dim3((unsigned)(2 * C), BN_SPLITS), dim3(256), 0, st, hid, batch, 2 * C, n, bpart, gate);
        hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)((2 * C + 63) / 64)), dim3(64), 0, st, bpart, batch * (int64_t)n, 2 * C,
                           w->bn_a, w->bn_b, bn_eps, bsc, bsh, gate);
        if ((rc = check_launch("bn_stats kernels"))) return rc;
        sc = bsc; sh = bsh;
    }
    // mlp[2] ReLU + mlp[3] Conv1d(2C, C), BN affine + ReLU applied while staging; optional residual (desc + delta, :133)
    return launch_conv(ConvArgs{w->w2_t, hid, nullptr, 2 * C, 0, C, n, batch * n, sc, sh, w->b2, residual, out}, gate ? nullptr : redo + 5, st, gate);
}

// ---- AttentionalGNN.forward (modules.py:127-134) at the fine level's shape, descriptors kept in the one-kernel layer's own form ----
// (round 5) desc0, desc1 [batch, 264, 145] -> one array of P = 2 batch problems as (fp32 channel-blocked, TF image), `layers` launches
// of gnn_fine_layer_kernel (a layer updates BOTH descriptor sets: problem p < batch is desc0[p], p >= batch desc1[p - batch]; a
// 'cross' layer reads the source image (p + batch) % P, a 'self' layer its own), back to [batch, 264, 145].  Between the layers
// nothing but the two forms travels: no conversion, no q / k / v / message / hidden tensor in HBM.  Eval-mode BatchNorm only (the
// scale / shift given at pack time).  If any layer meets a non-finite value (an activation beyond the fp16 range of the split
// operands) a device flag is raised and the chain of per-layer compositions queued behind - every kernel gated on that flag -
// recomputes the whole stack from the inputs.
namespace pats {
// (rows past the device-side count are NOT copied: the redo chain recomputes them from padding inputs, and the fast path's contract -
//  gnn_fine_out_kernel leaves zeros there - must survive a redo; round-5 advice)
__global__ void __launch_bounds__(256) gated_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4, const int* __restrict__ gate,
                                                         int64_t row4, const int64_t* __restrict__ live, int64_t live_off) {
    if (*gate == 0) return;
    int64_t lim = n4;
    if (live) {
        int64_t rows = *live - live_off;
        rows = rows < 0 ? 0 : rows;
        lim = rows * row4 < n4 ? rows * row4 : n4;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < lim; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
}
extern "C" size_t pats_attentional_gnn_packed_workspace_bytes(int64_t batch, int C, int heads, int n) {
    if (batch <= 0 || !fine_layer_supported(C, heads, n, n)) return 0;
    const int64_t P = 2 * batch;
    const size_t fused = 3 * al256(fine_image_bytes(P));
    const size_t redo = 4 * al256((size_t)batch * C * n * sizeof(float)) + al256(pats_attentional_propagation_workspace_bytes(batch, C, n, n));
    return std::max(fused, redo) + al256(fine_scratch_bytes(P)) + 256;
}
extern "C" int pats_attentional_gnn_packed_f32(const float* desc0, const float* desc1, int64_t batch, const int64_t* live, int64_t live_off,
                                               int C, int heads, int n, int layers,
                                               const pats_propagation_weights* const* weights, const void* const* packed,
                                               const int* cross, float bn_eps, float* out0, float* out1, void* workspace,
                                               size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && layers >= 0, "attentional_gnn_packed: bad shape");
    if (!fine_layer_supported(C, heads, n, n) || !gnn_fold_enabled()) return PATS_ERR_UNSUPPORTED;
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(desc0 && desc1 && out0 && out1 && (layers == 0 || (weights && packed && cross)), "attentional_gnn_packed: null pointer");
    // (round-5 advice) the inline redo chain re-reads the inputs AFTER the fast path has written the outputs
    PATS_REQUIRE(out0 != desc0 && out0 != desc1 && out1 != desc0 && out1 != desc1, "attentional_gnn_packed: the outputs must not alias the inputs");
    PATS_REQUIRE(workspace && workspace_bytes >= pats_attentional_gnn_packed_workspace_bytes(batch, C, heads, n), "attentional_gnn_packed: workspace too small");
    for (int l = 0; l < layers; ++l) PATS_REQUIRE(weights[l] && packed[l], "attentional_gnn_packed: null layer");
    hipStream_t st = as_stream(stream);
    const int64_t P = 2 * batch;
    const size_t elems = (size_t)batch * C * n;
    char* p = (char*)workspace;
    const size_t fused = 3 * al256(fine_image_bytes(P));
    const size_t redo_b = 4 * al256(elems * sizeof(float)) + al256(pats_attentional_propagation_workspace_bytes(batch, C, n, n));
    char* tf[2] = {p, p + al256(fine_image_bytes(P))};
    char* tf_att = p + 2 * al256(fine_image_bytes(P));
    char* scratch = p + std::max(fused, redo_b);
    const bool deferred = gnn_redo_deferred();          // no redo chain: the device's sticky flag, read by the caller (host.cpp)
    int* flag = deferred ? gnn_overflow_flag() : (int*)(scratch + al256(fine_scratch_bytes(P)));
    if (!deferred && fill_bytes(flag, 0, sizeof(int), st)) return PATS_ERR_LAUNCH;
    int rc;
    // in: the two descriptor sets into one array of problems
    if ((rc = launch_fine_in(desc0, batch, tf[0], st))) return rc;
    if ((rc = launch_fine_in(desc1, batch, tf[0] + fine_image_bytes(batch), st))) return rc;
    int cur = 0;
    // the first layer's projections; every later layer's are made by the MLP launch of the layer before it, from the tile in its LDS
    if (layers > 0 && (rc = launch_fine_qkv(tf[0], P, packed_fine_section(packed[0], C, heads), scratch, 1, 1, nullptr, st, 2, live, live_off)))
        return rc;                   // (PATS_ERR_UNSUPPORTED: the LDS attribute was refused - the caller takes the per-layer path)
    for (int l = 0; l < layers; ++l) {
        if ((rc = launch_fine_attn(scratch, cross[l] ? batch : 0, tf_att, P, nullptr, st, 2, live, live_off))) return rc;
        rc = launch_fine_mlp(tf[cur], tf_att, 1, packed_fine_section(packed[l], C, heads), tf[1 - cur],
                             l + 1 < layers ? packed_fine_section(packed[l + 1], C, heads) : nullptr, scratch, P, flag, nullptr, st, 2, live, live_off);
        if (rc) return rc;
        cur = 1 - cur;
    }
    if ((rc = launch_fine_out(tf[cur], batch, out0, st, live, live_off))) return rc;
    if ((rc = launch_fine_out(tf[cur] + fine_image_bytes(batch), batch, out1, st, live, live_off))) return rc;
    if (deferred) return check_launch("attentional_gnn_packed");
    // the redo chain (no-ops unless the flag is up): the layers one by one on the round-2 composition, from the inputs
    float* d[2][2] = {{(float*)p, (float*)(p + al256(elems * sizeof(float)))},
                      {(float*)(p + 2 * al256(elems * sizeof(float))), (float*)(p + 3 * al256(elems * sizeof(float)))}};
    void* cws = p + 4 * al256(elems * sizeof(float));
    const size_t cws_b = pats_attentional_propagation_workspace_bytes(batch, C, n, n);
    const float *c0 = desc0, *c1 = desc1;
    for (int l = 0; l < layers; ++l) {
        float *n0 = d[l & 1][0], *n1 = d[l & 1][1];
        const float *s0 = cross[l] ? c1 : c0, *s1 = cross[l] ? c0 : c1;
        if ((rc = propagation_impl(c0, s0, batch, C, heads, n, n, weights[l], 0, bn_eps, c0, n0, cws, cws_b, stream, nullptr, flag))) return rc;
        if ((rc = propagation_impl(c1, s1, batch, C, heads, n, n, weights[l], 0, bn_eps, c1, n1, cws, cws_b, stream, nullptr, flag))) return rc;
        c0 = n0; c1 = n1;
    }
    const int64_t n4 = (int64_t)(elems / 4);
    const unsigned cg = (unsigned)std::min<int64_t>((n4 + 255) / 256, 4096);
    const int64_t row4 = (int64_t)C * n / 4;            // (C n is a multiple of four at the one supported shape: 264 x 145)
    hipLaunchKernelGGL(gated_copy_kernel, dim3(cg), dim3(256), 0, st, (const float4*)c0, (float4*)out0, n4, (const int*)flag, row4, live, live_off);
    hipLaunchKernelGGL(gated_copy_kernel, dim3(cg), dim3(256), 0, st, (const float4*)c1, (float4*)out1, n4, (const int*)flag, row4, live, live_off);
    return check_launch("gated_copy_kernel");
}

// ---- the building blocks on their own: Conv1d(kernel_size = 1) and the BatchNorm1d + ReLU that follows it in MLP ------
// (models/modules.py:57-69; KeypointEncoder :70-82; final_proj first_layer.py:34-36,105 / second_layer.py:40-42,91)
extern "C" size_t pats_conv1x1_workspace_bytes(void) { return 256; }

extern "C" int pats_conv1x1_f32(const float* w_t, const float* bias, const float* x, int64_t batch, int K, int M, int n,
                                const float* in_scale, const float* in_shift, const float* residual, float* y,
                                void* workspace, size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && K > 0 && M > 0 && n > 0, "conv1x1: bad shape");
    PATS_REQUIRE((K % 8) == 0, "conv1x1: input channels must be a multiple of 8 (operand slabs of 8 channels; pad with zero channels)");
    PATS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv1x1: in_scale and in_shift come together");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(w_t && x && y, "conv1x1: null pointer");
    PATS_REQUIRE(workspace && workspace_bytes >= pats_conv1x1_workspace_bytes(), "conv1x1: workspace too small");
    hipStream_t st = as_stream(stream);
    int* redo = (int*)workspace;
    if (fill_bytes(redo, 0, sizeof(int), st)) return PATS_ERR_LAUNCH;
    return launch_conv(ConvArgs{w_t, x, nullptr, K, 0, M, n, batch * n, in_scale, in_shift, bias, residual, y}, redo, st);
}

extern "C" size_t pats_bn_fold_workspace_bytes(int C) {
    return C > 0 ? al256((size_t)C * pats::BN_SPLITS * 2 * sizeof(double)) : 0;
}

extern "C" int pats_bn_fold_f32(const float* h, int64_t batch, int C, int n, const float* gamma, const float* beta,
                                float eps, float* scale, float* shift, void* workspace, size_t workspace_bytes,
                                pats_stream_t stream) {
    PATS_REQUIRE(batch > 0 && C > 0 && n > 0, "bn_fold: bad shape (batch statistics need at least one column)");
    PATS_REQUIRE(h && gamma && beta && scale && shift, "bn_fold: null pointer");
    PATS_REQUIRE(workspace && workspace_bytes >= pats_bn_fold_workspace_bytes(C), "bn_fold: workspace too small");
    hipStream_t st = as_stream(stream);
    double* part = (double*)workspace;
    hipLaunchKernelGGL(bn_partial_kernel, dim3((unsigned)C, BN_SPLITS), dim3(256), 0, st, h, batch, C, n, part);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, st, part, batch * (int64_t)n, C, gamma, beta,
                       eps, scale, shift);
    return check_launch("bn_fold kernels");
}
