// AttentionalPropagation (models/modules.py:107-117) at the third level's shape - x, source [b,128,65], 4 heads - as ONE kernel:
// a problem's activations never leave the CU between the two descriptor blocks coming in and the layer's output going out.
//
//   message = merge(attention(proj_q(x), proj_k(source), proj_v(source)))           MultiHeadedAttention.forward :100-105
//   hidden  = Conv1d(256,256)(cat([x, message]))                                    AttentionalPropagation.forward :114-117, MLP :57-69
//   out     = [residual +] Conv1d(256,128)(relu(bn(hidden)))                        (AttentionalGNN.forward :131-133)
//
// The composition of csrc/gnn.hip runs this as seven launches with every intermediate tensor going through HBM (about 800 KB
// of traffic per 65-token problem for 100 KB of input + output: the layer was HBM-bound in aggregate, 5.7 ms per 25 920
// problems).  Here one 512-thread workgroup owns a problem at a time (persistent grid, one workgroup per CU):
//
//   * activations live in LDS pre-split for the fp16 matrix pipe: x * 2^6 = hi + lo (both fp16, round to nearest: 22 mantissa
//     bits), in MFMA FRAGMENT ORDER ("TF layout": per 32-channel k-step and 16-token tile, lane (k / 8, token) holds its 8
//     consecutive channels as one 16-byte read - directly the B operand of v_mfma_f32_16x16x32_f16, and equally the A operand
//     of the transposed product).  A [128 x 65] tensor takes 33 280 bytes: four k-steps x (hi, lo) x (four full tiles + token
//     64 alone; the lanes of the fifth tile all read token 64, their columns are never stored).
//   * every Conv1d is the three-pass contraction of the cost build (lo.hi + hi.lo + hi.hi, fp32 accumulation) with the WEIGHTS
//     as the A operand, pre-split and pre-tiled once per layer by gnn_pack_kernel (640 KB, L2-resident, streamed by every
//     workgroup: one 16-byte load per lane and fragment).  Wave w owns output rows 16 w.. (mlp[0]: two row tiles) and walks
//     the five token tiles; its epilogue (bias, BatchNorm affine + ReLU, re-split) writes the next stage's operand in place.
//   * heads: the reference views the projections as [b, dim, heads, n] (channel = d * 4 + h).  The packed q / k / v weights
//     have their output rows permuted to head-major (h * 32 + d), the merge its input columns: a head is then exactly one
//     k-step of the TF layout.
//   * attention per (head, 16-query tile): S^T = K^T Q (keys as rows: the softmax over the keys is in-lane + two lane
//     exchanges), exp, and the accumulator registers of two key tiles ARE the B operand of out = V P^T (key slot 8 q + e of a
//     k-step = key 4 q + e of the even tile, e < 4, or of the odd tile) - the A operand V comes from a token-major copy
//     ("TT layout") that the v projection produces directly by running transposed (activations as A, weights as B).
//   * BatchNorm in eval mode (folded running statistics: the outdoor configuration, pats.py:112-116) -> the whole layer in this
//     kernel.  On batch statistics (indoor: pats.py:117-118 leaves the third layer in train mode) the kernel stops behind
//     mlp[0] and writes the hidden tensor; the statistics kernels and the last convolution of gnn.hip finish the layer.
//
// LDS: x | source -> q -> attention output -> hidden[0:128] | k -> message | v^T -> hidden[128:256] = 134 656 bytes.
// Range: |activation| < 1023 (the hi half overflows beyond, as in the cost build).  A non-finite output raises *flag and the
// host entry then runs the composition of gnn.hip behind it, gated on that flag (its kernels return at once otherwise).
#include "common.hpp"

#include <cstdlib>

namespace pats {

namespace {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int GC = 128, GN = 65;
constexpr float PRE = 64.0f, UNS = 1.0f / 4096.0f;
constexpr int TF_BLK = 4 * 1024 + 64;             // bytes of one (k-step, hi | lo) block
constexpr int TF_BYTES = 4 * 2 * TF_BLK;          // [128 x 65] tensor: 33 280
constexpr int TT_ROW = 68;                        // tokens per channel row of the token-major copy (65 + zero padding)
constexpr int TT_PLANE = GC * TT_ROW * 2;         // bytes of its hi (or lo) plane
constexpr int OFF_X = 0, OFF_S = TF_BYTES, OFF_K = 2 * TF_BYTES, OFF_V = 3 * TF_BYTES;
constexpr int FUSED_MAX_GRID = 512;                           // workgroups of the persistent grid = partial-sum slots per channel

// packed weights (units of h8v = 16 bytes): fragment (row tile mt, k-step ks) = [hi | lo][64 lanes]
constexpr int FR = 2 * 64;
constexpr int PW_Q = 0, PW_K = PW_Q + 8 * 4 * FR, PW_V = PW_K + 8 * 4 * FR, PW_M = PW_V + 8 * 4 * FR, PW_1 = PW_M + 8 * 4 * FR,
              PW_2 = PW_1 + 16 * 8 * FR, PW_END = PW_2 + 8 * 8 * FR;
// biases behind them (floats): bq', bk', bv' (head-major), bm, b1 [256], b2
constexpr int PB_Q = 0, PB_K = 128, PB_V = 256, PB_M = 384, PB_1 = 512, PB_2 = 768, PB_END = 896;
constexpr int FUSED_LDS = 3 * TF_BYTES + 2 * TT_PLANE;        // 134 656

struct FusedArgs {
    const float* x;
    const float* source;
    const float* residual;     // or null
    float* out;                // [b,128,65]            (eval)
    float* hid;                // [b,256,65] pre-BN     (train)
    const h8v* pw;
    const float* pb;
    const float* bn_a;         // eval: folded scale / shift [256]
    const float* bn_b;
    int64_t batch;
    int* flag;
    double* bn_part;           // train: per-workgroup partial sums of the hidden tensor, [256][gridDim.x][2] (sum, sum of squares)
    const int64_t* live;       // optional device-side problem count (eval mode): problems >= clamp(*live - live_off, 0, batch) are skipped
    int64_t live_off;
};

__device__ __forceinline__ int tf_tile_off(int nt, int lane) {      // byte offset of this lane's 16-byte fragment piece in a block
    return nt < 4 ? nt * 1024 + lane * 16 : 4096 + (lane >> 4) * 16;
}

__device__ __forceinline__ void split4(const f4v v, h4v& hi, h4v& lo) {
    const f4v s = v * PRE;
    hi = __builtin_convertvector(s, h4v);
    lo = __builtin_convertvector(s - __builtin_convertvector(hi, f4v), h4v);
}

// the same for a value that already carries the factor PRE (the epilogues fold it - a power of two - into their one fma)
__device__ __forceinline__ void split4_pre(const f4v s, h4v& hi, h4v& lo) {
    hi = __builtin_convertvector(s, h4v);
    lo = __builtin_convertvector(s - __builtin_convertvector(hi, f4v), h4v);
}
__device__ __forceinline__ f4v fma4(const f4v a, const f4v b, const f4v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f4v bcast4(float x) { return f4v{x, x, x, x}; }

// rows 16 mt + 4 q' + r (r = 0..3) of token 16 nt + j -> the TF tensor at dst.  PRESCALED: v = PRE x the value
template <bool PRESCALED = false>
__device__ __forceinline__ void store_tf(char* dst, int mt, int nt, const f4v v, int lane) {
    const int qp = lane >> 4, j = lane & 15;
    if (nt == 4 && j != 0) return;
    h4v hi, lo;
    if (PRESCALED) split4_pre(v, hi, lo);
    else split4(v, hi, lo);
    const int kq = 2 * (mt & 1) + (qp >> 1);
    const int off = ((mt >> 1) * 2) * TF_BLK + (nt < 4 ? nt * 1024 + (kq * 16 + j) * 16 : 4096 + kq * 16) + (qp & 1) * 8;
    *reinterpret_cast<h4v*>(dst + off) = hi;
    *reinterpret_cast<h4v*>(dst + off + TF_BLK) = lo;
}

// [128][65] fp32 in global memory -> registers -> TF tensor.  Item = (channel group of 8, token): tokens 0..63 are 16 x 64 items
// = two per thread (consecutive lanes = consecutive tokens: 256-byte runs per load instruction), token 64 one more item for
// threads 0..15.  The loads of problem p + 1 are issued right after problem p has been written to LDS and land under its compute.
struct InRegs { f4v a[3], b[3]; };

__device__ __forceinline__ void load_in(const float* __restrict__ src, int t, InRegs& q) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const bool live = it < 2 || t < 16;
        const int cg = it < 2 ? (t + 512 * it) >> 6 : (t & 15), tok = it < 2 ? (t & 63) : 64;
        const float* p = src + (cg * 8) * GN + tok;
        if (live) {
            q.a[it] = f4v{p[0], p[GN], p[2 * GN], p[3 * GN]};
            q.b[it] = f4v{p[4 * GN], p[5 * GN], p[6 * GN], p[7 * GN]};
        }
    }
}

__device__ __forceinline__ void store_in(const InRegs& q, char* dst, int t) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        if (it < 2 || t < 16) {
            const int cg = it < 2 ? (t + 512 * it) >> 6 : (t & 15), tok = it < 2 ? (t & 63) : 64;
            h4v ah, al, bh, bl;
            split4(q.a[it], ah, al);
            split4(q.b[it], bh, bl);
            const int tile = tok >> 4, j = tok & 15, kq = cg & 3;
            const int off = ((cg >> 2) * 2) * TF_BLK + (tile < 4 ? tile * 1024 + (kq * 16 + j) * 16 : 4096 + kq * 16);
            *reinterpret_cast<h8v*>(dst + off) = h8v{ah.x, ah.y, ah.z, ah.w, bh.x, bh.y, bh.z, bh.w};
            *reinterpret_cast<h8v*>(dst + off + TF_BLK) = h8v{al.x, al.y, al.z, al.w, bl.x, bl.y, bl.z, bl.w};
        }
    }
}

__device__ __forceinline__ f4v mfma3(const h8v ah, const h8v al, const h8v bh, const h8v bl, f4v c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);       // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    return c;
}

// Weight fragments of a stage: a ring of four k-steps per row tile (hi, lo), loaded AHEAD - the first four k-steps of stage
// s + 1 are issued before the epilogue and the barrier of stage s, so that their L2 round trip (the weights are streamed by
// every workgroup, 640 KB per problem) runs under them; the counters of the first version showed the waves waiting 73 % of
// their cycles with the matrix pipe 21 % busy: the one-k-step-ahead fetch of 240 MFMA cycles did not cover the latency.
// (depth 4 for one row tile per wave, 2 for two: a k-step of two row tiles is 30 MFMAs per wave, twice the cover)
// A fragment's address is wave-uniform + lane * 16: the uniform part is forced into SGPRs (readfirstlane), so that a load is
// `global_load_dwordx4 v, v_lane, s[base] offset:0 | 1024`.  Left to itself the compiler kept one 64-bit VECTOR address per
// load, spilled them, and every reload put an s_waitcnt vmcnt(0) in front of the next weight load - no prefetch at all.
typedef const __attribute__((address_space(1))) h8v* gptr_h8;        // explicitly GLOBAL: the integer round trip below would otherwise
                                                                     // leave a generic pointer and flat_load (which also counts on lgkmcnt)
__device__ __forceinline__ gptr_h8 uniform_ptr(const h8v* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr_h8)(((uint64_t)hi << 32) | lo);
}
template <int MT>
struct WRing {
    static constexpr int D = MT == 1 ? 4 : 2;
    h8v a[D][MT][2];
};

template <int KS, int MT>
__device__ __forceinline__ void ring_fill(const h8v* __restrict__ W, const int (&mts)[MT], int lane, WRing<MT>& r) {
#pragma unroll
    for (int ks = 0; ks < WRing<MT>::D; ++ks)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            gptr_h8 Wf = uniform_ptr(W + (mts[m] * KS + ks) * FR);
            r.a[ks][m][0] = Wf[lane];
            r.a[ks][m][1] = Wf[64 + lane];
        }
}

// acc[m][nt] += W[row tile mts[m]] . src over KS k-steps (k-steps 0..3 from src0, 4..7 from src1); weights = A operand,
// the ring holds k-steps 0..D-1 on entry
template <int KS, int MT, bool PIPE = (MT == 1)>
__device__ __forceinline__ void gemm_w(const h8v* __restrict__ W, const int (&mts)[MT], const char* src0, const char* src1,
                                       int lane, f4v (&acc)[MT][5], WRing<MT>& r) {
    constexpr int D = WRing<MT>::D;
    // PIPE: the activations' fragments (B operand) of k-step ks + 1 are read from LDS while the MFMAs of k-step ks run - two
    // register sets.  (With the reads and their MFMAs in one scheduling unit every k-step begins with an LDS round trip, two waves
    // per SIMD: the one-row-tile stages ran 5-15 % longer, profiles/r04_gnn_fused_timeline.txt.)  Off where the registers are
    // needed: two row tiles per wave - 30 MFMAs per k-step cover the round trip better anyway - and the last stage, which holds
    // the next problem's inputs.
    constexpr int NB = PIPE ? 2 : 1;
    h8v bq[NB][5][2];
    auto bload = [&](int ks, int buf) {
        const char* blk = (ks < 4 ? src0 : src1) + ((ks & 3) * 2) * TF_BLK;
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) {
            bq[buf][nt][0] = *reinterpret_cast<const h8v*>(blk + tf_tile_off(nt, lane));
            bq[buf][nt][1] = *reinterpret_cast<const h8v*>(blk + TF_BLK + tf_tile_off(nt, lane));
        }
    };
    if (NB == 2) bload(0, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (NB == 2) { if (ks + 1 < KS) bload(ks + 1, (ks + 1) & 1); }
        else bload(ks, 0);
        h8v ah[MT], al[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) { ah[m] = r.a[ks % D][m][0]; al[m] = r.a[ks % D][m][1]; }
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][nt] = mfma3(ah[m], al[m], bq[ks & (NB - 1)][nt][0], bq[ks & (NB - 1)][nt][1], acc[m][nt]);
        }
        if (ks + D < KS) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                gptr_h8 Wf = uniform_ptr(W + (mts[m] * KS + ks + D) * FR);
                r.a[ks % D][m][0] = Wf[lane];
                r.a[ks % D][m][1] = Wf[64 + lane];
            }
        }
        // a k-step is a closed unit for the scheduler: left alone it hoists the LDS reads of many k-steps ahead of their MFMAs
        // and spills (222 VGPRs in the first version of this loop)
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MT>
__device__ __forceinline__ void zero_acc(f4v (&acc)[MT][5]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) acc[m][nt] = f4v{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ f4v load4(const float* p) { return f4v{p[0], p[1], p[2], p[3]}; }

}  // namespace

// ---- merge folded into mlp[0] at pack time (round 4) ----------------------------------------------------------------------------
// modules.py:104,116: message = Wm att + bm is consumed by ONE product, hidden = W1 (x | message) + b1.  With W1 = (W1x | W1m):
//     hidden = W1x x + (W1m Wm) att + (W1m bm + b1)
// so the layer needs no merge product at all once W1m Wm [2C x C] and the bias are formed - once per layer, here, in double, rounded
// once to fp32 (the folded matrix then goes through the same split as every weight).  One Conv1d, its launch and two tensor passes
// less per layer (the fused kernel: one stage and one barrier less); the result differs from the two-step form by fp32 rounding
// only (1e-7 relative per term: the goldens' tolerance is 2e-4).  PATS_GNN_FOLD=0 (read once per process) keeps the two products.
// w1f_t: [2C in][2C out] like w1_t - rows 0..C-1 = W1x, rows C.. = the folded matrix by attention channel; b1f [2C].
__global__ void __launch_bounds__(256)
gnn_fold_kernel(pats_propagation_weights w, int C, float* __restrict__ w1f_t, float* __restrict__ b1f) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, M = 2 * C;
    if (gid < M) {                                             // b1f[o] = b1[o] + sum_c W1[o][C + c] bm[c]
        double s = 0.0;
        for (int c = 0; c < C; ++c) s += (double)w.w1_t[(int64_t)(C + c) * M + gid] * (double)w.bm[c];
        b1f[gid] = (float)((double)w.b1[gid] + s);
    }
    if (gid >= M * M) return;
    const int k = (int)(gid / M), o = (int)(gid - (int64_t)k * M);
    if (k < C) { w1f_t[gid] = w.w1_t[gid]; return; }
    const int i = k - C;                                       // attention channel: sum_c Wm[c][i] W1[o][C + c]
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += (double)w.wm_t[(int64_t)i * C + c] * (double)w.w1_t[(int64_t)(C + c) * M + o];
    w1f_t[gid] = (float)s;
}

// weights of one layer -> fragment order, split, head permutation.  One thread per (fragment, lane).
// fold: w.w1_t / w.b1 are the folded ones and mlp[0]'s second operand is the ATTENTION slot, whose channels are head-major.
__global__ void __launch_bounds__(256)
gnn_pack_kernel(pats_propagation_weights w, h8v* __restrict__ pw, float* __restrict__ pb, int fold) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid < PB_END) {                       // biases (q / k / v in head-major order)
        const int i = gid & 127, pi = (i & 31) * 4 + (i >> 5);
        float v;
        if (gid < PB_K) v = w.bq[pi];
        else if (gid < PB_V) v = w.bk[pi];
        else if (gid < PB_M) v = w.bv[pi];
        else if (gid < PB_1) v = w.bm[i];
        else if (gid < PB_2) v = w.b1[gid - PB_1];
        else v = w.b2[i];
        pb[gid] = v;
    }
    const int lane = gid & 63, f = gid >> 6;                   // fragment index over all matrices
    if (f >= PW_END / FR) return;
    const float* wt;
    int K, M, mt, ks, mode;                                    // mode 1: permute output rows, 2: permute input channels
    int base;
    if (f < 4 * 32) { const int mat = f >> 5; wt = mat == 0 ? w.wq_t : mat == 1 ? w.wk_t : mat == 2 ? w.wv_t : w.wm_t;
                      K = 128; M = 128; mt = (f & 31) >> 2; ks = f & 3; mode = mat == 3 ? 2 : 1; base = mat * 32 * FR; }
    else if (f < 4 * 32 + 128) { const int g = f - 128; wt = w.w1_t; K = 256; M = 256; mt = g >> 3; ks = g & 7; mode = fold ? 3 : 0; base = PW_1; }
    else { const int g = f - 256; wt = w.w2_t; K = 256; M = 128; mt = g >> 3; ks = g & 7; mode = 0; base = PW_2; }
    (void)K;
    int row = mt * 16 + (lane & 15);
    if (mode == 1) row = (row & 31) * 4 + (row >> 5);
    h8v hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int k = ks * 32 + 8 * (lane >> 4) + e;
        if (mode == 2) k = (k & 31) * 4 + (k >> 5);
        if (mode == 3 && k >= 128) k = 128 + ((k - 128) & 31) * 4 + ((k - 128) >> 5);
        const float s = wt[(int64_t)k * M + row] * PRE;
        const _Float16 h = (_Float16)s;
        hi[e] = h;
        lo[e] = (_Float16)(s - (float)h);
    }
    const int KSm = f < 128 ? 4 : 8, fl = f < 128 ? (f & 31) : (f < 256 ? f - 128 : f - 256);
    (void)KSm;
    pw[base + fl * FR + lane] = hi;
    pw[base + fl * FR + 64 + lane] = lo;
}

template <bool TRAIN, bool FOLD>
__global__ void __launch_bounds__(512, 1)
gnn_layer_fused_kernel(FusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int qp = lane >> 4, j = lane & 15;
    const float* pb = g.pb;
    const int mt1[1] = {wave};
    const int mt2[2] = {wave, wave + 8};
    bool bad = false;
    // TRAIN: this lane's share of the batch statistics of the hidden tensor (mlp[1] = BatchNorm1d on batch statistics): per
    // problem the fp32 sums over a channel's 65 tokens (fixed order: in-lane over the token tiles, then a 16-lane butterfly), across
    // the workgroup's problems in double.  Every workgroup writes its partials, bn_finish_kernel adds them in index order:
    // the result does not depend on scheduling.
    double st1[2][4], st2[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st1[m][r] = 0.0; st2[m][r] = 0.0; }
    int64_t b = blockIdx.x;
    if (g.live) { const int64_t l_ = *g.live - g.live_off; g.batch = l_ < 0 ? 0 : (l_ < g.batch ? l_ : g.batch); }
    InRegs xin, sin;
    if (b < g.batch) {
        load_in(g.x + b * (GC * GN), t, xin);
        load_in(g.source + b * (GC * GN), t, sin);
    }
    WRing<1> rk;
    ring_fill<4, 1>(g.pw + PW_K, mt1, lane, rk);
    for (; b < g.batch; b += gridDim.x) {
        store_in(xin, lds + OFF_X, t);           // loaded under the previous problem's last stage (48 registers: holding them
        store_in(sin, lds + OFF_S, t);           // through the whole problem made the mlp[0] stage spill)
        wg_barrier();
        // ---- k = Wk' source (TF), v^T = source^T Wv'^T (token-major) -----------------------------------------------
        WRing<1> rv, rq, rm;
        ring_fill<4, 1>(g.pw + PW_V, mt1, lane, rv);
        {
            // ONE loop for both products: a fragment of source read from LDS is the B operand of k's row tile (weights x source) and
            // the A operand of v's channel tile (source^T x weights^T) - the two layouts coincide - so v costs no LDS read of its own
            // (k, v + barrier 4.55 -> 3.7 us per problem)
            f4v acck[5], acc[5];
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) { acck[tt] = f4v{0.f, 0.f, 0.f, 0.f}; acc[tt] = f4v{0.f, 0.f, 0.f, 0.f}; }
            h8v bq[2][5][2];
            auto bload = [&](int ks, int buf) {
                const char* blk = lds + OFF_S + (ks * 2) * TF_BLK;
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) {
                    bq[buf][nt][0] = *reinterpret_cast<const h8v*>(blk + tf_tile_off(nt, lane));
                    bq[buf][nt][1] = *reinterpret_cast<const h8v*>(blk + TF_BLK + tf_tile_off(nt, lane));
                }
            };
            bload(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) bload(ks + 1, (ks + 1) & 1);
                const h8v kh = rk.a[ks][0][0], kl = rk.a[ks][0][1], wh = rv.a[ks][0][0], wl = rv.a[ks][0][1];
#pragma unroll
                for (int tt = 0; tt < 5; ++tt) {
                    acck[tt] = mfma3(kh, kl, bq[ks & 1][tt][0], bq[ks & 1][tt][1], acck[tt]);
                    acc[tt] = mfma3(bq[ks & 1][tt][0], bq[ks & 1][tt][1], wh, wl, acc[tt]);   // rows = tokens 16 tt + 4 q' + r, column = channel 16 w + j
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                // (acc UNS + bias) PRE in one fma: the same bits as mul, add, mul - the two factors are powers of two
                const f4v bias = load4(pb + PB_K + 16 * wave + 4 * qp) * PRE;
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) store_tf<true>(lds + OFF_K, wave, nt, fma4(acck[nt], bcast4(UNS * PRE), bias), lane);
            }
            ring_fill<4, 1>(g.pw + PW_Q, mt1, lane, rq);
            const float bias = pb[PB_V + 16 * wave + j] * PRE;
            char* vrow = lds + OFF_V + ((16 * wave + j) * TT_ROW) * 2;
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) {
                f4v v = fma4(acc[tt], bcast4(UNS * PRE), bcast4(bias));
                if (tt == 4) { v.y = 0.f; v.z = 0.f; v.w = 0.f; }       // tokens 65..67: zero padding (rows 1.. alias token 64)
                if (tt < 4 || qp == 0) {
                    h4v hi, lo;
                    split4_pre(v, hi, lo);
                    *reinterpret_cast<h4v*>(vrow + (16 * tt + 4 * qp) * 2) = hi;
                    *reinterpret_cast<h4v*>(vrow + TT_PLANE + (16 * tt + 4 * qp) * 2) = lo;
                }
            }
        }
        wg_barrier();
        // ---- q = Wq' x, into the slot source leaves ------------------------------------------------------------------
        {
            f4v acc[1][5];
            zero_acc(acc);
            gemm_w<4, 1>(g.pw + PW_Q, mt1, lds + OFF_X, nullptr, lane, acc, rq);
            const f4v bias = load4(pb + PB_Q + 16 * wave + 4 * qp) * PRE;
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) store_tf<true>(lds + OFF_S, wave, nt, fma4(acc[0][nt], bcast4(UNS * PRE), bias), lane);
        }
        wg_barrier();
        // ---- attention: unit = (head, 16-query tile); its output replaces its own q tile -----------------------------
        for (int u = wave; u < 20; u += 8) {
            const int h = u / 5, qt = u - 5 * h;
            const char* qblk = lds + OFF_S + (h * 2) * TF_BLK;
            const char* kblk = lds + OFF_K + (h * 2) * TF_BLK;
            const h8v qh = *reinterpret_cast<const h8v*>(qblk + tf_tile_off(qt, lane));
            const h8v ql = *reinterpret_cast<const h8v*>(qblk + TF_BLK + tf_tile_off(qt, lane));
            f4v st[5];
            const float sq = 5.656854249492381f, rsq = 1.0f / 5.656854249492381f;        // dim ** .5, dim = 32
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                const h8v kh = *reinterpret_cast<const h8v*>(kblk + tf_tile_off(kt, lane));
                const h8v kl = *reinterpret_cast<const h8v*>(kblk + TF_BLK + tf_tile_off(kt, lane));
                f4v s = mfma3(kh, kl, qh, ql, f4v{0.f, 0.f, 0.f, 0.f});                 // rows = keys 16 kt + 4 q' + r, column = query
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = div_invariant(s[r] * UNS, sq, rsq);
                    if (kt == 4 && (qp != 0 || r != 0)) v = -INFINITY;                  // keys 65..: not there
                    s[r] = v;
                    mx = fmaxf(mx, v);
                }
                st[kt] = s;
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float den = 0.f;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp2((st[kt][r] - mx) * LOG2E);
                    st[kt][r] = p;
                    den += p;
                }
            den += __shfl_xor(den, 16);
            den += __shfl_xor(den, 32);
            const float inv = 1.0f / den;
            // B operands of out = V P^T: k-step kk = key tiles 2 kk (slots e < 4) and 2 kk + 1 (e >= 4)
            h8v ph[3], pl[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                h4v h0, l0, h1 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
                split4(st[2 * kk], h0, l0);
                if (kk < 2) split4(st[2 * kk + 1], h1, l1);
                ph[kk] = h8v{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                pl[kk] = h8v{l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const char* vrow = lds + OFF_V + ((h * 32 + dt * 16 + j) * TT_ROW) * 2;      // A: channel j of the tile, key slots
                f4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    h4v a0h = {0, 0, 0, 0}, a0l = {0, 0, 0, 0}, a1h = {0, 0, 0, 0}, a1l = {0, 0, 0, 0};
                    if (kk < 2 || qp == 0) {
                        a0h = *reinterpret_cast<const h4v*>(vrow + (32 * kk + 4 * qp) * 2);
                        a0l = *reinterpret_cast<const h4v*>(vrow + TT_PLANE + (32 * kk + 4 * qp) * 2);
                    }
                    if (kk < 2) {
                        a1h = *reinterpret_cast<const h4v*>(vrow + (32 * kk + 16 + 4 * qp) * 2);
                        a1l = *reinterpret_cast<const h4v*>(vrow + TT_PLANE + (32 * kk + 16 + 4 * qp) * 2);
                    }
                    const h8v vh = {a0h.x, a0h.y, a0h.z, a0h.w, a1h.x, a1h.y, a1h.z, a1h.w};
                    const h8v vl = {a0l.x, a0l.y, a0l.z, a0l.w, a1l.x, a1l.y, a1l.z, a1l.w};
                    o = mfma3(vh, vl, ph[kk], pl[kk], o);                                  // rows = channels, column = query
                }
                store_tf<true>(lds + OFF_S, 2 * h + dt, qt, o * (UNS * PRE * inv), lane);
            }
        }
        WRing<2> r1;
        if (FOLD) {
            // (the merge is folded into mlp[0]'s weights, gnn_fold_kernel: no message stage)
            ring_fill<8, 2>(g.pw + PW_1, mt2, lane, r1);          // under the barrier wait
            wg_barrier();
        } else {
            ring_fill<4, 1>(g.pw + PW_M, mt1, lane, rm);          // under the barrier wait
            wg_barrier();
            // ---- message = Wm' attention + bm, into the slot k leaves ----------------------------------------------------
            f4v acc[1][5];
            zero_acc(acc);
            gemm_w<4, 1>(g.pw + PW_M, mt1, lds + OFF_S, nullptr, lane, acc, rm);
            ring_fill<8, 2>(g.pw + PW_1, mt2, lane, r1);
            const f4v bias = load4(pb + PB_M + 16 * wave + 4 * qp);
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) store_tf(lds + OFF_K, wave, nt, acc[0][nt] * UNS + bias, lane);
            wg_barrier();
        }
        // ---- hidden = W1 (x | message) + b1: row tiles w and w + 8.  FOLD: (x | attention) with the folded weights; the attention
        //      slot (q's) is still being read while the first waves finish, so hidden[0:128] goes to the slot k left, not to q's ------
        constexpr int OFF_H0 = FOLD ? OFF_K : OFF_S;
        WRing<1> r2;
        f4v res[5];
        {
            f4v acc[2][5];
            zero_acc(acc);
            gemm_w<8, 2>(g.pw + PW_1, mt2, lds + OFF_X, lds + (FOLD ? OFF_S : OFF_K), lane, acc, r1);
            if (!TRAIN) {
                ring_fill<8, 1>(g.pw + PW_2, mt1, lane, r2);
                if (g.residual) {                                  // the residual rows of this lane's outputs, ahead of mlp[3]
                    const float* R = g.residual + (b * GC + 16 * wave + 4 * qp) * GN;
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt)
                        if (nt < 4 || j == 0) res[nt] = f4v{R[16 * nt + j], R[GN + 16 * nt + j], R[2 * GN + 16 * nt + j], R[3 * GN + 16 * nt + j]};
                }
            } else {
                ring_fill<4, 1>(g.pw + PW_K, mt1, lane, rk);       // the next problem's first stage
                if (b + gridDim.x < g.batch) {                     // and its descriptors: in flight under this epilogue
                    load_in(g.x + (b + gridDim.x) * (GC * GN), t, xin);
                    load_in(g.source + (b + gridDim.x) * (GC * GN), t, sin);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int ch = 16 * mt2[m] + 4 * qp;
                const f4v bias = load4(pb + PB_1 + ch);
                if (TRAIN) {
                    float* H = g.hid + (b * 256 + ch) * GN;
                    f4v p1 = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt) {
                        const f4v v = fma4(acc[m][nt], bcast4(UNS), bias);            // = acc UNS + bias bit for bit (UNS is a power of two)
                        if (nt < 4 || j == 0) {
                            p1 = p1 + v;
                            p2 = p2 + v * v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                H[r * GN + 16 * nt + j] = v[r];
                                bad |= !(fabsf(v[r]) <= 3.0e38f);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a1 = p1[r], a2 = p2[r];
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }     // the 16 lanes of the row tile's token axis
                        st1[m][r] += (double)a1;
                        st2[m][r] += (double)a2;
                    }
                } else {
                    // bias, BatchNorm affine and the split's factor in ONE fma per value: ((acc UNS + bias) sc + sh) PRE =
                    // acc (sc UNS PRE) + (bias sc + sh) PRE (mul, add, mul, add, mul before: -ffp-contract=off; the ReLU commutes with PRE)
                    const f4v sc = load4(g.bn_a + ch), sh = load4(g.bn_b + ch);
                    const f4v scl = sc * (UNS * PRE), shf = (bias * sc + sh) * PRE;
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt) {
                        f4v v = fma4(acc[m][nt], scl, shf);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];        // ReLU that keeps NaN
                        store_tf<true>(lds + (m == 0 ? OFF_H0 : OFF_V), wave, nt, v, lane);
                    }
                }
            }
        }
        wg_barrier();
        if (!TRAIN) {
            // ---- out = W2 relu(bn(hidden)) + b2 [+ residual] -----------------------------------------------------------
            if (b + gridDim.x < g.batch) {       // the next problem's descriptors: in flight under this stage
                load_in(g.x + (b + gridDim.x) * (GC * GN), t, xin);
                load_in(g.source + (b + gridDim.x) * (GC * GN), t, sin);
            }
            f4v acc[1][5];
            zero_acc(acc);
            gemm_w<8, 1, false>(g.pw + PW_2, mt1, lds + OFF_H0, lds + OFF_V, lane, acc, r2);
            ring_fill<4, 1>(g.pw + PW_K, mt1, lane, rk);           // the next problem's first stage
            const int ch = 16 * wave + 4 * qp;
            const f4v bias = load4(pb + PB_2 + ch);
            float* O = g.out + (b * GC + ch) * GN;
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                if (nt < 4 || j == 0) {
                    const f4v v = fma4(acc[0][nt], bcast4(UNS), bias);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float o = v[r];
                        if (g.residual) o = res[nt][r] + o;
                        O[r * GN + 16 * nt + j] = o;
                        bad |= !(fabsf(o) <= 3.0e38f);
                    }
                }
            }
            wg_barrier();                        // the hidden tensor has been read: the next problem may land
        }
    }
    if (bad) atomicOr(g.flag, 1);
    if (TRAIN && j == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 16 * mt2[m] + 4 * qp + r;
                g.bn_part[((int64_t)ch * gridDim.x + blockIdx.x) * 2 + 0] = st1[m][r];
                g.bn_part[((int64_t)ch * gridDim.x + blockIdx.x) * 2 + 1] = st2[m][r];
            }
    }
}

// ---- the layer's tail on batch statistics: out = W2 relu(hidden * scale + shift) + b2 [+ residual] from the hidden tensor the
// kernel above wrote (TRAIN) and the scale / shift bn_finish_kernel made of its partial sums.  One problem per workgroup pass:
// hidden [256][65] fp32 -> affine + ReLU -> split -> TF layout (two 128-channel tensors) -> the mlp[3] stage of the fused
// kernel.  66.5 KB of LDS: two workgroups per CU.
__global__ void __launch_bounds__(512, 4)          // four waves per SIMD = two 8-wave workgroups per CU: 128 registers
gnn_tail_kernel(FusedArgs g, const float* __restrict__ scale, const float* __restrict__ shift) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int qp = lane >> 4, j = lane & 15;
    const int mt1[1] = {wave};
    bool bad = false;
    for (int64_t b = blockIdx.x; b < g.batch; b += gridDim.x) {
        WRing<1> r2;
        ring_fill<8, 1>(g.pw + PW_2, mt1, lane, r2);
        const float* H = g.hid + b * (256 * GN);
        // items (channel group of 8, token): 32 x 64 = four per thread, token 64: threads 0..31
        auto item = [&](int cg, int tok) {
            const float* p = H + (cg * 8) * GN + tok;
            const f4v sc0 = load4(scale + cg * 8), sc1 = load4(scale + cg * 8 + 4), sh0 = load4(shift + cg * 8), sh1 = load4(shift + cg * 8 + 4);
            f4v a = f4v{p[0], p[GN], p[2 * GN], p[3 * GN]} * sc0 + sh0, c = f4v{p[4 * GN], p[5 * GN], p[6 * GN], p[7 * GN]} * sc1 + sh1;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = a[r] < 0.f ? 0.f : a[r]; c[r] = c[r] < 0.f ? 0.f : c[r]; }
            h4v ah, al, ch_, cl;
            split4(a, ah, al);
            split4(c, ch_, cl);
            const int tile = tok >> 4, jj = tok & 15, kq = cg & 3;
            char* dst = lds + (cg >> 4) * TF_BYTES;
            const int off = (((cg & 15) >> 2) * 2) * TF_BLK + (tile < 4 ? tile * 1024 + (kq * 16 + jj) * 16 : 4096 + kq * 16);
            *reinterpret_cast<h8v*>(dst + off) = h8v{ah.x, ah.y, ah.z, ah.w, ch_.x, ch_.y, ch_.z, ch_.w};
            *reinterpret_cast<h8v*>(dst + off + TF_BLK) = h8v{al.x, al.y, al.z, al.w, cl.x, cl.y, cl.z, cl.w};
        };
#pragma unroll 2
        for (int it = 0; it < 4; ++it) item((t + 512 * it) >> 6, t & 63);
        if (t < 32) item(t, 64);
        wg_barrier();
        f4v acc[1][5];
        zero_acc(acc);
        gemm_w<8, 1, false>(g.pw + PW_2, mt1, lds, lds + TF_BYTES, lane, acc, r2);
        const int ch = 16 * wave + 4 * qp;
        const f4v bias = load4(g.pb + PB_2 + ch);
        float* O = g.out + (b * GC + ch) * GN;
        const float* R = g.residual ? g.residual + (b * GC + ch) * GN : nullptr;
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) {
            if (nt < 4 || j == 0) {
                const f4v v = fma4(acc[0][nt], bcast4(UNS), bias);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float o = v[r];
                    if (R) o = R[r * GN + 16 * nt + j] + o;
                    O[r * GN + 16 * nt + j] = o;
                    bad |= !(fabsf(o) <= 3.0e38f);
                }
            }
        }
        wg_barrier();
    }
    if (bad) atomicOr(g.flag, 1);
}

// Measured and not kept (round 4): the same layer with SIXTEEN waves per workgroup (four per SIMD, 128 registers each; the
// 128-row stages split their token tiles between two waves per row tile, mlp[0] one wave per row tile, the attention's twenty
// units in two rounds instead of three): 3.62 against 2.95 ms per 25 920 problems - 81-111 spilled registers at the 128 budget
// and the row tiles' weight fragments streamed twice.

// ---- host side ----------------------------------------------------------------------------------------------------------
int fused_layer_supported(int C, int heads, int n, int m) {
    static const bool off = [] { const char* e = env_switch("PATS_GNN_FUSED"); return e && atoi(e) == 0; }();
    return !off && C == GC && heads == 4 && n == GN && m == GN;
}

}  // namespace pats

using namespace pats;

// The packed buffer: [ the fused layer's section (only C = 128, 4 heads) | the six matrices for conv_pk_kernel, any C ]
namespace pats {
size_t packed_fused_bytes(int C, int heads) {
    return (C == GC && heads == 4) ? (((size_t)PW_END * sizeof(h8v) + (size_t)PB_END * sizeof(float) + 255) & ~(size_t)255) : 0;
}
size_t conv_packed_bytes(int K, int M);                                                                     // conv_pk.hip
int launch_conv_pack(const float* wt, int K, int M, void* packed, hipStream_t st);
// merge folded into mlp[0] (gnn_fold_kernel): on unless PATS_GNN_FOLD=0; the same answer at pack time and at run time
bool gnn_fold_enabled() {
    static const bool on = [] { const char* e = env_switch("PATS_GNN_FOLD"); return !(e && atoi(e) == 0); }();
    return on;
}
static size_t packed_matrices_bytes(int C, int heads) {
    return packed_fused_bytes(C, heads) + 4 * conv_packed_bytes(C, C) + conv_packed_bytes(2 * C, 2 * C) + conv_packed_bytes(2 * C, C);
}
// behind the packed matrices: the folded fp32 weights [2C][2C] and bias [2C] (the bias is read at run time)
const float* packed_folded_bias(const void* packed, int C, int heads) {
    return (const float*)((const char*)packed + packed_matrices_bytes(C, heads)) + (size_t)4 * C * C;
}
// and behind those (round 5): the fine level's one-kernel layer has its own fragment set (gnn_fine.hip; only C = 264, 4 heads)
size_t packed_fine_bytes(int C, int heads);
int launch_fine_pack(const pats_propagation_weights& w, void* section, hipStream_t st);
static size_t packed_fine_offset(int C, int heads) {
    return (packed_matrices_bytes(C, heads) + ((size_t)4 * C * C + 2 * C) * sizeof(float) + 255) & ~(size_t)255;
}
const void* packed_fine_section(const void* packed, int C, int heads) {
    return packed_fine_bytes(C, heads) ? (const char*)packed + packed_fine_offset(C, heads) : nullptr;
}
}

extern "C" size_t pats_propagation_packed_bytes(int C, int heads) {
    if (C <= 0 || heads <= 0 || (C % 8) != 0 || (C % heads) != 0) return 0;
    return packed_fine_offset(C, heads) + packed_fine_bytes(C, heads);
}

extern "C" int pats_propagation_pack_f32(const pats_propagation_weights* w, int C, int heads, void* packed, size_t packed_bytes,
                                         pats_stream_t stream) {
    PATS_REQUIRE(C > 0 && heads > 0 && (C % 8) == 0 && (C % heads) == 0, "propagation_pack: bad shape");
    PATS_REQUIRE(w && packed && packed_bytes >= pats_propagation_packed_bytes(C, heads), "propagation_pack: null pointer / buffer too small");
    PATS_REQUIRE(w->wq_t && w->bq && w->wk_t && w->bk && w->wv_t && w->bv && w->wm_t && w->bm && w->w1_t && w->b1 && w->w2_t && w->b2,
                 "propagation_pack: null weight pointer");
    PATS_REQUIRE(((uintptr_t)packed & 15) == 0, "propagation_pack: the buffer must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    pats_propagation_weights wf = *w;
    const bool fold = gnn_fold_enabled();
    if (fold) {
        float* w1f = (float*)((char*)packed + packed_matrices_bytes(C, heads));
        float* b1f = w1f + (size_t)4 * C * C;
        hipLaunchKernelGGL(gnn_fold_kernel, dim3((unsigned)(((int64_t)4 * C * C + 255) / 256)), dim3(256), 0, st, *w, C, w1f, b1f);
        int rc = check_launch("gnn_fold_kernel");
        if (rc) return rc;
        wf.w1_t = w1f;
        wf.b1 = b1f;
    }
    w = &wf;
    if (fold && packed_fine_bytes(C, heads)) {       // the fine level's fused layer runs on the FOLDED mlp[0] only
        PATS_REQUIRE(w->bn_a && w->bn_b, "propagation_pack: the fine level's packed layer keeps the BatchNorm scale / shift (eval mode)");
        int rc = launch_fine_pack(wf, (char*)packed + packed_fine_offset(C, heads), st);
        if (rc) return rc;
    }
    if (packed_fused_bytes(C, heads)) {
        h8v* pw = (h8v*)packed;
        float* pb = (float*)(pw + PW_END);
        const int threads = (PW_END / FR) * 64;
        hipLaunchKernelGGL(gnn_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *w, pw, pb, fold ? 1 : 0);
        int rc = check_launch("gnn_pack_kernel");
        if (rc) return rc;
    }
    char* p = (char*)packed + packed_fused_bytes(C, heads);
    const float* mats[6] = {w->wq_t, w->wk_t, w->wv_t, w->wm_t, w->w1_t, w->w2_t};
    const int Ks[6] = {C, C, C, C, 2 * C, 2 * C}, Ms[6] = {C, C, C, C, 2 * C, C};
    for (int i = 0; i < 6; ++i) {
        int rc = launch_conv_pack(mats[i], Ks[i], Ms[i], p, st);
        if (rc) return rc;
        p += conv_packed_bytes(Ks[i], Ms[i]);
    }
    return PATS_OK;
}

namespace pats {
// The layer (eval: whole; train: up to the hidden tensor) for batch problems.  flag: one int, zero on entry.
int launch_fused_layer(const float* x, const float* source, int64_t batch, const void* packed, const float* bn_a, const float* bn_b,
                       int bn_train, const float* residual, float* out, float* hid, int* flag, double* bn_part, int* splits_out,
                       hipStream_t st, const int64_t* live, int64_t live_off) {
    struct PerDevice { int state = 0; int n_cu = 256; };
    static PerDevice per_dev[64];
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) { (void)hipGetLastError(); dev_id = 0; }
    PerDevice& pd = per_dev[dev_id];
    if (pd.state == 0) {
        const bool ok = hipFuncSetAttribute((const void*)gnn_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TF_BYTES) == hipSuccess &&
                        hipFuncSetAttribute((const void*)gnn_layer_fused_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)gnn_layer_fused_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)gnn_layer_fused_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)gnn_layer_fused_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        int v = 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess) (void)hipGetLastError();
        pd.n_cu = v > 0 ? v : 256;
        pd.state = ok ? 1 : -1;
    }
    if (pd.state != 1) return PATS_ERR_UNSUPPORTED;
    const h8v* pw = (const h8v*)packed;
    FusedArgs g{x, source, residual, out, hid, pw, (const float*)(pw + PW_END), bn_a, bn_b, batch, flag, bn_part, bn_train ? nullptr : live, live_off};
    const unsigned grid = (unsigned)std::min<int64_t>(batch, std::min(pd.n_cu, FUSED_MAX_GRID));
    if (splits_out) *splits_out = (int)grid;
    const bool fold = gnn_fold_enabled();
    if (bn_train && fold) hipLaunchKernelGGL((gnn_layer_fused_kernel<true, true>), dim3(grid), dim3(512), FUSED_LDS, st, g);
    else if (bn_train) hipLaunchKernelGGL((gnn_layer_fused_kernel<true, false>), dim3(grid), dim3(512), FUSED_LDS, st, g);
    else if (fold) hipLaunchKernelGGL((gnn_layer_fused_kernel<false, true>), dim3(grid), dim3(512), FUSED_LDS, st, g);
    else hipLaunchKernelGGL((gnn_layer_fused_kernel<false, false>), dim3(grid), dim3(512), FUSED_LDS, st, g);
    return check_launch("gnn_layer_fused_kernel");
}

// mlp[1..3] of the layer on batch statistics, from the hidden tensor (see gnn_tail_kernel)
int launch_gnn_tail(const float* hid, int64_t batch, const void* packed, const float* scale, const float* shift, const float* residual,
                    float* out, int* flag, hipStream_t st) {
    int dev_id = 0, n_cu = 256;
    if (hipGetDevice(&dev_id) != hipSuccess) { (void)hipGetLastError(); dev_id = 0; }
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess || n_cu <= 0) { (void)hipGetLastError(); n_cu = 256; }
    const h8v* pw = (const h8v*)packed;
    FusedArgs g{nullptr, nullptr, residual, out, const_cast<float*>(hid), pw, (const float*)(pw + PW_END), nullptr, nullptr, batch, flag, nullptr, nullptr, 0};
    const unsigned grid = (unsigned)std::min<int64_t>(batch, 2 * (int64_t)n_cu);
    hipLaunchKernelGGL(gnn_tail_kernel, dim3(grid), dim3(512), 2 * TF_BYTES, st, g, scale, shift);
    return check_launch("gnn_tail_kernel");
}
}  // namespace pats
