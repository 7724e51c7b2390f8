// AttentionalPropagation (models/modules.py:107-117) + the residual of AttentionalGNN.forward (:131-133) at the FINE level's shape
// - x, source [b, 264, 145], 4 heads of 66 channels (second_layer.py:44,89 runs 18 such layers on both descriptor sets) - in THREE
// kernels a layer (round 5).  Round 4 ran this layer as six conv_pk_kernel launches around attention145_kernel: 24 tensor passes of
// 153 KB per problem through HBM and 360 KB of packed weights streamed per 64-column tile (5.7 MB per problem through each CU's L1).
//
//   q = Wq x, k = Wk s, v = Wv s;  att = softmax(q^T k / sqrt(66)) v  per head           MultiHeadedAttention.forward :100-105
//   hidden = relu(bn(W1x x + (W1m Wm) att + b1'))                                          (merge folded into mlp[0]: gnn_fold_kernel)
//   out = x + W2 hidden + b2                                                              AttentionalPropagation :114-117, GNN :133
//
// Everything but the attention core is per TOKEN, so it runs on 64-column tiles of the FLATTENED list of (problem, 16-token tile)
// pairs (gnn_fine_tile_kernel: no token padding - 145 = 9 x 16 + 1 costs a per-problem kernel 10 %); the attention core runs per
// problem (gnn_fine_attn_kernel).  A layer = attention of layer l, then ONE tile launch that does the MLP of layer l and, from the
// output tile still in LDS, the q / k / v projections of layer l + 1.  (The first version of this file was one persistent kernel per
// problem - DESIGN.md section 7a has its timeline and why it lost: its far-memory phases ran at the 1 / 256 share of the memory side.)
//   * a [264 x 145] tensor split for the fp16 matrix pipe (x 2^6 = hi + lo) in MFMA fragment order is 153 120 bytes ("TF image":
//     per 32-channel k-step and 16-token tile, lane (k / 8, token) holds its 8 channels as one 16-byte piece = the B operand of
//     v_mfma_f32_16x16x32_f16; token 144 and channels 256..263 are ragged blocks without padding).  Operands reach LDS by DMA
//     (global_load_lds_dwordx4: no registers, no VALU), a product's OUTPUT lives in the accumulators and leaves split again, in the
//     next consumer's fragment order.
//   * the WEIGHTS are the A operand straight from L2 into a register ring (pre-split fragments, one 16-byte load per lane, packed once
//     per layer by gnn_fine_pack_kernel): 2.8 MB per 64-column tile.
//   * descriptors travel BETWEEN layers as TF images: the layer's epilogue writes one, the next layer gathers it and rebuilds the
//     residual from it ((hi + lo) / 2^6: exact to the 22 bits an image holds) - 16-byte accesses everywhere, no 4-byte strided
//     loads (round 4's limiter).  gnn_fine_in_kernel / gnn_fine_out_kernel convert at the ends of a stack.
//   * heads: the reference views a projection as [b, 66, 4, n] (channel = d * 4 + h).  q / k / v rows are permuted at pack time to
//     [head][d < 64] (head h = k-steps 2 h, 2 h + 1 of a TF image) followed by the eight "extra" channels (d = 64, 65 of each head) in
//     the ragged 17th row tile; the folded mlp[0] matrix has its attention columns in the same order.
//   * attention per (head, 16-query tile) as in csrc/attention145.hip: S^T = K^T Q with the keys as rows (softmax in-lane + two
//     exchanges), the accumulators of key tiles 2 kk, 2 kk + 1 ARE the B operand of out^T = V P^T; v is produced TRANSPOSED by its
//     projection (activations as A, weights as B) directly in that A-fragment order.  The two extra channels of a head cost ONE more
//     MFMA per key tile instead of a k-step of three: A = (kh0 kh1 kl0 kl1 kh0 kh1 0 0), B = (qh0 qh1 qh0 qh1 ql0 ql1 0 0) gives
//     hi.hi + lo.hi + hi.lo of both channels.
// BatchNorm in eval mode only (the second layer always is: pats.py:112-114).  Range: |activation| < 1023; a non-finite output raises
// *flag and the round-2 composition queued behind, gated on it, redoes the layer.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#ifdef PATS_DIAG
#include <cstdio>
#include <vector>
#endif

namespace pats {

namespace {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) h8v* gptr_h8;

constexpr int FC = 264, FN = 145, FNT = 10;
constexpr float PRE = 64.0f, UNS = 1.0f / 4096.0f;
constexpr int TFB = 9 * 1024 + 64;            // one (k-step, hi | lo) block: nine full token tiles + token 144 (four 16-byte pieces)
constexpr int TFR = 9 * 256 + 16;             // the ragged k-step (channels 256..263: lanes k / 8 = 0 only), per plane
constexpr int TF_MAIN = 16 * TFB;             // 148 480
constexpr int TF_BYTES = TF_MAIN + 2 * TFR;   // 153 120
constexpr int XB = 4 * FN * 16;               // the extra channels of q / k: [head][token] one 16-byte packed piece
constexpr int V_TILE = 5 * 2 * 1024;          // one 16-channel tile of v as A fragments: [key pair tile kk][hi | lo][64 lanes x 16 B]
constexpr int V_BYTES = 17 * V_TILE;
// LDS while the attention runs
constexpr int KH_BYTES = 4 * TFB + FN * 16;   // one head of k: two k-steps x (hi, lo) + its extras
// packed weights (units of h8v): fragment (row tile, k-step) = [hi | lo][64 lanes]
constexpr int FR = 128;
constexpr int FW_Q = 0, FW_K = FW_Q + 17 * 9 * FR, FW_V = FW_K + 17 * 9 * FR, FW_1 = FW_V + 17 * 9 * FR, FW_2 = FW_1 + 34 * 18 * FR,
              FW_END = FW_2 + 17 * 18 * FR;
// biases behind them (floats): q', k', v' (permuted), b1' (two halves), bn scale, bn shift, b2 - every vector padded to 272
constexpr int FB_Q = 0, FB_K = 272, FB_V = 544, FB_1 = 816, FB_A = FB_1 + 544, FB_S = FB_A + 544, FB_2 = FB_S + 544, FB_END = FB_2 + 272;

// Diagnostic library only (python -m pats_amd.build --diag, PATS_AMD_DIAG_LIB=1, PATS_FINE_TL=1): s_memrealtime stamps at the stage
// boundaries of thread 0, summed over the workgroup's problems; launch_fine_layer prints the means per problem
[[maybe_unused]] constexpr int FT_N = 24;          // (diagnostic library)
#ifdef PATS_DIAG
#define FT(k) { __builtin_amdgcn_sched_barrier(0); const long long now_ = __builtin_amdgcn_s_memrealtime(); tsum[k] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define FT(k) do { } while (0)
#endif

__device__ __forceinline__ int tf_blk(int ks, int plane) { return ks < 8 ? (2 * ks + plane) * TFB : TF_MAIN + plane * TFR; }
// byte offset of fragment piece (token tile t, lane) inside a block; ragged k-step: valid for lanes < 16 only
__device__ __forceinline__ int tf_off(bool ragged, int t, int lane) {
    if (!ragged) return t < 9 ? t * 1024 + lane * 16 : 9216 + (lane >> 4) * 16;
    return t < 9 ? t * 256 + (lane & 15) * 16 : 2304;
}

__device__ __forceinline__ void split4_pre(const f4v s, h4v& hi, h4v& lo) {       // s already carries the factor PRE
    hi = __builtin_convertvector(s, h4v);
    lo = __builtin_convertvector(s - __builtin_convertvector(hi, f4v), h4v);
}
__device__ __forceinline__ f4v fma4(const f4v a, const f4v b, const f4v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f4v bcast4(float x) { return f4v{x, x, x, x}; }
__device__ __forceinline__ f4v load4(const float* p) { return *reinterpret_cast<const f4v*>(p); }
__device__ __forceinline__ h8v zero8() { return h8v{0, 0, 0, 0, 0, 0, 0, 0}; }

__device__ __forceinline__ f4v mfma3(const h8v ah, const h8v al, const h8v bh, const h8v bl, f4v c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);       // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
}

__device__ __forceinline__ gptr_h8 uniform_ptr(const h8v* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr_h8)(((uint64_t)hi << 32) | lo);
}

// ---- global -> LDS, 16 bytes a lane, no registers (bytes % 16 == 0; src 16-byte aligned).  Completion: vmcnt. ----------------
template <int BYTES>
__device__ __forceinline__ void dma_fill(char* lds_dst, const char* src, int wave, int lane) {
    static_assert(BYTES % 16 == 0, "whole 16-byte pieces");
    constexpr int PIECES = BYTES / 16, FULL = PIECES / 512, REST = PIECES - 512 * FULL;
    const char* s = src + (size_t)(wave * 64 + lane) * 16;
    char* d = lds_dst + wave * 1024;
    // a ROLLED loop on one running address pair: unrolled, every round keeps its own 64-bit vector address (the immediate offset
    // of the instruction reaches 4 KB) and eighteen of them spill the accumulators around the call
#pragma unroll 1
    for (int r = 0; r < FULL; ++r) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        s += 8192;
        d += 8192;
    }
    if (REST > 0 && wave * 64 < REST) {                    // the last, partial round: whole waves, then one partial wave
        if (wave * 64 + lane < REST)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    }
}

// rows 16 mt + 4 g + r (r = 0..3) of token 16 t + j, value v = PRE x the element -> a TF image at dst (global memory).
// A lane's four values are HALF of a 16-byte piece (lanes g = 2 k, 2 k + 1 share one): the pair exchanges halves through
// v_permlane16_swap - the even row of 16 lanes ends up with the whole hi piece, the odd row with the whole lo piece - and every lane
// issues ONE 16-byte store; a wave's store covers whole 256-byte runs.  (The first version stored 8-byte halves: every 128-byte line
// of an image was written by two instructions with half its bytes enabled - read-for-ownership traffic at the memory side; the
// output epilogue alone took 20 us of a problem's 230.)  Called by all 64 lanes (the exchange), inactive tiles masked at the store.
// (addresses: uniform base + per-store scalar constant + one of three 32-bit lane offsets made once per problem - global_store
//  with an SGPR base; per-store 64-bit vector addresses were what the epilogues spilled and reloaded behind s_waitcnt vmcnt(0))
struct LaneOff { unsigned tf0, tf1, tf2; };
__device__ __forceinline__ LaneOff lane_offsets(int lane) {
    const unsigned g = (unsigned)lane >> 4, j = (unsigned)lane & 15u;
    LaneOff o;
    o.tf0 = (g & 1u) * TFB + (g >> 1) * 256u + j * 16u;    // full k-step, token tiles 0..8
    o.tf1 = (g & 1u) * TFB + (g >> 1) * 16u;               // full k-step, token 144 (lanes j = 0)
    o.tf2 = (g & 1u) * TFR + j * 16u;                      // ragged k-step (lanes g < 2)
    return o;
}
__device__ __forceinline__ void store_tf(char* dst, int mt, int t, const f4v v, int lane, const LaneOff& lo_) {
    const int g = lane >> 4, j = lane & 15;
    h4v hi, lo;
    split4_pre(v, hi, lo);
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const u2v H = __builtin_bit_cast(u2v, hi), L = __builtin_bit_cast(u2v, lo);
    // lane_swap16(a, b): odd rows of a <-> even rows of b.  Afterwards an even-row lane holds (own hi, partner's hi) = the hi piece,
    // an odd-row lane (partner's lo, own lo) = the lo piece - both as (H, L)
    unsigned hx = H.x, hy = H.y, lx = L.x, ly = L.y;
    lane_swap16(hx, lx);
    lane_swap16(hy, ly);
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v piece = {hx, hy, lx, ly};
    if (mt < 16) {
        char* d = dst + (mt >> 1) * (2 * TFB) + (mt & 1) * (t < 9 ? 512 : 32) + (t < 9 ? t * 1024 : 9216);
        if (t < 9) *reinterpret_cast<u4v*>(d + lo_.tf0) = piece;
        else if (j == 0) *reinterpret_cast<u4v*>(d + lo_.tf1) = piece;
    } else {
        char* d = dst + TF_MAIN + (t < 9 ? t * 256 : 2304);
        if (g < 2 && (t < 9 || j == 0)) *reinterpret_cast<u4v*>(d + lo_.tf2) = piece;
    }
}

// position p of the permuted q / k / v row order (and of the attention channels) -> the reference's channel d * 4 + h
__host__ __device__ inline int perm_channel(int p) {
    const int h = p < 256 ? (p >> 6) : ((p - 256) >> 1), d = p < 256 ? (p & 63) : 64 + ((p - 256) & 1);
    return d * 4 + h;
}

}  // namespace

// ---- weights of one layer -> fragments.  One thread per (fragment, lane).  w.w1_t / w.b1 are the FOLDED ones (gnn_fold_kernel). ----
__global__ void __launch_bounds__(256)
gnn_fine_pack_kernel(pats_propagation_weights w, h8v* __restrict__ pw, float* __restrict__ pb) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid < FB_END) {
        float v = 0.f;
        if (gid < FB_1) {                                   // q', k', v' biases in the permuted order
            const int which = gid / 272, p = gid - 272 * which;
            const float* b = which == 0 ? w.bq : which == 1 ? w.bk : w.bv;
            if (p < FC) v = b[perm_channel(p)];
        } else if (gid < FB_2) {                            // b1', bn scale, bn shift: two halves of 264, each padded to 272
            const int which = (gid - FB_1) / 544, i = (gid - FB_1) - 544 * which, hf = i / 272, p = i - 272 * hf;
            const float* b = which == 0 ? w.b1 : which == 1 ? w.bn_a : w.bn_b;
            if (p < FC) v = b[hf * FC + p];
        } else {
            const int p = gid - FB_2;
            if (p < FC) v = w.b2[p];
        }
        pb[gid] = v;
    }
    const int lane = gid & 63, f = gid >> 6;
    if (f >= FW_END / FR) return;
    const float* wt;
    int M, mt, ks, kind;                                    // kind 0: q / k / v; 1: mlp[0] (folded); 2: mlp[3]
    if (f < 3 * 153) { const int mat = f / 153, r = f - 153 * mat; wt = mat == 0 ? w.wq_t : mat == 1 ? w.wk_t : w.wv_t; M = FC; mt = r / 9; ks = r - 9 * mt; kind = 0; }
    else if (f < 3 * 153 + 34 * 18) { const int r = f - 3 * 153; wt = w.w1_t; M = 2 * FC; mt = r / 18; ks = r - 18 * mt; kind = 1; }
    else { const int r = f - 3 * 153 - 34 * 18; wt = w.w2_t; M = FC; mt = r / 18; ks = r - 18 * mt; kind = 2; }
    // the output row of this lane
    const int pr = (kind == 1 ? (mt % 17) : mt) * 16 + (lane & 15);         // position inside a 272-row group
    int row = -1;
    if (pr < FC) row = kind == 0 ? perm_channel(pr) : kind == 1 ? (mt / 17) * FC + pr : pr;
    h8v hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int part = ks / 9, kk = ks - 9 * part;                        // input tensor (x | att, hidden0 | hidden1) and its k-step
        const int pk = kk * 32 + 8 * (lane >> 4) + e;                       // position inside that tensor
        const bool live = kk < 8 ? true : (lane >> 4) == 0;                 // ragged k-step: positions 256..263 in lanes 0..15
        int k = -1;
        if (live && pk < FC) k = kind == 1 && part == 1 ? FC + perm_channel(pk) : part * FC + pk;
        float s = 0.f;
        if (row >= 0 && k >= 0) s = wt[(int64_t)k * M + row] * PRE;
        const _Float16 h = (_Float16)s;
        hi[e] = h;
        lo[e] = (_Float16)(s - (float)h);
    }
    pw[(size_t)f * FR + lane] = hi;
    pw[(size_t)f * FR + 64 + lane] = lo;
}

// ---- [P][264][145] fp32 -> TF image (and back).  One thread per (problem, 8-channel group, token). -------------------------------
__global__ void __launch_bounds__(256)
gnn_fine_in_kernel(const float* __restrict__ x, int64_t P, char* __restrict__ tf) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= P * 33 * FN) return;
    const int64_t p = gid / (33 * FN);
    const int r = (int)(gid - p * (33 * FN)), cg = r / FN, tok = r - cg * FN;
    const float* src = x + (p * FC + cg * 8) * FN + tok;
    const f4v a = {src[0], src[FN], src[2 * FN], src[3 * FN]}, b = {src[4 * FN], src[5 * FN], src[6 * FN], src[7 * FN]};
    h4v ah, al, bh, bl;
    split4_pre(a * PRE, ah, al);
    split4_pre(b * PRE, bh, bl);
    const int ks = cg >> 2, kq = cg & 3, t = tok >> 4, j = tok & 15;
    char* img = tf + p * TF_BYTES;
    const int off = tf_off(ks == 8, t, kq * 16 + j);
    *reinterpret_cast<h8v*>(img + tf_blk(ks, 0) + off) = h8v{ah.x, ah.y, ah.z, ah.w, bh.x, bh.y, bh.z, bh.w};
    *reinterpret_cast<h8v*>(img + tf_blk(ks, 1) + off) = h8v{al.x, al.y, al.z, al.w, bl.x, bl.y, bl.z, bl.w};
}

__global__ void __launch_bounds__(256)
gnn_fine_out_kernel(const char* __restrict__ tf, int64_t P, float* __restrict__ y, const int64_t* __restrict__ live, int64_t live_off,
                    const float* __restrict__ add) {
    // rows past the device-side count (no layer computed them): zeros, so that whatever runs over the capacity next reads finite values
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= P * 33 * FN) return;
    const int64_t p = gid / (33 * FN);
    const int r = (int)(gid - p * (33 * FN)), cg = r / FN, tok = r - cg * FN;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!live || p < *live - live_off) {
        const int ks = cg >> 2, kq = cg & 3, t = tok >> 4, j = tok & 15;
        const char* img = tf + p * TF_BYTES;
        const int off = tf_off(ks == 8, t, kq * 16 + j);
        const h8v hi = *reinterpret_cast<const h8v*>(img + tf_blk(ks, 0) + off), lo = *reinterpret_cast<const h8v*>(img + tf_blk(ks, 1) + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((float)hi[e] + (float)lo[e]) * (1.0f / PRE);
    }
    float* d = y + (p * FC + cg * 8) * FN + tok;
    if (add) {                 // a residual that is neither null nor the layer's own x (the single-layer entry): added here, in fp32
        const float* a_ = add + (p * FC + cg * 8) * FN + tok;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += a_[e * FN];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e * FN] = v[e];
}

// ---- column tiles of the per-token products -------------------------------------------------------------------------------------
// Everything of the layer except the attention core is per TOKEN: q / k / v = W x, hidden = relu(bn(W1x x + W1a att + b1')),
// out = W2 hidden + b2 + x.  gnn_fine_tile_kernel runs those on tiles of 64 columns = four 16-token tiles taken from the flattened
// list of (problem, token tile) pairs, so that no token padding is computed (145 = 9 x 16 + 1 costs a per-problem kernel 10 %):
//   class F  tiles 0..3 and 4..7 of one problem (two workgroup tiles a problem),
//   class E  tile 8 of four consecutive problems,
//   class T  token 144 of 64 consecutive problems (column tile ct, column j = problem 64 n + 16 ct + j).
// A column tile of classes F / E is a 16-token tile of ONE problem, aligned as the attention kernel's fragments are - which is what
// lets v leave TRANSPOSED (activations as the A operand) directly as the A fragments of out^T = V P^T.
// MLP: x-tile + att-tile (2 x 66 KB as B fragments) sit in LDS together; mlp[0] is one 18-k-step loop whose 528 x 64 output lives
// in the accumulators (17 tile units a wave), is written - BatchNorm, ReLU, split - over the operands it came from and feeds mlp[3]
// from there: the hidden tensor never leaves the CU.  The residual is rebuilt from the x fragments before they are overwritten.
// QKV: the projections of the NEXT layer from the tile the MLP has just produced (written to LDS as fragments besides leaving as
// the next layer's image) - or, QKV alone, of an image tile fetched for it (the first layer of a stack).  q and k leave as TF images
// + the packed extras, v^T as A fragments, into the per-problem block the attention kernel reads (QKV_BYTES a problem).
// Operands arrive by gather DMA (a lane's source address is its column's; the 64 pieces of a fragment land contiguously).
constexpr int MT_HALF = 8 * 8192 + 2 * 1024;          // one operand half-tile: eight full k-steps [plane][4 column tiles][64 x 16 B] + the ragged one
constexpr int L_PB = 2 * MT_HALF;                     // the layer's bias / BatchNorm vectors (FB_END floats) and the q / k / v biases of the
constexpr int L_PBQ = L_PB + FB_END * 4;              // projections' layer (FB_1 floats): read by every epilogue - from LDS, not through L2
constexpr int MLP_LDS = L_PBQ + FB_1 * 4;             // 149 312
// per-problem block of projections
constexpr int QO_Q = 0, QO_QX = QO_Q + TF_MAIN, QO_K = QO_QX + XB, QO_KX = QO_K + TF_MAIN, QO_V = QO_KX + XB, QKV_BYTES = QO_V + V_BYTES;
static_assert(QKV_BYTES % 16 == 0, "16-byte pieces");

struct TileArgs {
    const char* tf_x;          // [P] TF images: the layer's x (MLP) / the tensor to project (QKV alone)
    const char* tf_att;        // [P] the attention output (MLP)
    char* tf_out;              // [P] the layer's output (MLP)
    char* qkv;                 // [P][QKV_BYTES] projections (QKV)
    const h8v* pw; const float* pb;          // the layer's packed section (MLP)
    const h8v* pwq; const float* pbq;        // the section whose q / k / v matrices the QKV phase applies
    int64_t P, half; int sets;
    const int64_t* live; int64_t live_off;
    int* flag;
    int residual;              // MLP: add x (AttentionalGNN.forward's desc + delta); 0: the delta alone
    int want_q, want_kv;       // QKV: which of the projections leave
    char* dump;                // 256 bytes nobody reads: where a lane with nothing to store stores (see `sink` in the kernel)
    const int* gate;
#ifdef PATS_DIAG
    long long* tl;
#endif
};

// byte offset of token t's 16-byte piece (k-group kq) inside a (k-step, plane) block of a per-problem image
__device__ __forceinline__ int img_tok_off(bool ragged, int t, int kq) {
    if (!ragged) return t < 144 ? (t >> 4) * 1024 + kq * 256 + (t & 15) * 16 : 9216 + kq * 16;
    return t < 144 ? (t >> 4) * 256 + (t & 15) * 16 : 2304;
}

// B fragment pair of k-step kk (0..8; 8 = ragged) of operand half `part` of the tile in LDS, column tile ct
__device__ __forceinline__ void bload1(const char* lds, int part, int kk, int ct, int lane, h8v& bh, h8v& bl) {
    const char* base = lds + part * MT_HALF;
    if (kk < 8) {
        bh = *reinterpret_cast<const h8v*>(base + kk * 8192 + ct * 1024 + lane * 16);
        bl = *reinterpret_cast<const h8v*>(base + kk * 8192 + 4096 + ct * 1024 + lane * 16);
    } else {
        const h8v a = *reinterpret_cast<const h8v*>(base + 65536 + ct * 256 + (lane & 15) * 16);
        const h8v b = *reinterpret_cast<const h8v*>(base + 65536 + 1024 + ct * 256 + (lane & 15) * 16);
        bh = lane < 16 ? a : zero8();
        bl = lane < 16 ? b : zero8();
    }
}

// The weights are the A operand straight from L2 into a register ring of two k-steps.  On entry slot 0 holds k-step 0 - wload()
// issued by the caller BEFORE the epilogue of the product in front, so that no product starts with an L2 round trip.  (A ring of
// three k-steps - 20 KB a wave in flight - changes nothing: the stream is not latency-bound; it costs 40 registers and spills.)
template <int N>
__device__ __forceinline__ void wload(const h8v* const (&w)[N], int ks, int lane, h8v (&r)[N][2]) {
#pragma unroll
    for (int m = 0; m < N; ++m) {
        gptr_h8 Wf = uniform_ptr(w[m] + (size_t)ks * FR);
        r[m][0] = Wf[lane];
        r[m][1] = Wf[64 + lane];
    }
}
#if defined(PATS_DIAG) && defined(PATS_EXP_NOW)      // experiment: no weight stream (k-step 0's fragments for every k-step; wrong results)
#define WLOAD_IN_LOOP(N, W, ks, lane, r) do { } while (0)
#define WSLOT(e) 0
#else
#define WLOAD_IN_LOOP(N, W, ks, lane, r) wload<N>(W, ks, lane, r)
#define WSLOT(e) (e)
#endif
#if defined(PATS_DIAG) && defined(PATS_EXP_NOB)      // experiment: no B fragments from LDS after the first k-step (wrong results)
#define BKK(kk) 0
#else
#define BKK(kk) (kk)
#endif
// five row tiles (w[m]: fragment of k-step 0; k-step ks at + ks * FR) x four column tiles over NKS k-steps; the fifth - a ragged row
// tile shared by four waves - for column tile rag_ct only
// Two waves share a SIMD's matrix pipe, and when both have an MFMA ready the OLDER wave's is taken (priority, then age): waves 0..3
// ran a product's k loop nearly unimpeded, waves 4..7 on what was left - and finished it alone, with nobody to cover their LDS and
// weight latencies (thread 0 waited 5 of mlp[0]'s 21 us at the barrier behind it).  Half way through a k loop the younger wave takes
// priority 1: the older one leads the first half, the younger the second, both arrive together.
#if defined(PATS_DIAG) && defined(PATS_EXP_NOPRIO)
#define PRIO_SWAP(kp, NKS, younger) do { } while (0)
#define PRIO_END(younger) do { } while (0)
#else
#define PRIO_SWAP(kp, NKS, younger) do { if ((kp) == ((NKS) + 1) / 4 && (younger)) __builtin_amdgcn_s_setprio(1); } while (0)
#define PRIO_END(younger) do { if (younger) __builtin_amdgcn_s_setprio(0); } while (0)
#endif
struct W5 { const h8v* w[5]; };
template <int NKS>
__device__ __forceinline__ void prod5(const W5& W, h8v (&a)[2][5][2], const char* lds, int lane, int rag_ct, f4v (&acc)[4][4], f4v& accr, bool younger) {
#pragma unroll 1
    for (int kp = 0; kp < (NKS + 1) / 2; ++kp) {
        PRIO_SWAP(kp, NKS, younger);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ks = 2 * kp + e;
            if (ks < NKS) {
                // (UNCONDITIONAL - the last k-step fetches its own fragments again: behind `if (ks + 1 < NKS)` the compiler's one
                //  `s_waitcnt vmcnt(n)` for the fragments of THIS k-step must also be right on the path that skipped the loads, so n
                //  counted none of them and every k-step waited for the loads it had just issued - no prefetch distance at all)
                WLOAD_IN_LOOP(5, W.w, ks + 1 < NKS ? ks + 1 : NKS - 1, lane, a[1 - e]);
                const int part = ks >= 9 ? 1 : 0, kk = BKK(ks >= 9 ? ks - 9 : ks);
                h8v bh[4], bl[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) bload1(lds, part, kk, ct, lane, bh[ct], bl[ct]);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m][ct] = mfma3(a[WSLOT(e)][m][0], a[WSLOT(e)][m][1], bh[ct], bl[ct], acc[m][ct]);
                    if (ct == rag_ct) accr = mfma3(a[WSLOT(e)][4][0], a[WSLOT(e)][4][1], bh[ct], bl[ct], accr);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    PRIO_END(younger);
}
// three row tiles (the third for column tile rag_ct only; rag_ct < 0: none).  TRANS: the tile's fragments are the A operand, the
// weights B - the accumulators hold rows = tokens 4 g + r of the column tile, column = channel j of the weight tile
struct W3 { const h8v* w[3]; };
template <int NKS, bool TRANS>
__device__ __forceinline__ void prod3(const W3& W, h8v (&a)[2][3][2], const char* lds, int lane, int rag_ct, f4v (&o)[2][4], f4v& orr, bool younger) {
#pragma unroll 1
    for (int kp = 0; kp < (NKS + 1) / 2; ++kp) {
        PRIO_SWAP(kp, NKS, younger);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ks = 2 * kp + e;
            if (ks < NKS) {
                WLOAD_IN_LOOP(3, W.w, ks + 1 < NKS ? ks + 1 : NKS - 1, lane, a[1 - e]);       // (unconditional: see prod5)
                const int part = ks >= 9 ? 1 : 0, kk = BKK(ks >= 9 ? ks - 9 : ks);
                h8v bh[4], bl[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) bload1(lds, part, kk, ct, lane, bh[ct], bl[ct]);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const h8v xh = bh[ct], xl = bl[ct];
                    if (TRANS) {
                        o[0][ct] = mfma3(xh, xl, a[WSLOT(e)][0][0], a[WSLOT(e)][0][1], o[0][ct]);
                        o[1][ct] = mfma3(xh, xl, a[WSLOT(e)][1][0], a[WSLOT(e)][1][1], o[1][ct]);
                        if (ct == rag_ct) orr = mfma3(xh, xl, a[WSLOT(e)][2][0], a[WSLOT(e)][2][1], orr);
                    } else {
                        o[0][ct] = mfma3(a[WSLOT(e)][0][0], a[WSLOT(e)][0][1], xh, xl, o[0][ct]);
                        o[1][ct] = mfma3(a[WSLOT(e)][1][0], a[WSLOT(e)][1][1], xh, xl, o[1][ct]);
                        if (ct == rag_ct) orr = mfma3(a[WSLOT(e)][2][0], a[WSLOT(e)][2][1], xh, xl, orr);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    PRIO_END(younger);
}

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

// a lane's four values (rows 4 g + r of its column, already x PRE) -> the 16-byte piece its lane pair shares: afterwards an even
// row of 16 lanes holds the hi piece, an odd row the lo piece (see store_tf)
__device__ __forceinline__ u4v make_piece(const f4v v) {
    h4v hi, lo;
    split4_pre(v, hi, lo);
    const u2v H = __builtin_bit_cast(u2v, hi), Lo = __builtin_bit_cast(u2v, lo);
    unsigned hx = H.x, hy = H.y, lx = Lo.x, ly = Lo.y;
    lane_swap16(hx, lx);
    lane_swap16(hy, ly);
    return u4v{hx, hy, lx, ly};
}
// where that piece goes: row tile mt of the tile's operand half in LDS (column tile ct) ...
__device__ __forceinline__ int lds_piece_off(int mt, int ct, int gq, int j) {
    return mt < 16 ? (mt >> 1) * 8192 + (gq & 1) * 4096 + ct * 1024 + ((2 * (mt & 1) + (gq >> 1)) * 16 + j) * 16
                   : 65536 + (gq & 1) * 1024 + ct * 256 + j * 16;
}
// ... and of a per-problem TF image (token tk)
__device__ __forceinline__ int img_piece_off(int mt, int tk, int gq) {
    return mt < 16 ? ((mt >> 1) * 2 + (gq & 1)) * TFB + img_tok_off(false, tk, 2 * (mt & 1) + (gq >> 1))
                   : TF_MAIN + (gq & 1) * TFR + img_tok_off(true, tk, 0);
}

#ifdef PATS_DIAG
#define TT(k) { __builtin_amdgcn_sched_barrier(0); const long long now_ = __builtin_amdgcn_s_memrealtime(); tsum[k] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define TT(k) do { } while (0)
#endif

template <bool MLP, bool QKV>
__global__ void __launch_bounds__(512, 1)
gnn_fine_tile_kernel(TileArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (g.gate && *g.gate == 0) return;
    const int t_ = threadIdx.x, lane0 = t_ & 63, wave0 = __builtin_amdgcn_readfirstlane(t_ >> 6);
    int64_t L = g.half;
    if (g.live) { const int64_t l_ = *g.live - g.live_off; L = l_ < 0 ? 0 : (l_ < g.half ? l_ : g.half); }
    const int64_t NQ = L * g.sets;                         // live problems: index qi -> problem (qi / L) * half + qi % L
    const int64_t nF = 2 * NQ, nE = (NQ + 3) >> 2, nT = (NQ + 63) >> 6, ntile = nF + nE + nT;
    bool bad = false;
#ifdef PATS_DIAG
    long long tsum[FT_N] = {0}, tlast = 0, nprob = 0;
    const long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const float* pb = reinterpret_cast<const float*>(lds + L_PB);
    const float* pbq = reinterpret_cast<const float*>(lds + L_PBQ);
    if (MLP) for (int i = t_; i < FB_END; i += 512) reinterpret_cast<float*>(lds + L_PB)[i] = g.pb[i];
    if (QKV) for (int i = t_; i < FB_1; i += 512) reinterpret_cast<float*>(lds + L_PBQ)[i] = g.pbq[i];
    // ---- gather: 72 fragments per operand half, 9 (QKV alone) or 18 a wave: column tile w & 3 of the tile whose column (p_, t_) this
    //      lane fetches.  Issued for tile i + 1 as soon as every wave has read the last LDS operand of tile i - BEFORE that tile's last
    //      epilogue, whose conversions and stores then run under the DMA's flight ----------------------------------------------------
    auto gather = [&](int p_, int t_) {
        const int wave = wave0, lane = lane0, gq = lane >> 4, myct = wave & 3;
#pragma unroll 2
        for (int i = 0; i < (MLP ? 18 : 9); ++i) {
            const int idx = wave + 8 * i, part = idx >= 72 ? 1 : 0, r = idx - 72 * part;
            const char* img = part ? g.tf_att : g.tf_x;
            // (fragment r = (k-step, plane, column tile r & 3): wave w's are all of column tile w & 3)
            if (r < 64) {
                const int ks = r >> 3, plane = (r >> 2) & 1;
                const char* src = img + (int64_t)p_ * TF_BYTES + (2 * ks + plane) * TFB + img_tok_off(false, t_, gq);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lds + part * MT_HALF + ks * 8192 + plane * 4096 + myct * 1024), 16, 0, 0);
            } else {
                const int plane = (r - 64) >> 2;
                const char* src = img + (int64_t)p_ * TF_BYTES + TF_MAIN + plane * TFR + img_tok_off(true, t_, 0);
                if (lane < 16)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(lds + part * MT_HALF + 65536 + plane * 1024 + myct * 256), 16, 0, 0);
            }
        }
    };
    // the column (problem, token) this wave's lanes gather for tile tl: column tile w & 3
    auto gather_column = [&](int64_t tl, int& pi, int& tk) {
        const int j = lane0 & 15, ct = wave0 & 3;
        int64_t qi;
        if (tl < nF) { qi = tl >> 1; tk = 64 * (int)(tl & 1) + 16 * ct + j; }
        else if (tl < nF + nE) { qi = 4 * (tl - nF) + ct; tk = 128 + j; }
        else { qi = 64 * (tl - nF - nE) + 16 * ct + j; tk = 144; }
        if (qi >= NQ) qi = NQ - 1;
        pi = (int)(qi < L ? qi : g.half + (qi - L));
    };
    // the next tile's operands: called by every wave behind the barrier that ends the tile's last LDS read
    auto gather_next = [&](int64_t tile) {
        const int64_t tn = tile + gridDim.x;
        if (tn < ntile) {
            int pn, tkn;
            gather_column(tn, pn, tkn);
            gather(pn, tkn);
        }
    };
    // A lane with nothing to store stores to `sink` instead of branching around the instruction: behind a skipped-or-not branch the
    // compiler's `s_waitcnt vmcnt(n)` for a LOAD issued before the stores must assume they were skipped - n = 0, and the first
    // weight fragments of the next product waited for every store of the epilogue in front of it to be acknowledged.
    char* const sink = g.dump;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int lane = lane0, wave = wave0;
        const h8v* pw = g.pw;
        const h8v* pwq = g.pwq;
        asm volatile("" : "+v"(lane), "+s"(wave), "+s"(pw), "+s"(pwq));       // (nothing hoisted out of the tile loop: see the attention kernel)
        const int gq = lane >> 4, j = lane & 15;
        const int cls = tile < nF ? 0 : tile < nF + nE ? 1 : 2;
        // this lane's four columns (one per column tile): problem and token
        int pidx[4], tok[4];
        bool colok[4];
        auto column = [&](int64_t tl, int ct, int& pi, int& tk, bool& ok) {
            int64_t qi;
            if (cls == 0) { qi = tl >> 1; tk = 64 * (int)(tl & 1) + 16 * ct + j; }
            else if (cls == 1) { qi = 4 * (tl - nF) + ct; tk = 128 + j; }
            else { qi = 64 * (tl - nF - nE) + 16 * ct + j; tk = 144; }
            ok = qi < NQ;
            if (qi >= NQ) qi = NQ - 1;
            pi = (int)(qi < L ? qi : g.half + (qi - L));
        };
        // the column tile this wave gathers and owns the ragged unit of: w & 3 (computed, not selected out of the arrays above - a
        // runtime index would put them in scratch memory)
        const int myct = wave & 3;
        int my_p, my_t;
        bool my_ok;
        auto remap = [&]() {
            int64_t tl = tile;
            asm volatile("" : "+s"(tl));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) column(tl, ct, pidx[ct], tok[ct], colok[ct]);
            column(tl, myct, my_p, my_t, my_ok);
        };
        remap();
#ifdef PATS_DIAG
        tlast = __builtin_amdgcn_s_memrealtime();
#endif
        if (tile == (int64_t)blockIdx.x) {                 // (every later tile's gather went out behind the last product of the tile before)
            wg_barrier();
            gather(my_p, my_t);
        }
        TT(12);
        // the weight fragments of the tile's first k-step go out behind the gather (and every later product's before the epilogue
        // in front of it)
        const int rag_tile = wave < 4 ? 16 : 33;
        W5 W0;
        h8v a5[2][5][2];
        if (MLP) {
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const int mt = m < 2 ? 2 * wave + m : m < 4 ? 17 + 2 * wave + (m - 2) : rag_tile;
                W0.w[m] = (const h8v*)uniform_ptr(pw + FW_1 + (size_t)mt * 18 * FR);
            }
        } else {
            W0.w[0] = (const h8v*)uniform_ptr(pwq + FW_Q + (size_t)(2 * wave) * 9 * FR);
            W0.w[1] = (const h8v*)uniform_ptr(pwq + FW_Q + (size_t)(2 * wave + 1) * 9 * FR);
            W0.w[2] = (const h8v*)uniform_ptr(pwq + FW_K + (size_t)(2 * wave) * 9 * FR);
            W0.w[3] = (const h8v*)uniform_ptr(pwq + FW_K + (size_t)(2 * wave + 1) * 9 * FR);
            W0.w[4] = (const h8v*)uniform_ptr(pwq + (wave < 4 ? FW_Q : FW_K) + (size_t)16 * 9 * FR);
        }
        wload<5>(W0.w, 0, lane, a5[0]);
        TT(13);
        wg_barrier_global();
        TT(0);
        h8v a3[2][3][2];
        if (MLP) {
            // ---- mlp[0]: row tiles 2 w, 2 w + 1 of both halves (FW_1 numbering: half 1 starts at tile 17) + one unit of a ragged tile ----
            const int rag_ct = wave & 3;
            f4v acc[4][4], accr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[m][ct] = f4v{0.f, 0.f, 0.f, 0.f};
            prod5<18>(W0, a5, lds, lane, rag_ct, acc, accr, wave >= 4);
            W3 W2;
            W2.w[0] = (const h8v*)uniform_ptr(pw + FW_2 + (size_t)(2 * wave) * 18 * FR);
            W2.w[1] = (const h8v*)uniform_ptr(pw + FW_2 + (size_t)(2 * wave + 1) * 18 * FR);
            W2.w[2] = (const h8v*)uniform_ptr(pw + FW_2 + (size_t)16 * 18 * FR);
            wload<3>(W2.w, 0, lane, a3[0]);
            TT(1);
            // ---- the residual of this wave's mlp[3] units, from the x fragments still in LDS: rows 16 mt + 4 g.. of its four columns ------
            f4v res[2][4], resr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) res[m][ct] = f4v{0.f, 0.f, 0.f, 0.f};
            if (g.residual) {
                auto rd = [&](int mt, int ct) {
                    const char* bx = lds + lds_piece_off(mt, ct, 0, j) + (mt < 16 ? (gq >> 1) * 256 : 0) + (gq & 1) * 8;
                    const int pl = mt < 16 ? 4096 : 1024;
                    const h4v hi = *reinterpret_cast<const h4v*>(bx), lo = *reinterpret_cast<const h4v*>(bx + pl);
                    return (__builtin_convertvector(hi, f4v) + __builtin_convertvector(lo, f4v)) * (1.0f / PRE);
                };
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) res[m][ct] = rd(2 * wave + m, ct);
                if (wave < 4 && gq < 2) resr = rd(16, wave);
            }
            TT(9);
            wg_barrier();                                  // every wave is done with x | att: hidden takes their place
            TT(10);
            {
                auto put = [&](int mtl, int hf, int ct, const f4v acc_, bool ragged) {
                    const int ch = hf * 272 + 16 * mtl + 4 * gq;
                    const f4v bias = load4(pb + FB_1 + ch), sc = load4(pb + FB_A + ch), sh = load4(pb + FB_S + ch);
                    f4v v = fma4(acc_, sc * (UNS * PRE), (bias * sc + sh) * PRE);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];              // ReLU that keeps NaN
                    const u4v piece = make_piece(v);
                    if (!ragged || gq < 2) *reinterpret_cast<u4v*>(lds + hf * MT_HALF + lds_piece_off(mtl, ct, gq, j)) = piece;
                };
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) put(2 * wave + (m & 1), m >> 1, ct, acc[m][ct], false);
                put(16, wave >> 2, rag_ct, accr, true);
            }
            TT(11);
            wg_barrier();
            TT(2);
            // ---- mlp[3]: row tiles 2 w, 2 w + 1 over the four column tiles + (waves 0..3) column tile w of the ragged 17th -------------
            f4v o[2][4], orr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) o[m][ct] = f4v{0.f, 0.f, 0.f, 0.f};
            prod3<18, false>(W2, a3, lds, lane, wave < 4 ? wave : -1, o, orr, wave >= 4);
            if (QKV) {
                W0.w[0] = (const h8v*)uniform_ptr(pwq + FW_Q + (size_t)(2 * wave) * 9 * FR);
                W0.w[1] = (const h8v*)uniform_ptr(pwq + FW_Q + (size_t)(2 * wave + 1) * 9 * FR);
                W0.w[2] = (const h8v*)uniform_ptr(pwq + FW_K + (size_t)(2 * wave) * 9 * FR);
                W0.w[3] = (const h8v*)uniform_ptr(pwq + FW_K + (size_t)(2 * wave + 1) * 9 * FR);
                W0.w[4] = (const h8v*)uniform_ptr(pwq + (wave < 4 ? FW_Q : FW_K) + (size_t)16 * 9 * FR);
                wload<5>(W0.w, 0, lane, a5[0]);
            }
            TT(3);
            wg_barrier();                                  // every wave is done with the hidden fragments: the output tile takes their place
            if (!QKV) gather_next(tile);                   // ... or the next tile's operands, under this epilogue
            // ---- out = . + b2 + x -> the next layer's images (16-byte pieces, lane pairs exchange halves) [-> LDS: the QKV phase's operand] --
            {
                auto emit = [&](int mt, int ct, int cp, int ctk, bool cok, const f4v acc_, const f4v res_, bool mine) {
                    const f4v bias = load4(pb + FB_2 + 16 * mt + 4 * gq);
                    const f4v v = fma4(acc_, bcast4(UNS), bias) + res_;
                    const bool ok = cok && mine && (mt < 16 || gq < 2);
                    if (ok) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) bad |= !(fabsf(v[r]) <= 3.0e38f);
                    }
                    const u4v piece = make_piece(v * PRE);
                    // with a QKV phase behind it the tile goes to LDS only and leaves for global memory at the END of the tile (out_store):
                    // a load issued behind a store completes behind it (one in-order queue), so stores issued here would cost the q, k
                    // product's first weight fragments a far-memory write acknowledgement (~2 us)
                    char* d = g.tf_out + (int64_t)cp * TF_BYTES + img_piece_off(mt, ctk, gq);
                    if (!QKV) *reinterpret_cast<u4v*>(ok ? d : sink) = piece;
                    if (QKV && mine && (mt < 16 || gq < 2)) *reinterpret_cast<u4v*>(lds + lds_piece_off(mt, ct, gq, j)) = piece;
                };
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) emit(2 * wave + m, ct, pidx[ct], tok[ct], colok[ct], o[m][ct], res[m][ct], true);
                emit(16, myct, my_p, my_t, my_ok, orr, resr, wave < 4);
            }
            if (QKV) wg_barrier();
            TT(4);
        }
        if (QKV) {
            // ---- q, k: row tiles 2 w, 2 w + 1 of each + one unit of a ragged tile (waves 0..3: q's, 4..7: k's) -------------------------------
            W3 WV;
            WV.w[0] = (const h8v*)uniform_ptr(pwq + FW_V + (size_t)(2 * wave) * 9 * FR);
            WV.w[1] = (const h8v*)uniform_ptr(pwq + FW_V + (size_t)(2 * wave + 1) * 9 * FR);
            WV.w[2] = (const h8v*)uniform_ptr(pwq + FW_V + (size_t)16 * 9 * FR);
            {
                const int rag_ct = wave & 3;
                f4v acc[4][4], accr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[m][ct] = f4v{0.f, 0.f, 0.f, 0.f};
                prod5<9>(W0, a5, lds, lane, rag_ct, acc, accr, wave >= 4);
                wload<3>(WV.w, 0, lane, a3[0]);
                TT(5);
                // q, k leave BEHIND the v^T product (unit u = row tile u >> 2 of the wave's four, column tile u & 3; unit 16 = the
                // extras): see out_store.  (Slices of this epilogue under the k-steps of v^T - vector work while its MFMAs occupy the
                // pipe - made that product 16 instead of 5.5 us: every k-step's weights waited behind the slice's stores.)
                auto qk_unit = [&](int u) {
                    if (u < 16) {
                        const int m = u >> 2, ct = u & 3;
                        const int mt = 2 * wave + (m & 1);
                        const f4v bias = load4(pbq + (m < 2 ? FB_Q : FB_K) + 16 * mt + 4 * gq) * PRE;
                        const bool want = m < 2 ? g.want_q != 0 : g.want_kv != 0;
                        const u4v piece = make_piece(fma4(acc[m][ct], bcast4(UNS * PRE), bias));
                        char* d = g.qkv + (int64_t)pidx[ct] * QKV_BYTES + (m < 2 ? QO_Q : QO_K) + img_piece_off(mt, tok[ct], gq);
                        *reinterpret_cast<u4v*>(want && colok[ct] ? d : sink) = piece;
                    } else {
                        // extras: rows 256 + 4 g + r = (head 2 g + (r >> 1), channel 64 + (r & 1)); k as the A operand (h0 h1 l0 l1 h0 h1 0 0),
                        // q as B (h0 h1 h0 h1 l0 l1 0 0): ONE more MFMA per key tile gives hi.hi + lo.hi + hi.lo of both channels
                        const bool isq = wave < 4;
                        const f4v bias = load4(pbq + (isq ? FB_Q : FB_K) + 256 + 4 * (gq & 1)) * PRE;
                        h4v hi, lo;
                        split4_pre(fma4(accr, bcast4(UNS * PRE), bias), hi, lo);
                        char* d = g.qkv + (int64_t)my_p * QKV_BYTES + (isq ? QO_QX : QO_KX) + ((2 * gq) * FN + my_t) * 16;
                        const bool st_ = (isq ? g.want_q != 0 : g.want_kv != 0) && my_ok && gq < 2;
                        const h8v p0 = isq ? h8v{hi.x, hi.y, hi.x, hi.y, lo.x, lo.y, 0, 0} : h8v{hi.x, hi.y, lo.x, lo.y, hi.x, hi.y, 0, 0};
                        const h8v p1 = isq ? h8v{hi.z, hi.w, hi.z, hi.w, lo.z, lo.w, 0, 0} : h8v{hi.z, hi.w, lo.z, lo.w, hi.z, hi.w, 0, 0};
                        *reinterpret_cast<h8v*>(st_ ? d : sink) = p0;
                        *reinterpret_cast<h8v*>(st_ ? d + FN * 16 : sink + 16) = p1;
                    }
                };
                // ---- v^T: channel tiles 2 w, 2 w + 1 (+ the ragged 17th for column tile w, waves 0..3); rows = the column tile's tokens ------
                f4v o[2][4], orr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) o[m][ct] = f4v{0.f, 0.f, 0.f, 0.f};
                prod3<9, true>(WV, a3, lds, lane, wave < 4 ? wave : -1, o, orr, wave >= 4);
                TT(7);
                // Every store of the tile from here on, in one burst in front of the next tile's gather wait: the output tile (read
                // back from LDS - each lane its own pieces, no barrier needed), then - behind the barrier that frees the LDS and the
                // next tile's gather - q, k and v^T.  Their acknowledgements and the DMA fly together.
                if (MLP) {
                    auto out_store = [&](int mt, int ct, int cp, int ctk, bool cok, bool mine) {
                        const bool ok = cok && mine && (mt < 16 || gq < 2);
                        const u4v piece = *reinterpret_cast<const u4v*>(lds + lds_piece_off(mt, ct, gq, j));
                        char* d = g.tf_out + (int64_t)cp * TF_BYTES + img_piece_off(mt, ctk, gq);
                        *reinterpret_cast<u4v*>(ok ? d : sink) = piece;
                    };
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) out_store(2 * wave + m, ct, pidx[ct], tok[ct], colok[ct], true);
                    out_store(16, myct, my_p, my_t, my_ok, wave < 4);
                }
                wg_barrier();                              // every wave has read the tile's last operand
                gather_next(tile);
#pragma unroll
                for (int u = 0; u < 17; ++u) qk_unit(u);
                TT(6);
                TT(7);
                if (g.want_kv) {
                    // A fragment (channel tile mt, key pair tile kk) of a problem: lane (g, j = channel) holds keys 32 kk + 4 g + r (first
                    // half) and 32 kk + 16 + 4 g + r (second half), hi at +0, lo at +1024
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        if (m == 2 && wave >= 4) break;
                        const int mt = m < 2 ? 2 * wave + m : 16;
                        const f4v bias = bcast4(pbq[FB_V + 16 * mt + j] * PRE);
                        if (cls == 0) {
                            char* vb = g.qkv + (int64_t)pidx[0] * QKV_BYTES + QO_V + mt * V_TILE + lane * 16;
                            if (m < 2) {
#pragma unroll
                                for (int pp = 0; pp < 2; ++pp) {
                                    h4v ah, al, bh, bl;
                                    split4_pre(fma4(o[m][2 * pp], bcast4(UNS * PRE), bias), ah, al);
                                    split4_pre(fma4(o[m][2 * pp + 1], bcast4(UNS * PRE), bias), bh, bl);
                                    char* d = vb + (2 * (int)(tile & 1) + pp) * 2048;
                                    if (colok[0]) {
                                        *reinterpret_cast<h8v*>(d) = h8v{ah.x, ah.y, ah.z, ah.w, bh.x, bh.y, bh.z, bh.w};
                                        *reinterpret_cast<h8v*>(d + 1024) = h8v{al.x, al.y, al.z, al.w, bl.x, bl.y, bl.z, bl.w};
                                    }
                                }
                            } else {                           // the ragged unit's pair partner is another wave's: 8-byte halves
                                h4v ah, al;
                                split4_pre(fma4(orr, bcast4(UNS * PRE), bias), ah, al);
                                char* d = vb + (2 * (int)(tile & 1) + (wave >> 1)) * 2048 + (wave & 1) * 8;
                                if (colok[0]) {
                                    *reinterpret_cast<h4v*>(d) = ah;
                                    *reinterpret_cast<h4v*>(d + 1024) = al;
                                }
                            }
                        } else if (cls == 1) {                 // token tile 8 = first half of key pair tile 4; its second half is token 144 (class T)
#pragma unroll                                                 // and zeros: lanes g > 0 write them, lane group 0 leaves its second half to class T
                            for (int ct = 0; ct < 4; ++ct) {
                                if (m == 2 && ct != wave) continue;
                                h4v ah, al;
                                split4_pre(fma4(m < 2 ? o[m < 2 ? m : 0][ct] : orr, bcast4(UNS * PRE), bias), ah, al);
                                char* d = g.qkv + (int64_t)pidx[ct] * QKV_BYTES + QO_V + mt * V_TILE + 4 * 2048 + lane * 16;
                                if (colok[ct]) {
                                    if (gq == 0) {
                                        *reinterpret_cast<h4v*>(d) = ah;
                                        *reinterpret_cast<h4v*>(d + 1024) = al;
                                    } else {
                                        *reinterpret_cast<h8v*>(d) = h8v{ah.x, ah.y, ah.z, ah.w, 0, 0, 0, 0};
                                        *reinterpret_cast<h8v*>(d + 1024) = h8v{al.x, al.y, al.z, al.w, 0, 0, 0, 0};
                                    }
                                }
                            }
                        } else {                               // token 144 of problems 64 n + 16 ct + 4 g + r: (key 144, 0, 0, 0) of lane (0, j)
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) {
                                if (m == 2 && ct != wave) continue;
                                h4v ah, al;
                                split4_pre(fma4(m < 2 ? o[m < 2 ? m : 0][ct] : orr, bcast4(UNS * PRE), bias), ah, al);
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int64_t qi = 64 * (tile - nF - nE) + 16 * ct + 4 * gq + r;
                                    const int64_t pr = qi < L ? qi : g.half + (qi - L);
                                    char* d = g.qkv + pr * QKV_BYTES + QO_V + mt * V_TILE + 4 * 2048 + j * 16 + 8;
                                    if (qi < NQ) {
                                        *reinterpret_cast<h4v*>(d) = h4v{ah[r], 0, 0, 0};
                                        *reinterpret_cast<h4v*>(d + 1024) = h4v{al[r], 0, 0, 0};
                                    }
                                }
                            }
                        }
                    }
                }
                TT(8);
            }
        }
#ifdef PATS_DIAG
        ++nprob;
#endif
    }
    if (bad) atomicOr(g.flag, 1);
#ifdef PATS_DIAG
    tsum[20] = __builtin_amdgcn_s_memtime() - clk0;       // shader clocks / 100 MHz ticks: the clock the workgroup ran at
    tsum[21] = __builtin_amdgcn_s_memrealtime() - rt0;
    if (g.tl && t_ == 0) {
        for (int k = 0; k < FT_N - 1; ++k) g.tl[(size_t)blockIdx.x * FT_N + k] = tsum[k];
        g.tl[(size_t)blockIdx.x * FT_N + FT_N - 1] = nprob;
    }
#endif
}

// The same store through a buffer descriptor of the problem's image: a lane that has nothing to write takes an offset past the
// descriptor's range and the hardware drops its part - no branch around the instruction, so the number of stores a wave has in
// flight is a compile-time fact and a later `s_waitcnt vmcnt(n)` for a LOAD issued before them waits for that load alone (behind a
// skipped-or-not branch the compiler must assume the stores were skipped and wait for the whole queue).
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(char* img) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, TF_BYTES, 0x00020000);
}
__device__ __forceinline__ void store_tf_buf(__amdgpu_buffer_rsrc_t rs, int mt, int t, const f4v v, int lane, const LaneOff& lo_) {
    const int g = lane >> 4, j = lane & 15;
    const u4v piece = make_piece(v);
    unsigned off;
    if (mt < 16) {
        const unsigned d = (mt >> 1) * (2 * TFB) + (mt & 1) * (t < 9 ? 512 : 32) + (t < 9 ? t * 1024 : 9216);
        off = t < 9 ? d + lo_.tf0 : (j == 0 ? d + lo_.tf1 : BUF_OOB);
    } else {
        const unsigned d = TF_MAIN + (t < 9 ? t * 256 : 2304);
        off = (g < 2 && (t < 9 || j == 0)) ? d + lo_.tf2 : BUF_OOB;
    }
    __builtin_amdgcn_raw_buffer_store_b128(piece, rs, (int)off, 0, 0);
}

// ---- the attention core per problem ------------------------------------------------------------------------------------------------
// One persistent 512-thread workgroup per CU owns a problem at a time, its eight waves in ROLES:
//   waves 0..3  PAIR units: query tiles 2 w, 2 w + 1 of the head - every K and V fragment read from LDS once for 32 queries (a
//               one-tile unit reads 100 KB of fragments for 145 MFMAs: eight of them a round made the units LDS-read-bound),
//   waves 6, 7  SINGLE units: query tiles 8 and 9 (token 144 alone),
//   wave 4      stages K_h (+ its extras; double-buffered, one head ahead) and, once a problem, the extras tile of v,
//   wave 5      stages V_h (single buffer, under the scores of head h).
// The staging waves issue nothing but LDS DMA and the computing waves none, so a staging wave's `s_waitcnt vmcnt(0)` waits for its
// fill alone and the computing waves' queues hold only their query loads and output stores (which nobody waits for).
// Two barriers a head: X_h (K_h has landed; every wave is done with V_(h-1)) - scores S^T = K^T Q with the keys as rows, softmax
// in-lane + two exchanges - Y_h (V_h has landed; every wave is done with K_h) - out^T = V P^T with the accumulators of key tiles
// 2 kk, 2 kk + 1 AS the B operand.  The output leaves as a TF image in the folded mlp[0]'s channel order.  A head is one round:
// SIMDs 2, 3 carry 435 MFMAs of it (pair + single), SIMDs 0, 1 290 beside their staging wave.
struct AttnArgs {
    const char* qkv; char* tf_att;
    int64_t P, shift;
    const int* gate;
    const int64_t* live; int64_t live_off, half;
    int sets, stagger;
#ifdef PATS_DIAG
    long long* tl;
#endif
};
constexpr int A_K = 0, A_V = 2 * KH_BYTES, A_VX = A_V + 4 * V_TILE, ATT_LDS = A_VX + 2 * V_TILE;

// one wave's LDS DMA of BYTES (whole 16-byte pieces) - completion: the issuing wave's vmcnt
template <int BYTES>
__device__ __forceinline__ void dma_wave(char* lds_dst, const char* src, int lane) {
    static_assert(BYTES % 16 == 0, "whole 16-byte pieces");
    constexpr int PIECES = BYTES / 16, FULL = PIECES / 64, REST = PIECES - 64 * FULL;
    const char* s = src + lane * 16;
    char* d = lds_dst;
#pragma unroll 1
    for (int r = 0; r < FULL; ++r) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        s += 1024;
        d += 1024;
    }
    if (REST > 0 && lane < REST)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
}

struct QF { h8v h0, l0, h1, l1, x; };      // a query tile of one head as the B operand: two k-steps (hi, lo) + the packed extras
__device__ __forceinline__ void qload(const char* qb, int h, int qt, int lane, QF& q) {
    const int qoff = tf_off(false, qt, lane);
    const int qtok = qt < 9 ? 16 * qt + (lane & 15) : 144;
    q.h0 = *reinterpret_cast<const h8v*>(qb + QO_Q + tf_blk(2 * h, 0) + qoff);
    q.l0 = *reinterpret_cast<const h8v*>(qb + QO_Q + tf_blk(2 * h, 1) + qoff);
    q.h1 = *reinterpret_cast<const h8v*>(qb + QO_Q + tf_blk(2 * h + 1, 0) + qoff);
    q.l1 = *reinterpret_cast<const h8v*>(qb + QO_Q + tf_blk(2 * h + 1, 1) + qoff);
    q.x = *reinterpret_cast<const h8v*>(qb + QO_QX + (h * FN + qtok) * 16);
}

__global__ void __launch_bounds__(512, 1)
gnn_fine_attn_kernel(AttnArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (g.gate && *g.gate == 0) return;
    const int t = threadIdx.x, lane0 = t & 63, wave0 = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int i = (int)(blockIdx.x >> 3 & 31) * g.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(32);       // ~1 us a step
#ifdef PATS_DIAG
    long long tsum[FT_N] = {0}, tlast = 0, nprob = 0;
#endif
    int64_t L = g.half;                                   // live rows per descriptor set
    if (g.live) { const int64_t l_ = *g.live - g.live_off; L = l_ < 0 ? 0 : (l_ < g.half ? l_ : g.half); }
    const int64_t NQ = L * g.sets;                         // live problems: q -> problem (q / L) * half + q % L
    auto problem_of = [&](int64_t q_) { return q_ < L ? q_ : g.half + (q_ - L); };
    auto kv_block = [&](int64_t q_) {
        int64_t ps = problem_of(q_) + g.shift;
        if (ps >= g.P) ps -= g.P;
        return g.qkv + ps * QKV_BYTES;
    };
    if (wave0 == 4) {
        // ---- K_h one head ahead (buffer h & 1), the next problem's extras tile of v under head 3 -------------------------------
        int vxsel = 0;
        bool first = true;
        for (int64_t q_ = blockIdx.x; q_ < NQ; q_ += gridDim.x) {
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            const char* kvb = kv_block(q_);
            const bool has_next = q_ + gridDim.x < NQ;
            const char* kvn = has_next ? kv_block(q_ + gridDim.x) : kvb;
            if (first) {
                dma_wave<4 * TFB>(lds + A_K, kvb + QO_K, lane);
                dma_wave<FN * 16>(lds + A_K + 4 * TFB, kvb + QO_KX, lane);
                dma_wave<V_TILE>(lds + A_VX + vxsel * V_TILE, kvb + QO_V + 16 * V_TILE, lane);
            }
            first = false;
            for (int h = 0; h < 4; ++h) {
                wg_dma_landed();                           // X_h: K_h (and, at h = 0, the extras tile) are in LDS
                char* kn = lds + A_K + ((h + 1) & 1) * KH_BYTES;
                if (h < 3) {
                    dma_wave<4 * TFB>(kn, kvb + QO_K + (h + 1) * 4 * TFB, lane);
                    dma_wave<FN * 16>(kn + 4 * TFB, kvb + QO_KX + (h + 1) * FN * 16, lane);
                } else if (has_next) {
                    dma_wave<4 * TFB>(kn, kvn + QO_K, lane);
                    dma_wave<FN * 16>(kn + 4 * TFB, kvn + QO_KX, lane);
                    dma_wave<V_TILE>(lds + A_VX + (1 - vxsel) * V_TILE, kvn + QO_V + 16 * V_TILE, lane);
                }
                wg_dma_arrive();                           // Y_h
            }
            vxsel = 1 - vxsel;
        }
        return;
    }
    if (wave0 == 5) {
        // ---- V_h under the scores of head h --------------------------------------------------------------------------------------
        for (int64_t q_ = blockIdx.x; q_ < NQ; q_ += gridDim.x) {
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            const char* kvb = kv_block(q_);
            for (int h = 0; h < 4; ++h) {
                wg_dma_arrive();                           // X_h: every wave is done with V_(h-1)
                dma_wave<4 * V_TILE>(lds + A_V, kvb + QO_V + h * 4 * V_TILE, lane);
                wg_dma_landed();                           // Y_h
            }
        }
        return;
    }
    auto compute = [&](auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        QF q[NT];
        bool first = true;
        int vxsel = 0;
        for (int64_t q_ = blockIdx.x; q_ < NQ; q_ += gridDim.x) {
            const int64_t p = problem_of(q_);
            // Everything the epilogues address is invariant over the problem loop: left alone the compiler hoists the store addresses
            // and fragment bases out of it and spills the accumulators around them.  An opaque copy of the lane index per problem.
            int lane = lane0, wave = wave0;
            asm volatile("" : "+v"(lane), "+s"(wave));
            const int gq = lane >> 4, j = lane & 15;
            const LaneOff lo_ = lane_offsets(lane);
            const char* qb = g.qkv + p * QKV_BYTES;
            const bool has_next = q_ + gridDim.x < NQ;
            const char* qbn = g.qkv + problem_of(has_next ? q_ + gridDim.x : q_) * QKV_BYTES;
            const __amdgpu_buffer_rsrc_t att_rs = image_rsrc(g.tf_att + p * TF_BYTES);
            int qt[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) qt[n] = NT == 2 ? 2 * wave + n : wave + 2;
#ifdef PATS_DIAG
            tlast = __builtin_amdgcn_s_memrealtime();
#endif
            if (first) {
#pragma unroll
                for (int n = 0; n < NT; ++n) qload(qb, 0, qt[n], lane, q[n]);
            }
            first = false;
            const char* vx = lds + A_VX + vxsel * V_TILE;
            for (int h = 0; h < 4; ++h) {
                wg_barrier();                              // X_h
                FT(9);
                const char* kb = lds + A_K + (h & 1) * KH_BYTES;
                h8v ph[NT][5], pl[NT][5];
                float osc[NT];
                {
                    // ---- scores: rows = keys 16 kt + 4 g + r, column = query; the key fragments of tile kt + 1 are read while the
                    //      MFMAs of tile kt run (two register sets) ---------------------------------------------------------------
                    f4v st[NT][FNT];
                    float mx[NT];
                    h8v qx[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) { qx[n] = gq == 0 ? q[n].x : zero8(); mx[n] = -INFINITY; }
                    const float c = UNS * 0.12309149097933272f * LOG2E;          // accumulator -> exponent of 2: 2^-12 / sqrt(66) * log2(e)
                    struct KF { h8v h0, l0, h1, l1, x; };
                    auto kload = [&](int kt, KF& k) {
                        const int koff = tf_off(false, kt, lane);
                        k.h0 = *reinterpret_cast<const h8v*>(kb + koff);
                        k.l0 = *reinterpret_cast<const h8v*>(kb + TFB + koff);
                        k.h1 = *reinterpret_cast<const h8v*>(kb + 2 * TFB + koff);
                        k.l1 = *reinterpret_cast<const h8v*>(kb + 3 * TFB + koff);
                        k.x = *reinterpret_cast<const h8v*>(kb + 4 * TFB + (kt < 9 ? 16 * kt + j : 144) * 16);
                    };
                    KF kf[2];
                    kload(0, kf[0]);
#pragma unroll
                    for (int kt = 0; kt < FNT; ++kt) {
                        if (kt + 1 < FNT) kload(kt + 1, kf[(kt + 1) & 1]);
                        const KF& k = kf[kt & 1];
                        const h8v kx = gq == 0 ? k.x : zero8();
                        f4v s[NT];
#pragma unroll
                        for (int n = 0; n < NT; ++n) s[n] = mfma3(k.h0, k.l0, q[n].h0, q[n].l0, f4v{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int n = 0; n < NT; ++n) s[n] = mfma3(k.h1, k.l1, q[n].h1, q[n].l1, s[n]);
#pragma unroll
                        for (int n = 0; n < NT; ++n) s[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kx, qx[n], s[n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float v = s[n][r] * c;
                                if (kt == 9 && (gq != 0 || r != 0)) v = -INFINITY;          // keys 145..: not there
                                s[n][r] = v;
                                mx[n] = fmaxf(mx[n], v);
                            }
                            st[n][kt] = s[n];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        float m_ = mx[n];
                        m_ = fmaxf(m_, __shfl_xor(m_, 16));
                        m_ = fmaxf(m_, __shfl_xor(m_, 32));
                        float den = 0.f;
#pragma unroll
                        for (int kt = 0; kt < FNT; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float pr = fast_exp2(st[n][kt][r] - m_);
                                st[n][kt][r] = pr;
                                den += pr;
                            }
                        den += __shfl_xor(den, 16);
                        den += __shfl_xor(den, 32);
                        osc[n] = UNS * PRE * (1.0f / den);
#pragma unroll
                        for (int kk = 0; kk < 5; ++kk) {
                            h4v h0, l0, h1, l1;
                            split4_pre(st[n][2 * kk] * PRE, h0, l0);
                            split4_pre(st[n][2 * kk + 1] * PRE, h1, l1);
                            ph[n][kk] = h8v{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                            pl[n][kk] = h8v{l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                        }
                    }
                }
                FT(10);
                wg_barrier();                              // Y_h
                FT(11);
                // the next head's queries (the next problem's first head under the last): an L2 / Infinity-Cache round trip away
                // (unconditional, from a selected address: behind a branch the loaded registers are copied into q at the join - and
                //  waited for right here, a far-memory round trip at the head of every out phase)
                {
                    const char* qsrc = h < 3 ? qb : qbn;
#pragma unroll
                    for (int n = 0; n < NT; ++n) qload(qsrc, (h + 1) & 3, qt[n], lane, q[n]);
                }
                {
                    // ---- out^T = V P^T: rows = channels, column = query --------------------------------------------------------------
                    h8v vf[2][5][2];
                    auto vload = [&](int dt, h8v (&v)[5][2]) {
                        const char* vb = dt < 4 ? lds + A_V + dt * V_TILE : vx;
#pragma unroll
                        for (int kk = 0; kk < 5; ++kk) {
                            v[kk][0] = *reinterpret_cast<const h8v*>(vb + kk * 2048 + lane * 16);
                            v[kk][1] = *reinterpret_cast<const h8v*>(vb + kk * 2048 + 1024 + lane * 16);
                        }
                    };
                    vload(0, vf[0]);
#pragma unroll
                    for (int dt = 0; dt < 5; ++dt) {
                        if (dt + 1 < 5) vload(dt + 1, vf[(dt + 1) & 1]);
                        f4v o[NT];
#pragma unroll
                        for (int n = 0; n < NT; ++n) o[n] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 5; ++kk)
#pragma unroll
                            for (int n = 0; n < NT; ++n) o[n] = mfma3(vf[dt & 1][kk][0], vf[dt & 1][kk][1], ph[n][kk], pl[n][kk], o[n]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const f4v on = o[n] * osc[n];
                            if (dt < 4) {
                                store_tf_buf(att_rs, 4 * h + dt, qt[n], on, lane, lo_);
                            } else {
                                // rows 2 h', 2 h' + 1 of the extras tile are head h's channels 64, 65 -> bytes 4 h .. of the ragged block's piece
                                const float e0 = (h & 1) ? on.z : on.x, e1 = (h & 1) ? on.w : on.y;
                                const _Float16 a0 = (_Float16)e0, a1 = (_Float16)e1;
                                const h2v hi = {a0, a1}, lo = {(_Float16)(e0 - (float)a0), (_Float16)(e1 - (float)a1)};
                                const bool mine = gq == (h >> 1) && (qt[n] < 9 || j == 0);
                                const unsigned off = mine ? (unsigned)(tf_off(true, qt[n], j) + 4 * h) : BUF_OOB;
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hi), att_rs, (int)(off + tf_blk(8, 0)), 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lo), att_rs, (int)(off + tf_blk(8, 1)), 0, 0);
                            }
                        }
                    }
                }
                FT(12);
            }
            vxsel = 1 - vxsel;
#ifdef PATS_DIAG
            ++nprob;
#endif
        }
    };
    if (wave0 < 4) compute(std::integral_constant<int, 2>{});
    else compute(std::integral_constant<int, 1>{});
#ifdef PATS_DIAG
    if (g.tl && t == 0) {
        for (int k = 0; k < FT_N - 1; ++k) g.tl[(size_t)blockIdx.x * FT_N + k] = tsum[k];
        g.tl[(size_t)blockIdx.x * FT_N + FT_N - 1] = nprob;
    }
#endif
}

// ---- host side --------------------------------------------------------------------------------------------------------------
int fine_layer_supported(int C, int heads, int n, int m) {
    static const bool off = [] { const char* e = env_switch("PATS_GNN_FINE"); return e && atoi(e) == 0; }();
    return !off && C == FC && heads == 4 && n == FN && m == FN;
}
size_t packed_fine_bytes(int C, int heads) {
    return (C == FC && heads == 4) ? (((size_t)FW_END * sizeof(h8v) + (size_t)FB_END * sizeof(float) + 255) & ~(size_t)255) : 0;
}
// w: the layer's weights with w1_t / b1 FOLDED (gnn_fold_kernel) and bn_a / bn_b the eval-mode scale / shift
int launch_fine_pack(const pats_propagation_weights& w, void* section, hipStream_t st) {
    h8v* pw = (h8v*)section;
    float* pb = (float*)(pw + FW_END);
    const int threads = (FW_END / FR) * 64;
    hipLaunchKernelGGL(gnn_fine_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, w, pw, pb);
    return check_launch("gnn_fine_pack_kernel");
}

// CUs of the device (0: the kernels' LDS attributes were refused)
static int fine_cus() {
    struct PerDevice { int state = 0; int n_cu = 256; };
    static PerDevice per_dev[64];
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) { (void)hipGetLastError(); dev_id = 0; }
    PerDevice& pd = per_dev[dev_id];
    if (pd.state == 0) {
        const bool ok = hipFuncSetAttribute((const void*)gnn_fine_tile_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)gnn_fine_tile_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)gnn_fine_tile_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)gnn_fine_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        int v = 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess) (void)hipGetLastError();
        pd.n_cu = v > 0 ? v : 256;
        pd.state = ok ? 1 : -1;
    }
    return pd.state == 1 ? pd.n_cu : 0;
}
size_t fine_scratch_bytes(int64_t P) { return (size_t)P * QKV_BYTES + 256; }      // the per-problem blocks of projections + the sink of masked stores
size_t fine_image_bytes(int64_t P) { return (size_t)P * TF_BYTES; }

int launch_fine_in(const float* x, int64_t P, char* tf, hipStream_t st) {
    const int64_t items = P * 33 * FN;
    hipLaunchKernelGGL(gnn_fine_in_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, x, P, tf);
    return check_launch("gnn_fine_in_kernel");
}
int launch_fine_out(const char* tf, int64_t P, float* y, hipStream_t st, const int64_t* live, int64_t live_off, const float* add) {
    const int64_t items = P * 33 * FN;
    hipLaunchKernelGGL(gnn_fine_out_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, tf, P, y, live, live_off, add);
    return check_launch("gnn_fine_out_kernel");
}

#ifdef PATS_DIAG
static long long* tl_begin(unsigned wgs) {
    if (!diag_env("PATS_FINE_TL")) return nullptr;
    long long* tl = nullptr;
    (void)hipMalloc((void**)&tl, (size_t)wgs * FT_N * 8);
    (void)hipMemset(tl, 0, (size_t)wgs * FT_N * 8);
    return tl;
}
static void tl_end(long long* tl, unsigned wgs, hipStream_t st, const char* what, const char* unit, const char* const* names, int first, int last) {
    if (!tl) return;
    (void)hipStreamSynchronize(st);
    std::vector<long long> h((size_t)wgs * FT_N);
    (void)hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost);
    double sum[FT_N] = {0}, np = 0;
    for (unsigned w = 0; w < wgs; ++w) { for (int k = 0; k < FT_N - 1; ++k) sum[k] += (double)h[(size_t)w * FT_N + k]; np += (double)h[(size_t)w * FT_N + FT_N - 1]; }
    double tot = 0;
    for (int k = first; k <= last; ++k) tot += sum[k];
    fprintf(stderr, "%s (%u workgroups, %.0f %ss; mean us per %s, thread 0): total %.2f\n", what, wgs, np, unit, unit, tot / np / 100.0);
    for (int k = first; k <= last; ++k) if (sum[k] > 0) fprintf(stderr, "  %-38s %7.2f\n", names[k - first], sum[k] / np / 100.0);
    if (sum[21] > 0) fprintf(stderr, "  shader clock over the kernel: %.0f MHz\n", sum[20] / sum[21] * 100.0);
    (void)hipFree(tl);
}
#endif

// The per-token products over the flattened column tiles of P problems (sets descriptor sets of P / sets rows each; live (optional):
// device-side row count of a set, minus live_off).  mlp_section: the layer whose MLP runs on (tf_x, tf_att) -> tf_out (+ x when
// residual != 0), or null; qkv_section: the layer whose q / k / v are taken - of the MLP's output if there is one, else of tf_x - into
// the per-problem blocks at qkv, or null.
static int launch_fine_tile(const char* tf_x, const char* tf_att, char* tf_out, int residual, const void* mlp_section, const void* qkv_section,
                            char* qkv, int want_q, int want_kv, int64_t P, int* flag, const int* gate, hipStream_t st, int sets,
                            const int64_t* live, int64_t live_off) {
    const int cus = fine_cus();
    if (cus <= 0) return PATS_ERR_UNSUPPORTED;
    const h8v* pw = (const h8v*)mlp_section;
    const h8v* pwq = (const h8v*)qkv_section;
    TileArgs a{tf_x, tf_att, tf_out, qkv, pw, pw ? (const float*)(pw + FW_END) : nullptr, pwq, pwq ? (const float*)(pwq + FW_END) : nullptr,
               P, P / sets, sets, live, live_off, flag, residual, want_q, want_kv, qkv + (size_t)P * QKV_BYTES, gate};
    if (!qkv) return PATS_ERR_INVALID;      // (every launch is handed the scratch area: its tail holds the sink of the masked stores)
    const int64_t ntile = 2 * P + (P + 3) / 4 + (P + 63) / 64;
    const unsigned wgs = (unsigned)std::min<int64_t>(ntile, cus);
    // (A start stagger of the workgroups - the attention kernel's de-phasing - was measured here and does nothing: the burst at the
    //  end of a tile is bound by the rate a CU issues its stores, not by what the memory side serves.  DESIGN.md section 7a.)
#ifdef PATS_DIAG
    a.tl = tl_begin(wgs);
#endif
    if (pw && pwq) hipLaunchKernelGGL((gnn_fine_tile_kernel<true, true>), dim3(wgs), dim3(512), MLP_LDS, st, a);
    else if (pw) hipLaunchKernelGGL((gnn_fine_tile_kernel<true, false>), dim3(wgs), dim3(512), MLP_LDS, st, a);
    else hipLaunchKernelGGL((gnn_fine_tile_kernel<false, true>), dim3(wgs), dim3(512), MLP_LDS, st, a);
#ifdef PATS_DIAG
    static const char* names[14] = {"gather: wait", "mlp[0] product", "hidden: barrier behind the LDS writes", "mlp[3] product", "output tile", "q, k product", "q, k out",
                                    "v^T product", "v^T out", "hidden: residual from LDS", "hidden: barrier (operands read)", "hidden: BN, ReLU, split -> LDS",
                                    "tile start: mapping + barrier", "gather: issue + first weights"};
    tl_end(a.tl, wgs, st, pw && pwq ? "gnn_fine_tile_kernel<MLP, QKV>" : pw ? "gnn_fine_tile_kernel<MLP>" : "gnn_fine_tile_kernel<QKV>", "tile", names, 0, 13);
#endif
    return check_launch("gnn_fine_tile_kernel");
}
int launch_fine_qkv(const char* tf, int64_t P, const void* section, char* qkv, int want_q, int want_kv, const int* gate, hipStream_t st, int sets,
                    const int64_t* live, int64_t live_off) {
    return launch_fine_tile(tf, nullptr, nullptr, 0, nullptr, section, qkv, want_q, want_kv, P, nullptr, gate, st, sets, live, live_off);
}
int launch_fine_mlp(const char* tf_x, const char* tf_att, int residual, const void* section, char* tf_out, const void* next_section, char* qkv,
                    int64_t P, int* flag, const int* gate, hipStream_t st, int sets, const int64_t* live, int64_t live_off) {
    return launch_fine_tile(tf_x, tf_att, tf_out, residual, section, next_section, qkv, 1, 1, P, flag, gate, st, sets, live, live_off);
}
// the attention core of P problems: queries of block p, keys / values of block (p + shift) % P -> tf_att
int launch_fine_attn(const char* qkv, int64_t shift, char* tf_att, int64_t P, const int* gate, hipStream_t st, int sets, const int64_t* live,
                     int64_t live_off) {
    const int cus = fine_cus();
    if (cus <= 0) return PATS_ERR_UNSUPPORTED;
    const unsigned wgs = (unsigned)std::min<int64_t>(P, cus);
    // De-phasing (workgroup i starts ((i >> 3) % 32) x stagger us late) paid in the one-kernel layer, whose staging came in bursts that
    // every workgroup of a lockstep grid issued at the same instants.  With a staging wave per operand the fills are spread over the
    // heads: at 4 096 problems the kernel runs 0.556 ms without it and 0.623 with 4 us steps (profiles/r05_gnn_fine_attn_stagger.txt).
    // Off; the diagnostic library still reads PATS_FINE_STAGGER.
    static const int stagger_env = [] { const char* e = diag_env("PATS_FINE_STAGGER"); return e ? atoi(e) : -1; }();
    const int stagger = stagger_env >= 0 ? stagger_env : 0;
    AttnArgs a{qkv, tf_att, P, shift, gate, live, live_off, P / sets, sets, stagger};
#ifdef PATS_DIAG
    a.tl = tl_begin(wgs);
#endif
    hipLaunchKernelGGL(gnn_fine_attn_kernel, dim3(wgs), dim3(512), ATT_LDS, st, a);
#ifdef PATS_DIAG
    static const char* names[4] = {"X: wait for K_h / the out phases", "scores + softmax", "Y: wait for V_h / the scores", "out = V P"};
    tl_end(a.tl, wgs, st, "gnn_fine_attn_kernel", "problem", names, 9, 12);
#endif
    return check_launch("gnn_fine_attn_kernel");
}

// One layer: image p of tf_x with source image (p + shift) % P of tf_s -> tf_out (the single-layer entry; a stack chains the
// launches itself and lets a layer's MLP produce the next layer's projections)
int launch_fine_layer(const char* tf_x, const char* tf_s, int64_t shift, int residual, int64_t P, const void* section,
                      char* tf_out, char* tf_att, char* qkv, int* flag, const int* gate, hipStream_t st, int sets,
                      const int64_t* live, int64_t live_off) {
    int rc;
    if (tf_s == tf_x) {
        if ((rc = launch_fine_qkv(tf_x, P, section, qkv, 1, 1, gate, st, sets, live, live_off))) return rc;
    } else {
        if ((rc = launch_fine_qkv(tf_x, P, section, qkv, 1, 0, gate, st, sets, live, live_off))) return rc;
        if ((rc = launch_fine_qkv(tf_s, P, section, qkv, 0, 1, gate, st, sets, live, live_off))) return rc;
    }
    if ((rc = launch_fine_attn(qkv, shift, tf_att, P, gate, st, sets, live, live_off))) return rc;
    return launch_fine_mlp(tf_x, tf_att, residual, section, tf_out, nullptr, qkv, P, flag, gate, st, sets, live, live_off);
}

}  // namespace pats
