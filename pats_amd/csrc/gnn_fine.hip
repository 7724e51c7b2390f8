// AttentionalPropagation (models/modules.py:107-117) + the residual of AttentionalGNN.forward (:131-133) at the FINE level's shape
// - x, source [b, 264, 145], 4 heads of 66 channels (second_layer.py:44,89 runs 18 such layers on both descriptor sets) - as ONE
// kernel (round 5).  Round 4 ran this layer as six conv_pk_kernel launches around attention145_kernel: 24 tensor passes of 153 KB per
// problem through HBM and 360 KB of packed weights streamed per 64-column tile (5.7 MB per problem through each CU's L1).
//
//   q = Wq x, k = Wk s, v = Wv s;  att = softmax(q^T k / sqrt(66)) v  per head           MultiHeadedAttention.forward :100-105
//   hidden = relu(bn(W1x x + (W1m Wm) att + b1'))                                          (merge folded into mlp[0]: gnn_fold_kernel)
//   out = x + W2 hidden + b2                                                              AttentionalPropagation :114-117, GNN :133
//
// One persistent 512-thread workgroup per CU owns a PROBLEM at a time:
//   * a [264 x 145] tensor split for the fp16 matrix pipe (x 2^6 = hi + lo) in MFMA fragment order is 153 120 bytes ("TF image":
//     per 32-channel k-step and 16-token tile, lane (k / 8, token) holds its 8 channels as one 16-byte piece = the B operand of
//     v_mfma_f32_16x16x32_f16; token 144 and channels 256..263 are ragged blocks without padding) - exactly ONE of them fits the
//     CU's 160 KB of LDS.  So a stage is: DMA one TF image from global memory into LDS (global_load_lds_dwordx4: no registers, no
//     VALU), run a convolution whose OUTPUT lives in the accumulators (wave w: row tiles 2 w, 2 w + 1 and a share of the ragged
//     17th, all ten token tiles = 88 registers), write it - split again, in the next consumer's fragment order - to a per-workgroup
//     scratch block in global memory (L2 / Infinity-Cache resident: written and read back by the same CU), next stage.
//   * the WEIGHTS are the A operand straight from L2 into a register ring (pre-split fragments, one 16-byte load per lane, packed once
//     per layer by gnn_fine_pack_kernel): 2.7 MB per problem per CU instead of 5.7, at a quarter of the rate the L2 sustains
//     (tools/wstream_probe.hip: 115-135 GB/s per CU with every CU streaming).
//   * descriptors travel BETWEEN layers as TF images: the layer's epilogue writes one, the next layer DMAs it and rebuilds the
//     residual from it ((hi + lo) / 2^6: exact to the 22 bits an image holds) - 16-byte accesses everywhere, no 4-byte strided
//     loads (round 4's limiter).  gnn_fine_in_kernel / gnn_fine_out_kernel convert at the ends of a stack.
//   * heads: the reference views a projection as [b, 66, 4, n] (channel = d * 4 + h).  q / k / v rows are permuted at pack time to
//     [head][d < 64] (head h = k-steps 2 h, 2 h + 1 of a TF image) followed by the eight "extra" channels (d = 64, 65 of each head) in
//     the ragged 17th row tile; the folded mlp[0] matrix has its attention columns in the same order.
//   * attention per (head, 16-query tile) as in csrc/attention145.hip: S^T = K^T Q with the keys as rows (softmax in-lane + two
//     exchanges), the accumulators of key tiles 2 kk, 2 kk + 1 ARE the B operand of out^T = V P^T; K_h (double-buffered) and V_h are
//     DMA'd into LDS per head, v is produced TRANSPOSED by its projection (activations as A, weights as B) directly in that A-fragment
//     order.  The two extra channels of a head cost ONE more MFMA per key tile instead of a k-step of three: A = (kh0 kh1 kl0 kl1 kh0
//     kh1 0 0), B = (qh0 qh1 qh0 qh1 ql0 ql1 0 0) gives hi.hi + lo.hi + hi.lo of both channels.
// Stages per problem: s -> [k, v^T]; x -> [q]; attention; att -> [hidden0 (attention part)]; x -> [hidden0 (x part) -> scratch,
// hidden1 (x part)]; att -> [hidden1 -> LDS straight from the accumulators -> out (second half)]; hidden0 -> [out + b2 + x -> TF image].
// BatchNorm in eval mode only (the second layer always is: pats.py:112-114).  Range: |activation| < 1023; a non-finite output raises
// *flag and the round-2 composition queued behind, gated on it, redoes the layer.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>
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
// per-workgroup scratch block (global memory)
constexpr int SC_Q = 0;                       // q as a TF image; the attention output replaces it in place (main part) / fills its ragged block
constexpr int SC_QX = SC_Q + TF_BYTES;
constexpr int SC_K = SC_QX + XB;              // k (TF main part); later hidden[0:264] as a TF image
constexpr int SC_KX = SC_K + TF_BYTES;
constexpr int SC_V = SC_KX + XB;              // v^T fragments; later hidden[264:528] as a TF image
constexpr int SC_BYTES = SC_V + V_BYTES;      // 498 880
// LDS while the attention runs
constexpr int KH_BYTES = 4 * TFB + FN * 16;   // one head of k: two k-steps x (hi, lo) + its extras
constexpr int L_KA = 0, L_V = 2 * KH_BYTES, L_VX = L_V + 4 * V_TILE, L_ATT_END = L_VX + V_TILE;
constexpr int FINE_LDS = TF_BYTES + 32;
static_assert(L_ATT_END <= FINE_LDS, "attention staging fits the slot");
// packed weights (units of h8v): fragment (row tile, k-step) = [hi | lo][64 lanes]
constexpr int FR = 128;
constexpr int FW_Q = 0, FW_K = FW_Q + 17 * 9 * FR, FW_V = FW_K + 17 * 9 * FR, FW_1 = FW_V + 17 * 9 * FR, FW_2 = FW_1 + 34 * 18 * FR,
              FW_END = FW_2 + 17 * 18 * FR;
// biases behind them (floats): q', k', v' (permuted), b1' (two halves), bn scale, bn shift, b2 - every vector padded to 272
constexpr int FB_Q = 0, FB_K = 272, FB_V = 544, FB_1 = 816, FB_A = FB_1 + 544, FB_S = FB_A + 544, FB_2 = FB_S + 544, FB_END = FB_2 + 272;

struct FineArgs {
    const char* tf_x;          // [P] TF images of the descriptors
    const char* tf_s;          // [P] TF images of the sources; problem p reads image (p + shift) % P
    const char* tf_res;        // [P] TF images of the residual or null: added to the output (exact to the 22 bits an image holds)
    char* tf_out;              // [P] TF images of the output (FULL)
    char* tf_att;              // [P] TF images of the attention output (!FULL: the kernel stops behind the attention; gnn_fine_mlp_kernel follows)
    const h8v* pw;
    const float* pb;
    char* scratch;             // [gridDim.x][SC_BYTES]
    int64_t P, shift;
    int* flag;
    const int* gate;           // optional: no-op unless *gate != 0
    const int64_t* live;       // optional device-side ROW count: of every descriptor set (`sets` of them, `half` rows each: P = sets * half)
    int64_t live_off, half;    // only rows < clamp(*live - live_off, 0, half) are problems
    int sets;
    int stagger;               // workgroup i starts ((i >> 3) % 32) * stagger * ~1 us late: de-phases the memory bursts of the stages
#ifdef PATS_DIAG
    long long* tl;             // diagnostic library: FT_N accumulated stage durations per workgroup (thread 0, 10 ns units)
#endif
};

// Diagnostic library only (python -m pats_amd.build --diag, PATS_AMD_DIAG_LIB=1, PATS_FINE_TL=1): s_memrealtime stamps at the stage
// boundaries of thread 0, summed over the workgroup's problems; launch_fine_layer prints the means per problem
constexpr int FT_N = 24;
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

// the inverse: rows 16 mt + 4 g + r of token 16 t + j from a TF image in global memory.  load_tf_piece() issues the one 16-byte load
// of a lane (an even-row lane its pair's hi piece, an odd-row lane the lo piece; zeros where the tile has nothing for it);
// tf_piece_value() - called by all 64 lanes - gives the halves back to their owners and rebuilds (hi + lo) / 2^6.
typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u4v_t load_tf_piece(const char* src, int mt, int t, int lane, const LaneOff& lo_) {
    const int g = lane >> 4, j = lane & 15;
    u4v_t p = {0u, 0u, 0u, 0u};
    if (mt < 16) {
        const char* d = src + (mt >> 1) * (2 * TFB) + (mt & 1) * (t < 9 ? 512 : 32) + (t < 9 ? t * 1024 : 9216);
        if (t < 9) p = *reinterpret_cast<const u4v_t*>(d + lo_.tf0);
        else if (j == 0) p = *reinterpret_cast<const u4v_t*>(d + lo_.tf1);
    } else {
        const char* d = src + TF_MAIN + (t < 9 ? t * 256 : 2304);
        if (g < 2 && (t < 9 || j == 0)) p = *reinterpret_cast<const u4v_t*>(d + lo_.tf2);
    }
    return p;
}
__device__ __forceinline__ f4v tf_piece_value(const u4v_t p) {
    // even-row lane: (x, y) = own hi, (z, w) = partner's hi; odd-row lane: (x, y) = partner's lo, (z, w) = own lo.
    // lane_swap16(a, b): odd rows of a <-> even rows of b  =>  every lane ends with a = own hi, b = own lo
    unsigned ax = p.x, ay = p.y, bx = p.z, by = p.w;
    lane_swap16(ax, bx);
    lane_swap16(ay, by);
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const h4v hi = __builtin_bit_cast(h4v, u2v{ax, ay}), lo = __builtin_bit_cast(h4v, u2v{bx, by});
    return (__builtin_convertvector(hi, f4v) + __builtin_convertvector(lo, f4v)) * (1.0f / PRE);
}

// ---- one convolution pass over the nine k-steps of the TF image in LDS -----------------------------------------------------------
// Wave w accumulates row tiles tb + 2 w, tb + 2 w + 1 (acc[m][t], all ten token tiles) and two units (tr, t0), (tr, t1) of the
// ragged row tile tr = tb + 16 (accr; t1 < 0: none).  TRANS: the activations are the A operand, the weights B (v^T).
struct ARing { h8v a[2][3][2]; };             // [slot][main 0, main 1, ragged][hi, lo]

template <int KSM>
__device__ __forceinline__ void aload(const h8v* __restrict__ W, int tb, int wave, int ks, int lane, h8v (&a)[3][2]) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int mt = m < 2 ? tb + 2 * wave + m : tb + 16;
        gptr_h8 Wf = uniform_ptr(W + ((size_t)mt * KSM + ks) * FR);
        a[m][0] = Wf[lane];
        a[m][1] = Wf[64 + lane];
    }
}

template <bool TRANS, bool RAG>
__device__ __forceinline__ void kstep(const char* slot, int ks, const h8v (&a)[3][2], f4v (&acc)[2][FNT], f4v (&accr)[2], int t0, int t1,
                                      int lane) {
    const char* b0 = slot + tf_blk(RAG ? 8 : ks, 0);
    const char* b1 = slot + tf_blk(RAG ? 8 : ks, 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        h8v bh[5], bl[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            // (ragged k-step: every lane reads the piece of lane & 15 - a valid address - and lanes >= 16 then take zeros: selects
            //  instead of an exec-mask region per load)
            const int off = tf_off(RAG, 5 * half + i, lane);
            bh[i] = *reinterpret_cast<const h8v*>(b0 + off);
            bl[i] = *reinterpret_cast<const h8v*>(b1 + off);
            if (RAG) {
                bh[i] = lane < 16 ? bh[i] : zero8();
                bl[i] = lane < 16 ? bl[i] : zero8();
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int t = 5 * half + i;
            if (TRANS) {
                acc[0][t] = mfma3(bh[i], bl[i], a[0][0], a[0][1], acc[0][t]);
                acc[1][t] = mfma3(bh[i], bl[i], a[1][0], a[1][1], acc[1][t]);
                if (t == t0) accr[0] = mfma3(bh[i], bl[i], a[2][0], a[2][1], accr[0]);
                if (t == t1) accr[1] = mfma3(bh[i], bl[i], a[2][0], a[2][1], accr[1]);
            } else {
                acc[0][t] = mfma3(a[0][0], a[0][1], bh[i], bl[i], acc[0][t]);
                acc[1][t] = mfma3(a[1][0], a[1][1], bh[i], bl[i], acc[1][t]);
                if (t == t0) accr[0] = mfma3(a[2][0], a[2][1], bh[i], bl[i], accr[0]);
                if (t == t1) accr[1] = mfma3(a[2][0], a[2][1], bh[i], bl[i], accr[1]);
            }
        }
        // a half k-step is a closed unit for the scheduler: left alone it hoists the LDS reads of both halves (and of the next
        // k-step) above the MFMAs and spills
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool TRANS, int KSM>
__device__ __forceinline__ void conv_pass(const h8v* __restrict__ W, int tb, int ks0, const char* slot, int wave, int lane,
                                          f4v (&acc)[2][FNT], f4v (&accr)[2], int t0, int t1) {
    ARing r;
    aload<KSM>(W, tb, wave, ks0, lane, r.a[0]);
#pragma unroll 1
    for (int kp = 0; kp < 4; ++kp) {
        aload<KSM>(W, tb, wave, ks0 + 2 * kp + 1, lane, r.a[1]);
        kstep<TRANS, false>(slot, 2 * kp, r.a[0], acc, accr, t0, t1, lane);
        aload<KSM>(W, tb, wave, ks0 + 2 * kp + 2, lane, r.a[0]);
        kstep<TRANS, false>(slot, 2 * kp + 1, r.a[1], acc, accr, t0, t1, lane);
    }
    kstep<TRANS, true>(slot, 8, r.a[0], acc, accr, t0, t1, lane);
}

__device__ __forceinline__ void zero_acc(f4v (&acc)[2][FNT], f4v (&accr)[2]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < FNT; ++t) acc[m][t] = f4v{0.f, 0.f, 0.f, 0.f};
    accr[0] = f4v{0.f, 0.f, 0.f, 0.f};
    accr[1] = f4v{0.f, 0.f, 0.f, 0.f};
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

// ---- the layer -----------------------------------------------------------------------------------------------------------------
// FULL: the whole layer per problem (round 5, first form; diagnostic library only since the split below is faster).  !FULL: q / k / v
// and the attention only - the attention output leaves as a TF image and gnn_fine_mlp_kernel runs the MLP on flattened column tiles.
template <bool FULL>
__global__ void __launch_bounds__(512, 1)
gnn_fine_layer_kernel(FineArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (g.gate && *g.gate == 0) return;
    const int t = threadIdx.x, lane = t & 63, wave0 = __builtin_amdgcn_readfirstlane(t >> 6);
    char* scr0 = g.scratch + (size_t)blockIdx.x * SC_BYTES;
    bool bad = false, first = true;
    const int lane0 = lane;
    for (int i = (int)(blockIdx.x >> 3 & 31) * g.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(32);       // ~1 us a step
#ifdef PATS_DIAG
    long long tsum[FT_N] = {0}, tlast = 0, nprob = 0;
#endif
    int64_t L = g.half;                                   // live rows per descriptor set
    if (g.live) { const int64_t l_ = *g.live - g.live_off; L = l_ < 0 ? 0 : (l_ < g.half ? l_ : g.half); }
    const int64_t NQ = L * g.sets;                         // live problems: q -> problem (q / L) * half + q % L
    for (int64_t q_ = blockIdx.x; q_ < NQ; q_ += gridDim.x) {
        const int64_t p = q_ < L ? q_ : g.half + (q_ - L);
        // Everything the epilogues address is invariant over the problem loop (the scratch block, the lane's offsets in a TF image):
        // left alone the compiler hoists several hundred store addresses and fragment bases out of the loop and spills the
        // accumulators around them (672 spilled VGPRs in the first build).  An opaque copy of the lane index per problem ...
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        // ... and of the uniform bases (weights, biases, scratch): hoisted out of the problem loop, the fragment addresses of nine
        // products alone are several hundred SGPRs
        const h8v* pw = g.pw;
        const float* pb = g.pb;
        char* scr = scr0;
        int wave = wave0;
        asm volatile("" : "+s"(pw), "+s"(pb), "+s"(scr), "+s"(wave));
        // ragged-tile units of this wave: plain products - token tiles w and (waves 0, 1) w + 8; v^T - token tiles 2 w, 2 w + 1 (waves 0..4)
        const int rt0 = wave, rt1 = wave < 2 ? wave + 8 : -1;
        const int vt0 = wave < 5 ? 2 * wave : -1, vt1 = wave < 5 ? 2 * wave + 1 : -1;
        const int gq = lane >> 4, j = lane & 15;
        const LaneOff lo_ = lane_offsets(lane);
        const char* img_x = g.tf_x + p * TF_BYTES;
        int64_t ps = p + g.shift;
        if (ps >= g.P) ps -= g.P;
        const char* img_s = g.tf_s + ps * TF_BYTES;
        f4v acc[2][FNT], accr[2];
#ifdef PATS_DIAG
        tlast = __builtin_amdgcn_s_memrealtime();
#endif
        // ================= source -> k (TF image + packed extras), v^T (A fragments of the second attention product) =================
        if (first) dma_fill<TF_BYTES>(lds, img_s, wave, lane);      // (later problems: started under the previous output epilogue)
        first = false;
        wg_barrier_global();
        FT(0);
        zero_acc(acc, accr);
        conv_pass<false, 9>(pw + FW_K, 0, 0, lds, wave, lane, acc, accr, rt0, rt1);
        FT(1);
        {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f4v bias = load4(pb + FB_K + 16 * (2 * wave + m) + 4 * gq) * PRE;
#pragma unroll
                for (int tt = 0; tt < FNT; ++tt) store_tf(scr + SC_K, 2 * wave + m, tt, fma4(acc[m][tt], bcast4(UNS * PRE), bias), lane, lo_);
            }
            // extras: rows 256 + 4 g + r = (head 2 g + (r >> 1), channel 64 + (r & 1)); A packing (h0 h1 l0 l1 h0 h1 0 0)
            const f4v bias = load4(pb + FB_K + 256 + 4 * (gq & 1)) * PRE;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tt = u == 0 ? rt0 : rt1;
                if (tt >= 0 && gq < 2 && (tt < 9 || j == 0)) {
                    h4v hi, lo;
                    split4_pre(fma4(accr[u], bcast4(UNS * PRE), bias), hi, lo);
                    char* d = scr + SC_KX + ((2 * gq) * FN + 16 * tt + j) * 16;
                    *reinterpret_cast<h8v*>(d) = h8v{hi.x, hi.y, lo.x, lo.y, hi.x, hi.y, 0, 0};
                    *reinterpret_cast<h8v*>(d + FN * 16) = h8v{hi.z, hi.w, lo.z, lo.w, hi.z, hi.w, 0, 0};
                }
            }
        }
        FT(2);
        zero_acc(acc, accr);
        conv_pass<true, 9>(pw + FW_V, 0, 0, lds, wave, lane, acc, accr, vt0, vt1);
        FT(3);
        wg_barrier();                                      // the source image has been read: x lands under the epilogue
        dma_fill<TF_BYTES>(lds, img_x, wave, lane);
        {
            // rows = tokens 16 t + 4 g + r, column = channel 16 mt + j: key slots (g, e) of k-step kk = token tiles 2 kk (e < 4), 2 kk + 1
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (m == 2 && wave >= 5) break;
                const int mt = m < 2 ? 2 * wave + m : 16;
                const float bias = pb[FB_V + 16 * mt + j] * PRE;
#pragma unroll
                for (int kk = 0; kk < 5; ++kk) {
                    if (m == 2 && kk != wave) continue;
                    f4v a = fma4(m < 2 ? acc[m][2 * kk] : accr[0], bcast4(UNS * PRE), bcast4(bias));
                    f4v b = fma4(m < 2 ? acc[m][2 * kk + 1] : accr[1], bcast4(UNS * PRE), bcast4(bias));
                    if (kk == 4) {                        // token tile 9 holds token 144 alone; key slots past it must be exact zeros
                        b.y = 0.f; b.z = 0.f; b.w = 0.f;
                        if (gq != 0) b.x = 0.f;
                    }
                    h4v ah, al, bh, bl;
                    split4_pre(a, ah, al);
                    split4_pre(b, bh, bl);
                    char* d = scr + SC_V + (mt * 5 + kk) * 2048 + lane * 16;
                    *reinterpret_cast<h8v*>(d) = h8v{ah.x, ah.y, ah.z, ah.w, bh.x, bh.y, bh.z, bh.w};
                    *reinterpret_cast<h8v*>(d + 1024) = h8v{al.x, al.y, al.z, al.w, bl.x, bl.y, bl.z, bl.w};
                }
            }
        }
        FT(4);
        // ================= x -> q (TF image + packed extras, B packing (h0 h1 h0 h1 l0 l1 0 0)) ======================================
        wg_barrier_global();
        FT(5);
        zero_acc(acc, accr);
        conv_pass<false, 9>(pw + FW_Q, 0, 0, lds, wave, lane, acc, accr, rt0, rt1);
        FT(6);
        wg_barrier();                                      // the x image has been read: k_0, v_0 and the extras tile land under the epilogue
        dma_fill<4 * TFB>(lds + L_KA, scr + SC_K, wave, lane);       // (k and v have been in the scratch block since the barriers above)
        dma_fill<FN * 16>(lds + L_KA + 4 * TFB, scr + SC_KX, wave, lane);
        dma_fill<4 * V_TILE>(lds + L_V, scr + SC_V, wave, lane);
        dma_fill<V_TILE>(lds + L_VX, scr + SC_V + 16 * V_TILE, wave, lane);
        {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f4v bias = load4(pb + FB_Q + 16 * (2 * wave + m) + 4 * gq) * PRE;
#pragma unroll
                for (int tt = 0; tt < FNT; ++tt) store_tf(scr + SC_Q, 2 * wave + m, tt, fma4(acc[m][tt], bcast4(UNS * PRE), bias), lane, lo_);
            }
            const f4v bias = load4(pb + FB_Q + 256 + 4 * (gq & 1)) * PRE;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tt = u == 0 ? rt0 : rt1;
                if (tt >= 0 && gq < 2 && (tt < 9 || j == 0)) {
                    h4v hi, lo;
                    split4_pre(fma4(accr[u], bcast4(UNS * PRE), bias), hi, lo);
                    char* d = scr + SC_QX + ((2 * gq) * FN + 16 * tt + j) * 16;
                    *reinterpret_cast<h8v*>(d) = h8v{hi.x, hi.y, hi.x, hi.y, lo.x, lo.y, 0, 0};
                    *reinterpret_cast<h8v*>(d + FN * 16) = h8v{hi.z, hi.w, hi.z, hi.w, lo.z, lo.w, 0, 0};
                }
            }
        }
        FT(7);
        wg_barrier_global();                               // q, k, v are in the scratch block; the x image has been read
        FT(8);
        // ================= attention: unit = (head, 16-query tile); its output replaces its own q tile ===============================
        // this wave's queries of a unit: B operand, two k-steps + the packed extras (every lane reads the piece of its query; lanes
        // k / 8 > 0 then take zeros).  Loaded one unit AHEAD: the scratch block is an L2 / Infinity-Cache round trip away.
        struct QF { h8v h0, l0, h1, l1, x; };
        auto qload = [&](int h, int qt, QF& q) {
            const int qoff = tf_off(false, qt, lane);
            const int qtok = qt < 9 ? 16 * qt + j : 144;
            q.h0 = *reinterpret_cast<const h8v*>(scr + SC_Q + tf_blk(2 * h, 0) + qoff);
            q.l0 = *reinterpret_cast<const h8v*>(scr + SC_Q + tf_blk(2 * h, 1) + qoff);
            q.h1 = *reinterpret_cast<const h8v*>(scr + SC_Q + tf_blk(2 * h + 1, 0) + qoff);
            q.l1 = *reinterpret_cast<const h8v*>(scr + SC_Q + tf_blk(2 * h + 1, 1) + qoff);
            q.x = *reinterpret_cast<const h8v*>(scr + SC_QX + (h * FN + qtok) * 16);
        };
        char* att_dst = FULL ? scr + SC_Q : g.tf_att + p * TF_BYTES;
        QF q;
        qload(0, wave, q);
        for (int h = 0; h < 4; ++h) {
            wg_barrier_global();                           // k_h, v_h (and the extras tile) have landed
            FT(9);
            if (h < 3) {
                char* kn = lds + L_KA + ((h + 1) & 1) * KH_BYTES;
                dma_fill<4 * TFB>(kn, scr + SC_K + (h + 1) * 4 * TFB, wave, lane);
                dma_fill<FN * 16>(kn + 4 * TFB, scr + SC_KX + (h + 1) * FN * 16, wave, lane);
            }
            const char* kb = lds + L_KA + (h & 1) * KH_BYTES;
            for (int qt = wave; qt < FNT; qt += 8) {
                const h8v qx = gq == 0 ? q.x : zero8();
                f4v st[FNT];
                const float c = UNS * 0.12309149097933272f * LOG2E;          // accumulator -> exponent of 2: 2^-12 / sqrt(66) * log2(e)
                float mx = -INFINITY;
                // the key fragments of tile kt + 1 are read while the MFMAs of tile kt run (two register sets)
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
                    f4v s = mfma3(k.h0, k.l0, q.h0, q.l0, f4v{0.f, 0.f, 0.f, 0.f});         // rows = keys 16 kt + 4 g + r, column = query
                    s = mfma3(k.h1, k.l1, q.h1, q.l1, s);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(gq == 0 ? k.x : zero8(), qx, s, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = s[r] * c;
                        if (kt == 9 && (gq != 0 || r != 0)) v = -INFINITY;                  // keys 145..: not there
                        s[r] = v;
                        mx = fmaxf(mx, v);
                    }
                    st[kt] = s;
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the next unit's queries: this wave's second tile of the head (waves 0, 1), else its tile of the next head
                const int qt_own = qt;
                if (qt + 8 < FNT) qload(h, qt + 8, q);
                else if (h < 3) qload(h + 1, wave, q);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float den = 0.f;
#pragma unroll
                for (int kt = 0; kt < FNT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = fast_exp2(st[kt][r] - mx);
                        st[kt][r] = pr;
                        den += pr;
                    }
                den += __shfl_xor(den, 16);
                den += __shfl_xor(den, 32);
                const float inv = 1.0f / den;
                h8v ph[5], pl[5];
#pragma unroll
                for (int kk = 0; kk < 5; ++kk) {
                    h4v h0, l0, h1, l1;
                    split4_pre(st[2 * kk] * PRE, h0, l0);
                    split4_pre(st[2 * kk + 1] * PRE, h1, l1);
                    ph[kk] = h8v{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    pl[kk] = h8v{l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                }
                const float osc = UNS * PRE * inv;
                h8v vf[2][5][2];
                auto vload = [&](int dt, h8v (&v)[5][2]) {
                    const char* vb = dt < 4 ? lds + L_V + dt * V_TILE : lds + L_VX;
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
                    f4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 5; ++kk) o = mfma3(vf[dt & 1][kk][0], vf[dt & 1][kk][1], ph[kk], pl[kk], o);     // rows = channels, column = query
                    __builtin_amdgcn_sched_barrier(0);
                    o = o * osc;
                    if (dt < 4) {
                        store_tf(att_dst, 4 * h + dt, qt_own, o, lane, lo_);
                    } else if (gq == (h >> 1) && (qt_own < 9 || j == 0)) {
                        // rows 2 h', 2 h' + 1 of the extras tile are head h's channels 64, 65 -> bytes 4 h .. of the ragged block's piece
                        const float e0 = (h & 1) ? o.z : o.x, e1 = (h & 1) ? o.w : o.y;
                        const _Float16 a0 = (_Float16)e0, a1 = (_Float16)e1;
                        const h2v hi = {a0, a1}, lo = {(_Float16)(e0 - (float)a0), (_Float16)(e1 - (float)a1)};
                        const int off = tf_off(true, qt_own, j) + 4 * h;
                        *reinterpret_cast<h2v*>(att_dst + tf_blk(8, 0) + off) = hi;
                        *reinterpret_cast<h2v*>(att_dst + tf_blk(8, 1) + off) = lo;
                    }
                }
            }
            FT(10);
            if (h < 3) {
                wg_barrier();                              // every wave is done with v_h
                FT(11);
                dma_fill<4 * V_TILE>(lds + L_V, scr + SC_V + (h + 1) * 4 * V_TILE, wave, lane);
            }
        }
        wg_barrier_global();                               // the attention output is in the scratch block; the staging area is free
        FT(11);
        if (!FULL) {                                       // the MLP is gnn_fine_mlp_kernel's: next problem (its source lands now)
            if (q_ + gridDim.x < NQ) {
                const int64_t qn = q_ + gridDim.x;
                int64_t pn = (qn < L ? qn : g.half + (qn - L)) + g.shift;
                if (pn >= g.P) pn -= g.P;
                dma_fill<TF_BYTES>(lds, g.tf_s + pn * TF_BYTES, wave, lane);
            }
#ifdef PATS_DIAG
            ++nprob;
#endif
            continue;
        }
        // ================= hidden = relu(bn(W1x x + W1a att + b1')): two halves of 264 rows =========================================
        dma_fill<TF_BYTES>(lds, scr + SC_Q, wave, lane);   // att
        wg_barrier_global();
        FT(12);
        zero_acc(acc, accr);
        conv_pass<false, 18>(pw + FW_1, 0, 9, lds, wave, lane, acc, accr, rt0, rt1);
        FT(13);
        wg_barrier();
        dma_fill<TF_BYTES>(lds, img_x, wave, lane);
        wg_barrier_global();
        FT(14);
        conv_pass<false, 18>(pw + FW_1, 0, 0, lds, wave, lane, acc, accr, rt0, rt1);
        FT(15);
        auto hidden_out = [&](const f4v (&a)[2][FNT], const f4v (&ar)[2], int hf, char* dst) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int mt = m < 2 ? 2 * wave + m : 16, ch = hf * 272 + 16 * mt + 4 * gq;
                const f4v bias = load4(pb + FB_1 + ch), sc = load4(pb + FB_A + ch), sh = load4(pb + FB_S + ch);
                const f4v scl = sc * (UNS * PRE), shf = (bias * sc + sh) * PRE;
#pragma unroll
                for (int tt = 0; tt < (m < 2 ? FNT : 2); ++tt) {
                    const int tok_t = m < 2 ? tt : (tt == 0 ? rt0 : rt1);
                    if (tok_t < 0) continue;
                    f4v v = fma4(m < 2 ? a[m][tt] : ar[tt], scl, shf);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];            // ReLU that keeps NaN
                    store_tf(dst, mt, tok_t, v, lane, lo_);
                }
            }
        };
        hidden_out(acc, accr, 0, scr + SC_K);
        FT(16);
        zero_acc(acc, accr);
        conv_pass<false, 18>(pw + FW_1, 17, 0, lds, wave, lane, acc, accr, rt0, rt1);
        FT(15);
        wg_barrier();
        dma_fill<TF_BYTES>(lds, scr + SC_Q, wave, lane);   // att again
        wg_barrier_global();
        FT(12);
        conv_pass<false, 18>(pw + FW_1, 17, 9, lds, wave, lane, acc, accr, rt0, rt1);
        FT(13);
        wg_barrier();                                      // att has been read: hidden[264:528] goes from the accumulators straight into the slot
        hidden_out(acc, accr, 1, lds);
        FT(16);
        wg_barrier();
        // ================= out = W2 hidden + b2 [+ residual] -> fp32 blocked + TF image ==============================================
        zero_acc(acc, accr);
        conv_pass<false, 18>(pw + FW_2, 0, 9, lds, wave, lane, acc, accr, rt0, rt1);      // the second half of the channels first
        FT(18);
        wg_barrier();
        dma_fill<TF_BYTES>(lds, scr + SC_K, wave, lane);   // hidden[0:264] (in the scratch block since the barriers behind its epilogue)
        wg_barrier_global();
        FT(17);
        conv_pass<false, 18>(pw + FW_2, 0, 0, lds, wave, lane, acc, accr, rt0, rt1);
        FT(18);
        wg_barrier();                                      // the slot is free: the next problem's source lands under the output epilogue
        if (q_ + gridDim.x < NQ) {
            const int64_t qn = q_ + gridDim.x;
            int64_t pn = (qn < L ? qn : g.half + (qn - L)) + g.shift;
            if (pn >= g.P) pn -= g.P;
            dma_fill<TF_BYTES>(lds, g.tf_s + pn * TF_BYTES, wave, lane);
        }
        {
            const char* R = g.tf_res ? g.tf_res + p * TF_BYTES : nullptr;
            char* TO = g.tf_out + p * TF_BYTES;
            // per row tile: its ten residual pieces in one batch (a load -> add -> store chain per piece would pay a memory round trip
            // each; all 22 at once is 88 more registers beside the accumulators - the allocator then spills, and a spill reload waits
            // for every store issued before it).  The residual comes from the descriptor's IMAGE - lines the fills of this problem
            // just read - and is exact to the 22 bits an image holds.
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int mt = m < 2 ? 2 * wave + m : 16, ch = 16 * mt + 4 * gq;
                const f4v bias = load4(pb + FB_2 + ch);
                const bool rows_ok = mt < 16 || gq < 2;
                u4v_t res[FNT];
                if (R) {
#pragma unroll
                    for (int tt = 0; tt < (m < 2 ? FNT : 2); ++tt) {
                        const int tok_t = m < 2 ? tt : (tt == 0 ? rt0 : rt1);
                        res[tt] = u4v_t{0u, 0u, 0u, 0u};
                        if (tok_t >= 0) res[tt] = load_tf_piece(R, mt, tok_t, lane, lo_);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < (m < 2 ? FNT : 2); ++tt) {
                    const int tok_t = m < 2 ? tt : (tt == 0 ? rt0 : rt1);
                    if (tok_t < 0) continue;
                    const bool live = rows_ok && !(tok_t == 9 && j != 0);
                    f4v v = fma4(m < 2 ? acc[m][tt] : accr[tt], bcast4(UNS), bias);
                    if (R) v = tf_piece_value(res[tt]) + v;                  // (every lane: the pieces are paired across lanes)
                    if (live) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) bad |= !(fabsf(v[r]) <= 3.0e38f);
                    }
                    store_tf(TO, mt, tok_t, v * PRE, lane, lo_);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        FT(19);
#ifdef PATS_DIAG
        ++nprob;
#endif
    }
    if (bad) atomicOr(g.flag, 1);
#ifdef PATS_DIAG
    if (g.tl && t == 0) {
        for (int k = 0; k < FT_N - 1; ++k) g.tl[(size_t)blockIdx.x * FT_N + k] = tsum[k];
        g.tl[(size_t)blockIdx.x * FT_N + FT_N - 1] = nprob;
    }
#endif
}

// ---- the MLP half of the layer on FLATTENED column tiles ------------------------------------------------------------------------
// hidden = relu(bn(W1x x + W1a att + b1')), out = W2 hidden + b2 + x are per-token: a workgroup takes 64 columns of the flattened
// (problem, token) axis - no token padding (145 = 9 x 16 + 1 costs the per-problem kernel 10 %), and x-tile + att-tile (2 x 66 KB as
// fragments) sit in LDS TOGETHER, so mlp[0] is one 18-k-step loop whose 528 x 64 output lives in the accumulators (17 tile units a
// wave), is written - BatchNorm, ReLU, split - over the operands it came from, and feeds mlp[3] from there: the hidden tensor never
// leaves the CU and x / att are read once.  The residual is rebuilt from the x fragments in LDS before they are overwritten.
// Operands arrive by gather DMA from the per-problem TF images (a lane's source address is its column's; the 64 pieces of a
// fragment land contiguously), the output leaves as 16-byte pieces scattered into the per-problem images of the next layer.
constexpr int MT_HALF = 8 * 8192 + 2 * 1024;          // one operand half-tile: eight full k-steps [plane][4 column tiles][64 x 16 B] + the ragged one
constexpr int MLP_LDS = 2 * MT_HALF;                  // 135 168

struct MlpArgs {
    const char* tf_x; const char* tf_att; char* tf_out;
    const h8v* pw; const float* pb;
    int64_t P, half; int sets;
    const int64_t* live; int64_t live_off;
    int* flag;
    int residual;              // add x (AttentionalGNN.forward's desc + delta); 0: the delta alone
    const int* gate;
};

// byte offset of token t's 16-byte piece (k-group kq) inside a (k-step, plane) block of a per-problem image
__device__ __forceinline__ int img_tok_off(bool ragged, int t, int kq) {
    if (!ragged) return t < 144 ? (t >> 4) * 1024 + kq * 256 + (t & 15) * 16 : 9216 + kq * 16;
    return t < 144 ? (t >> 4) * 256 + (t & 15) * 16 : 2304;
}

__global__ void __launch_bounds__(512, 1)
gnn_fine_mlp_kernel(MlpArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (g.gate && *g.gate == 0) return;
    const int t_ = threadIdx.x, lane0 = t_ & 63, wave0 = __builtin_amdgcn_readfirstlane(t_ >> 6);
    int64_t L = g.half;
    if (g.live) { const int64_t l_ = *g.live - g.live_off; L = l_ < 0 ? 0 : (l_ < g.half ? l_ : g.half); }
    const int64_t ncol = L * g.sets * FN, ntile = (ncol + 63) >> 6;
    bool bad = false;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int lane = lane0, wave = wave0;
        const h8v* pw = g.pw;
        const float* pb = g.pb;
        asm volatile("" : "+v"(lane), "+s"(wave), "+s"(pw), "+s"(pb));       // (as in the layer kernel: nothing hoisted out of the tile loop)
        const int gq = lane >> 4, j = lane & 15;
        // this lane's four columns (one per column tile): problem image and token
        int64_t pbase[4];
        int tok[4];
        bool colok[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            int64_t c = tile * 64 + ct * 16 + j;
            colok[ct] = c < ncol;
            if (c >= ncol) c = ncol - 1;
            const int64_t q = c / FN;
            tok[ct] = (int)(c - q * FN);
            pbase[ct] = (q < L ? q : g.half + (q - L)) * (int64_t)TF_BYTES;
        }
        wg_barrier();                                      // the previous tile's hidden fragments have been read
        // ---- gather: 2 x 72 fragments, 18 a wave ------------------------------------------------------------------------------------
#pragma unroll 2
        for (int i = 0; i < 18; ++i) {
            const int idx = wave + 8 * i, part = idx >= 72 ? 1 : 0, r = idx - 72 * part;
            const char* img = part ? g.tf_att : g.tf_x;
            if (r < 64) {
                const int ks = r >> 3, plane = (r >> 2) & 1, ct = r & 3;
                const int64_t pbc = ct == 0 ? pbase[0] : ct == 1 ? pbase[1] : ct == 2 ? pbase[2] : pbase[3];
                const int tk = ct == 0 ? tok[0] : ct == 1 ? tok[1] : ct == 2 ? tok[2] : tok[3];
                const char* src = img + pbc + (2 * ks + plane) * TFB + img_tok_off(false, tk, gq);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lds + part * MT_HALF + ks * 8192 + plane * 4096 + ct * 1024), 16, 0, 0);
            } else {
                const int rr = r - 64, plane = rr >> 2, ct = rr & 3;
                const int64_t pbc = ct == 0 ? pbase[0] : ct == 1 ? pbase[1] : ct == 2 ? pbase[2] : pbase[3];
                const int tk = ct == 0 ? tok[0] : ct == 1 ? tok[1] : ct == 2 ? tok[2] : tok[3];
                const char* src = img + pbc + TF_MAIN + plane * TFR + img_tok_off(true, tk, 0);
                if (lane < 16)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(lds + part * MT_HALF + 65536 + plane * 1024 + ct * 256), 16, 0, 0);
            }
        }
        wg_barrier_global();
        // B fragments of k-step kk (0..8; 8 = ragged) of operand half `part`
        auto bload = [&](int part, int kk, h8v (&bh)[4], h8v (&bl)[4]) {
            const char* base = lds + part * MT_HALF;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (kk < 8) {
                    bh[ct] = *reinterpret_cast<const h8v*>(base + kk * 8192 + ct * 1024 + lane * 16);
                    bl[ct] = *reinterpret_cast<const h8v*>(base + kk * 8192 + 4096 + ct * 1024 + lane * 16);
                } else {
                    const h8v a = *reinterpret_cast<const h8v*>(base + 65536 + ct * 256 + j * 16);
                    const h8v b = *reinterpret_cast<const h8v*>(base + 65536 + 1024 + ct * 256 + j * 16);
                    bh[ct] = lane < 16 ? a : zero8();
                    bl[ct] = lane < 16 ? b : zero8();
                }
            }
        };
        // ---- mlp[0]: row tiles 2 w, 2 w + 1 of both halves (FW_1 numbering: half 1 starts at tile 17) + one unit of a ragged tile -------
        const int rag_tile = wave < 4 ? 16 : 33, rag_ct = wave & 3;
        f4v acc[4][4], accr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[m][ct] = f4v{0.f, 0.f, 0.f, 0.f};
        {
            h8v a[2][5][2];
            auto aload5 = [&](int ks, h8v (&r)[5][2]) {
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    const int mt = m < 2 ? 2 * wave + m : m < 4 ? 17 + 2 * wave + (m - 2) : rag_tile;
                    gptr_h8 Wf = uniform_ptr(pw + FW_1 + ((size_t)mt * 18 + ks) * FR);
                    r[m][0] = Wf[lane];
                    r[m][1] = Wf[64 + lane];
                }
            };
            aload5(0, a[0]);
#pragma unroll 1
            for (int kp = 0; kp < 9; ++kp) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ks = 2 * kp + e;
                    if (ks + 1 < 18) aload5(ks + 1, a[1 - e]);
                    h8v bh[4], bl[4];
                    bload(ks >= 9 ? 1 : 0, ks >= 9 ? ks - 9 : ks, bh, bl);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc[m][ct] = mfma3(a[e][m][0], a[e][m][1], bh[ct], bl[ct], acc[m][ct]);
                        if (ct == rag_ct) accr = mfma3(a[e][4][0], a[e][4][1], bh[ct], bl[ct], accr);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- the residual of this wave's mlp[3] units, from the x fragments still in LDS: rows 16 mt + 4 g.. of its four columns ---------
        f4v res[2][4], resr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) res[m][ct] = f4v{0.f, 0.f, 0.f, 0.f};
        if (g.residual) {
            auto rd = [&](int mt, int ct) {
                const char* bx = mt < 16 ? lds + (mt >> 1) * 8192 + ct * 1024 + ((2 * (mt & 1) + (gq >> 1)) * 16 + j) * 16 + (gq & 1) * 8
                                         : lds + 65536 + ct * 256 + j * 16 + (gq & 1) * 8;
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
        wg_barrier();                                      // every wave is done with x | att: hidden takes their place
        {
            auto put = [&](int mtl, int hf, int ct, const f4v acc_, bool ragged) {
                const int ch = hf * 272 + 16 * mtl + 4 * gq;
                const f4v bias = load4(pb + FB_1 + ch), sc = load4(pb + FB_A + ch), sh = load4(pb + FB_S + ch);
                f4v v = fma4(acc_, sc * (UNS * PRE), (bias * sc + sh) * PRE);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];                  // ReLU that keeps NaN
                h4v hi, lo;
                split4_pre(v, hi, lo);
                const u2v H = __builtin_bit_cast(u2v, hi), Lo = __builtin_bit_cast(u2v, lo);
                unsigned hx = H.x, hy = H.y, lx = Lo.x, ly = Lo.y;
                lane_swap16(hx, lx);
                lane_swap16(hy, ly);
                char* d = ragged ? lds + hf * MT_HALF + 65536 + (gq & 1) * 1024 + ct * 256 + j * 16
                                 : lds + hf * MT_HALF + (mtl >> 1) * 8192 + (gq & 1) * 4096 + ct * 1024 + ((2 * (mtl & 1) + (gq >> 1)) * 16 + j) * 16;
                if (!ragged || gq < 2) *reinterpret_cast<u4v*>(d) = u4v{hx, hy, lx, ly};
            };
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) put(2 * wave + (m & 1), m >> 1, ct, acc[m][ct], false);
            put(16, wave >> 2, rag_ct, accr, true);
        }
        wg_barrier();
        // ---- mlp[3]: row tiles 2 w, 2 w + 1 over the four column tiles + (waves 0..3) column tile w of the ragged 17th ---------------
        f4v o[2][4], orr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) o[m][ct] = f4v{0.f, 0.f, 0.f, 0.f};
        {
            h8v a[2][3][2];
            auto aload3 = [&](int ks, h8v (&r)[3][2]) {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int mt = m < 2 ? 2 * wave + m : 16;
                    gptr_h8 Wf = uniform_ptr(pw + FW_2 + ((size_t)mt * 18 + ks) * FR);
                    r[m][0] = Wf[lane];
                    r[m][1] = Wf[64 + lane];
                }
            };
            aload3(0, a[0]);
#pragma unroll 1
            for (int kp = 0; kp < 9; ++kp) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ks = 2 * kp + e;
                    if (ks + 1 < 18) aload3(ks + 1, a[1 - e]);
                    h8v bh[4], bl[4];
                    bload(ks >= 9 ? 1 : 0, ks >= 9 ? ks - 9 : ks, bh, bl);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        o[0][ct] = mfma3(a[e][0][0], a[e][0][1], bh[ct], bl[ct], o[0][ct]);
                        o[1][ct] = mfma3(a[e][1][0], a[e][1][1], bh[ct], bl[ct], o[1][ct]);
                        if (ct == wave) orr = mfma3(a[e][2][0], a[e][2][1], bh[ct], bl[ct], orr);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- out = . + b2 + x -> the next layer's images (16-byte pieces, lane pairs exchange halves) ---------------------------------
        {
            auto emit = [&](int mt, int ct, const f4v acc_, const f4v res_, bool mine) {
                const f4v bias = load4(pb + FB_2 + 16 * mt + 4 * gq);
                const f4v v = fma4(acc_, bcast4(UNS), bias) + res_;
                const int64_t pbc = ct == 0 ? pbase[0] : ct == 1 ? pbase[1] : ct == 2 ? pbase[2] : pbase[3];
                const int tk = ct == 0 ? tok[0] : ct == 1 ? tok[1] : ct == 2 ? tok[2] : tok[3];
                const bool ok = (ct == 0 ? colok[0] : ct == 1 ? colok[1] : ct == 2 ? colok[2] : colok[3]) && mine && (mt < 16 || gq < 2);
                if (ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bad |= !(fabsf(v[r]) <= 3.0e38f);
                }
                h4v hi, lo;
                split4_pre(v * PRE, hi, lo);
                const u2v H = __builtin_bit_cast(u2v, hi), Lo = __builtin_bit_cast(u2v, lo);
                unsigned hx = H.x, hy = H.y, lx = Lo.x, ly = Lo.y;
                lane_swap16(hx, lx);
                lane_swap16(hy, ly);
                char* d = g.tf_out + pbc + (mt < 16 ? ((mt >> 1) * 2 + (gq & 1)) * TFB + img_tok_off(false, tk, 2 * (mt & 1) + (gq >> 1))
                                                    : TF_MAIN + (gq & 1) * TFR + img_tok_off(true, tk, 0));
                if (ok) *reinterpret_cast<u4v*>(d) = u4v{hx, hy, lx, ly};
            };
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) emit(2 * wave + m, ct, o[m][ct], res[m][ct], true);
            emit(16, wave & 3, orr, resr, wave < 4);
        }
    }
    if (bad) atomicOr(g.flag, 1);
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

static int fine_grid(int64_t P) {
    struct PerDevice { int state = 0; int n_cu = 256; };
    static PerDevice per_dev[64];
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) { (void)hipGetLastError(); dev_id = 0; }
    PerDevice& pd = per_dev[dev_id];
    if (pd.state == 0) {
        bool ok = hipFuncSetAttribute((const void*)gnn_fine_layer_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FINE_LDS) == hipSuccess &&
                  hipFuncSetAttribute((const void*)gnn_fine_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS) == hipSuccess;
#ifdef PATS_DIAG
        ok = ok && hipFuncSetAttribute((const void*)gnn_fine_layer_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FINE_LDS) == hipSuccess;
#endif
        if (!ok) (void)hipGetLastError();
        int v = 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess) (void)hipGetLastError();
        pd.n_cu = v > 0 ? v : 256;
        pd.state = ok ? 1 : -1;
    }
    if (pd.state != 1) return 0;
    return (int)std::min<int64_t>(P, pd.n_cu);
}
int fine_max_grid() { return 512; }                      // scratch blocks a workspace must provide at most (CUs of the device, capped)
size_t fine_scratch_bytes(int64_t P) { return (size_t)std::min<int64_t>(P, fine_max_grid()) * SC_BYTES; }
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
// one layer over P problems: image p of tf_x with source image (p + shift) % P of tf_s
// One layer over P problems (sets descriptor sets of P / sets rows each; live (optional): device-side row count of a set, minus
// live_off): image p of tf_x with source image (p + shift) % P of tf_s.  Two launches: q / k / v + attention per problem
// (gnn_fine_layer_kernel<false> -> tf_att), then the MLP on flattened 64-column tiles (gnn_fine_mlp_kernel -> tf_out; residual != 0:
// + x).  Diagnostic library, PATS_FINE_SPLIT=0: the whole layer per problem in the first kernel (the round's first form).
int launch_fine_layer(const char* tf_x, const char* tf_s, int64_t shift, int residual, int64_t P, const void* section,
                      char* tf_out, char* tf_att, char* scratch, int* flag, const int* gate, hipStream_t st, int sets,
                      const int64_t* live, int64_t live_off) {
    const int grid = fine_grid(P);
    if (grid <= 0) return PATS_ERR_UNSUPPORTED;
    const h8v* pw = (const h8v*)section;
    // De-phasing: a layer's far-memory traffic comes in bursts (image fills, epilogues) that every workgroup of a lockstep grid
    // issues at the same instants; workgroup i starts ((i >> 3) % 32) x 4 us late, which spreads them over a problem's period.
    // Only where a workgroup has enough problems to pay for the ramp.
    static const int stagger_env = [] { const char* e = diag_env("PATS_FINE_STAGGER"); return e ? atoi(e) : -1; }();
    const int stagger = stagger_env >= 0 ? stagger_env : (P >= 8 * (int64_t)grid ? 4 : 0);
    FineArgs g{tf_x, tf_s, residual ? tf_x : nullptr, tf_out, tf_att, pw, (const float*)(pw + FW_END), scratch, P, shift, flag, gate, live, live_off,
               P / sets, sets, stagger};
    const unsigned wgs = (unsigned)std::min(grid, fine_max_grid());
    bool split = true;
#ifdef PATS_DIAG
    static const bool whole = [] { const char* e = diag_env("PATS_FINE_SPLIT"); return e && atoi(e) == 0; }();
    split = !whole;
    g.tl = nullptr;
    if (diag_env("PATS_FINE_TL")) { (void)hipMalloc((void**)&g.tl, (size_t)wgs * FT_N * 8); (void)hipMemset(g.tl, 0, (size_t)wgs * FT_N * 8); }
    if (!split) hipLaunchKernelGGL(gnn_fine_layer_kernel<true>, dim3(wgs), dim3(512), FINE_LDS, st, g);
    else
#endif
    hipLaunchKernelGGL(gnn_fine_layer_kernel<false>, dim3(wgs), dim3(512), FINE_LDS, st, g);
#ifdef PATS_DIAG
    if (g.tl) {
        (void)hipStreamSynchronize(st);
        std::vector<long long> h((size_t)wgs * FT_N);
        (void)hipMemcpy(h.data(), g.tl, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[FT_N] = {0}, np = 0;
        for (unsigned w = 0; w < wgs; ++w) { for (int k = 0; k < FT_N - 1; ++k) sum[k] += (double)h[(size_t)w * FT_N + k]; np += (double)h[(size_t)w * FT_N + FT_N - 1]; }
        static const char* names[FT_N] = {"fill s", "k product", "k epilogue", "v^T product", "v^T epilogue (x fill under it)", "barrier", "q product", "q epilogue (k, v staging under it)",
            "barrier (q k v visible)", "attention: wait k_h v_h", "attention: units", "attention: barriers", "fill att", "hidden: att part (x2)",
            "barrier + fill x", "hidden: x part (x2)", "hidden epilogues", "fill hidden0", "out product (x2)", "out epilogue", "", "", "", ""};
        double tot = 0;
        for (int k = 0; k <= 20; ++k) tot += sum[k];
        fprintf(stderr, "gnn_fine timeline, %s (%u workgroups, %.0f problems; mean us per problem, thread 0): total %.2f\n",
                split ? "first kernel of the split layer" : "whole layer in one kernel", wgs, np, tot / np / 100.0);
        for (int k = 0; k <= 19; ++k) if (sum[k] > 0) fprintf(stderr, "  %-38s %7.2f\n", names[k], sum[k] / np / 100.0);
        (void)hipFree(g.tl);
    }
#endif
    int rc = check_launch("gnn_fine_layer_kernel");
    if (rc || !split) return rc;
    MlpArgs m{tf_x, tf_att, tf_out, pw, (const float*)(pw + FW_END), P, P / sets, sets, live, live_off, flag, residual, gate};
    hipLaunchKernelGGL(gnn_fine_mlp_kernel, dim3(wgs), dim3(512), MLP_LDS, st, m);
    return check_launch("gnn_fine_mlp_kernel");
}

}  // namespace pats
