// Cost build of ONE 65x65 problem by one wave (models/third_layer.py:156-158), shared by the
// standalone cost kernel, sinkhorn65_kernel (sinkhorn.hip) and the block-layout fused kernel
// (third_fused.hip).
#pragma once
#include "common.hpp"

namespace pats {

constexpr int C65_NB = 64, C65_NT = 65;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

struct Cost65Acc {
    f32x16 c00, c01, c10, c11;      // tile (ti, tj): element [r] <-> matrix (2*rc + ti, 2*li + tj),
                                    // rc = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), li = lane & 31
    float er0, er1;                 // dustbin row    (64, 2*li + {0,1})   - complete after the half-wave combine
    float ec0, ec1;                 // dustbin column (2*li + {0,1}, 64)
    float cn;                       // corner (64, 64)
};

// The 64x64 core as 2x2 tiles of v_mfma_f32_32x32x2_f32 over D: MFMA row index li of tile t stands
// for matrix row 2*li + t (tile 0 = even rows, tile 1 = odd rows; same for columns), so one 8-byte
// load per lane feeds BOTH tiles and every 32-lane half reads one contiguous 256-byte stretch of a
// descriptor row; every descriptor element is loaded exactly once.  The dustbin row / column /
// corner ride along as fp32 FMA chains in the VALU slots between the MFMAs (even-k and odd-k
// half-wave chains, summed at the end); their operands d0[:,64], d1[:,64] are parked in `edge_lds`
// (2 x 512 floats of LDS the caller is not using yet).  D % 32 == 0, D <= 512.  Results are RAW dot
// products; cost65_scale applies the reference's `/ D**.5` then `0.1 *`.
// `ld` = elements between consecutive descriptor rows (65 for [D,65] descriptors; heads*65 for one
// head of a [dim,heads,65] attention operand).
template <bool NO_MFMA = false>     // NO_MFMA: timing ablation only (loads and edge chains stay, the four MFMAs become one add)
__device__ __forceinline__ void cost65_accumulate(const float* __restrict__ A, const float* __restrict__ B,
                                                  int D, float* edge_lds, int lane, Cost65Acc& o,
                                                  int ld = C65_NT) {
    const int li = lane & 31, lk = lane >> 5;
    constexpr int NB = C65_NB;
    f32x16 c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
    float er0 = 0.f, er1 = 0.f, ec0 = 0.f, ec1 = 0.f, cn = 0.f;
    float* eA = edge_lds;
    float* eB = edge_lds + 512;
    for (int k = lane; k < D; k += 64) {
        eA[k] = A[k * ld + NB];
        eB[k] = B[k * ld + NB];
    }
    const float* pa = A + lk * ld + 2 * li;
    const float* pb = B + lk * ld + 2 * li;
    // explicit software pipeline over blocks of KB = 4 k-steps (8 descriptor rows): a ring of four
    // named register buffers keeps three blocks of loads in flight ahead of the MFMAs (deeper rings
    // were measured slower: the phase is HBM-bandwidth-bound, and the extra registers spill)
    constexpr int KB = 4;
    struct Blk { f2u a[KB], b[KB]; };
    auto load_blk = [&](int k0, Blk& q) {
#pragma unroll
        for (int s_ = 0; s_ < KB; ++s_) {
            const int k = k0 + 2 * s_;                    // D % 32 == 0: whole rings only
            q.a[s_] = *reinterpret_cast<const f2u*>(pa + k * ld);
            q.b[s_] = *reinterpret_cast<const f2u*>(pb + k * ld);
        }
    };
    auto compute_blk = [&](int k0, const Blk& q) {
#pragma unroll
        for (int s_ = 0; s_ < KB; ++s_) {
            const f2u av = q.a[s_], bv = q.b[s_];
            const float ea = eA[k0 + 2 * s_ + lk], eb = eB[k0 + 2 * s_ + lk];
            if (NO_MFMA) {
                c00[s_ & 15] += av.x * bv.x + av.y * bv.y;
            } else {
            c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, c11, 0, 0, 0);
            }
            er0 = fmaf(ea, bv.x, er0);
            er1 = fmaf(ea, bv.y, er1);
            ec0 = fmaf(av.x, eb, ec0);
            ec1 = fmaf(av.y, eb, ec1);
            cn = fmaf(ea, eb, cn);
        }
    };
    const int nblk = D / (2 * KB);
    auto k0_of = [&](int blk) { return (blk < nblk ? blk : 0) * 2 * KB; };   // wrap: harmless reloads at the end
    Blk r0, r1, r2, r3;
    load_blk(k0_of(0), r0); load_blk(k0_of(1), r1); load_blk(k0_of(2), r2);
    wg_barrier();                                     // eA / eB visible
    for (int blk = 0; blk < nblk; blk += 4) {
        load_blk(k0_of(blk + 3), r3); compute_blk(k0_of(blk + 0), r0);
        load_blk(k0_of(blk + 4), r0); compute_blk(k0_of(blk + 1), r1);
        load_blk(k0_of(blk + 5), r1); compute_blk(k0_of(blk + 2), r2);
        load_blk(k0_of(blk + 6), r2); compute_blk(k0_of(blk + 3), r3);
    }
    wg_barrier();                                     // done with eA / eB: the caller may reuse the LDS
    o.er0 = er0 + __shfl_xor(er0, 32); o.er1 = er1 + __shfl_xor(er1, 32);
    o.ec0 = ec0 + __shfl_xor(ec0, 32); o.ec1 = ec1 + __shfl_xor(ec1, 32);
    o.cn = cn + __shfl_xor(cn, 32);
    o.c00 = c00; o.c01 = c01; o.c10 = c10; o.c11 = c11;
}

// The same contraction with every fp32 operand split into two fp16 halves, x = hi + lo (both round-to-nearest:
// |x - hi - lo| <= 2^-24 |x| unless lo is subnormal, i.e. at fp32 resolution), and THREE exact-product passes of
// v_mfma_f32_32x32x16_f16 per tile (hi.hi + hi.lo + lo.hi, fp32 accumulation; lo.lo <= 2^-22 |x y| is dropped):
// 96 matrix instructions of 32 cycles instead of 256 of 64 (the fp32 MFMA runs at the vector FMA rate on the vector
// FMA lanes), paid for with five VALU instructions per loaded pair (v_cvt_pk_f16_f32, two v_cvt_f32_f16, one
// v_pk_add_f32, v_cvt_pk_f16_f32).  Accuracy: the representation error of the split is 4x BELOW the rounding error the
// k-ordered fp32 fma chain itself accumulates over D = 128 (measured against float64 on the synthetic descriptors:
// mean |error| of the raw dot product 5.8e-6 for the split, 1.7e-5 for the fp32 chain).  |x| > 1023 overflows the
// (pre-scaled) fp16 hi part to inf -> NaN scores -> the Sinkhorn guard hands the problem to the fp32 kernel.
// Lane l supplies A[i = l & 31][k = 8 (l >> 5) + e], e = 0..7, so each lane loads eight consecutive descriptor rows
// per 16-channel step (same eight-byte loads, same even / odd column tiling, same D fragment layout as above).
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f2p __attribute__((ext_vector_type(2)));

constexpr float C65_PRESCALE = 64.0f;      // 2^6 on both operands, 2^-12 on the product: see below
__device__ __forceinline__ void split_pair(float x0, float x1, h2v& hi, h2v& lo) {
    // fp16 keeps 11 bits but only 5 of exponent: the lo half of an operand below 0.25 would be subnormal (and the
    // matrix pipe flushes subnormal fp16 inputs - measured: one target coordinate in 18 000 moved by 2e-2 px).  Scaling
    // by 2^6 first is exact and keeps lo normal down to |x| = 0.004; hi then overflows only beyond |x| = 1023.
    const f2p x = f2p{x0, x1} * C65_PRESCALE;
    hi = __builtin_convertvector(x, h2v);
    const f2p r = x - __builtin_convertvector(hi, f2p);
    lo = __builtin_convertvector(r, h2v);
}
__device__ __forceinline__ h8v pack8(h2v a, h2v b, h2v c, h2v d) { return h8v{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y}; }

__device__ __forceinline__ void cost65_accumulate_f16x2(const float* __restrict__ A, const float* __restrict__ B,
                                                        int D, float* edge_lds, int lane, Cost65Acc& o,
                                                        int ld = C65_NT) {
    const int li = lane & 31, kg = lane >> 5;
    constexpr int NB = C65_NB;
    f32x16 c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
    float er0 = 0.f, er1 = 0.f, ec0 = 0.f, ec1 = 0.f, cn = 0.f;
    float* eA = edge_lds;
    float* eB = edge_lds + 512;
    const float* pa = A + (8 * kg) * ld + 2 * li;
    const float* pb = B + (8 * kg) * ld + 2 * li;
    struct Raw { f2u a[8], b[8]; };
    auto load_blk = [&](int k0, Raw& q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            q.a[e] = *reinterpret_cast<const f2u*>(pa + (k0 + e) * ld);
            q.b[e] = *reinterpret_cast<const f2u*>(pb + (k0 + e) * ld);
        }
    };
    // One raw block in flight: a step converts its block to the eight fp16 operands and runs the edge chains, THEN
    // issues the next block's loads into the same registers, THEN the twelve MFMAs - accumulators (64) + operands
    // (32) + landing loads (32) stay inside the 3-waves-per-SIMD register budget; a second raw buffer spilled.
    const int nstep = D / 16;
    Raw q;
    load_blk(0, q);
    // Edge columns (node 64 of either side) -> LDS, 128 channels per round: all of a round's loads are issued behind
    // block 0's, and only then waited for - one memory round trip for the whole prologue (a plain k-loop was unrolled
    // by the compiler into three load-wait-store rounds, each a full latency, ahead of block 0's loads).
#pragma unroll 1
    for (int k = lane; k < D; k += 128) {
        const bool two = k + 64 < D;
        const int k1 = two ? k + 64 : k;                 // branch-free: a conditional load would be sunk behind a wait
        const float a0 = A[k * ld + NB], b0 = B[k * ld + NB];
        const float a1 = A[k1 * ld + NB], b1 = B[k1 * ld + NB];
        eA[k] = a0; eB[k] = b0;
        eA[k1] = a1; eB[k1] = b1;
    }
    wg_barrier();                                     // eA / eB visible
    for (int s = 0; s < nstep; ++s) {
        const int k0 = 16 * s;
        h2v ah[2][4], al[2][4], bh[2][4], bl[2][4];              // [column parity][pair of channels]
#pragma unroll
        for (int ep = 0; ep < 4; ++ep) {
            split_pair(q.a[2 * ep].x, q.a[2 * ep + 1].x, ah[0][ep], al[0][ep]);
            split_pair(q.a[2 * ep].y, q.a[2 * ep + 1].y, ah[1][ep], al[1][ep]);
            split_pair(q.b[2 * ep].x, q.b[2 * ep + 1].x, bh[0][ep], bl[0][ep]);
            split_pair(q.b[2 * ep].y, q.b[2 * ep + 1].y, bh[1][ep], bl[1][ep]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ea = eA[k0 + 8 * kg + e], eb = eB[k0 + 8 * kg + e];
            er0 = fmaf(ea, q.b[e].x, er0);
            er1 = fmaf(ea, q.b[e].y, er1);
            ec0 = fmaf(q.a[e].x, eb, ec0);
            ec1 = fmaf(q.a[e].y, eb, ec1);
            cn = fmaf(ea, eb, cn);
        }
        const h8v Aeh = pack8(ah[0][0], ah[0][1], ah[0][2], ah[0][3]), Ael = pack8(al[0][0], al[0][1], al[0][2], al[0][3]);
        const h8v Aoh = pack8(ah[1][0], ah[1][1], ah[1][2], ah[1][3]), Aol = pack8(al[1][0], al[1][1], al[1][2], al[1][3]);
        const h8v Beh = pack8(bh[0][0], bh[0][1], bh[0][2], bh[0][3]), Bel = pack8(bl[0][0], bl[0][1], bl[0][2], bl[0][3]);
        const h8v Boh = pack8(bh[1][0], bh[1][1], bh[1][2], bh[1][3]), Bol = pack8(bl[1][0], bl[1][1], bl[1][2], bl[1][3]);
        load_blk(s + 1 < nstep ? k0 + 16 : 0, q);        // wrap: a harmless reload at the end
        __builtin_amdgcn_sched_barrier(0);               // all eight operands complete before the first MFMA issues
#ifdef PATS_EXP_PREFENCE                                 // (round-5 experiment, diagnostic builds: the fence on BOTH sides of the block)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        asm volatile("" :: "v"(Aeh), "v"(Ael), "v"(Aoh), "v"(Aol), "v"(Beh), "v"(Bel), "v"(Boh), "v"(Bol));
        __builtin_amdgcn_sched_barrier(0);
#endif
        // small terms first, so that the dominant hi.hi product is added last to each accumulator
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ael, Beh, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ael, Boh, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aol, Beh, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aol, Boh, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aeh, Bel, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aeh, Bol, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aoh, Bel, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aoh, Bol, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aeh, Beh, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aeh, Boh, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aoh, Beh, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aoh, Boh, c11, 0, 0, 0);
        // HAZARD (ROCm 7.2, gfx950): left to itself hipcc interleaves the next conversions - VALU writes into registers
        // that are still the A / B operands of v_mfma_f32_32x32x16_f16 instructions issued a few slots earlier - with
        // those MFMAs, and the wait states it inserts do not cover that write-after-read: with three waves per SIMD
        // queueing on the matrix pipe 100-600 of 65 536 centre rows came out wrong, different ones in every run (the
        // score-matrix checksums of a diagnostic build without the solve were exact).  The three scheduling barriers
        // keep the phases apart - all operands first, twelve MFMAs back to back, then a full instruction's worth of
        // wait states (32 cycles = the 8 passes of the last MFMA) before any operand register may be rewritten.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        asm volatile("" :: "v"(Aeh), "v"(Ael), "v"(Aoh), "v"(Aol), "v"(Beh), "v"(Bel), "v"(Boh), "v"(Bol));
        __builtin_amdgcn_sched_barrier(0);
    }
    wg_barrier();                                     // done with eA / eB: the caller may reuse the LDS
    o.er0 = er0 + __shfl_xor(er0, 32); o.er1 = er1 + __shfl_xor(er1, 32);
    o.ec0 = ec0 + __shfl_xor(ec0, 32); o.ec1 = ec1 + __shfl_xor(ec1, 32);
    o.cn = cn + __shfl_xor(cn, 32);
    constexpr float UNSCALE = 1.0f / (C65_PRESCALE * C65_PRESCALE);            // exact power of two
    o.c00 = c00 * UNSCALE; o.c01 = c01 * UNSCALE; o.c10 = c10 * UNSCALE; o.c11 = c11 * UNSCALE;
}

struct Cost65Scale {
    float d, rd;                 // D**.5 and its IEEE reciprocal
    __device__ __forceinline__ explicit Cost65Scale(int D) : d(sqrtf((float)D)), rd(1.0f / sqrtf((float)D)) {}
};
__device__ __forceinline__ float cost65_scale(float x, const Cost65Scale& k) { return 0.1f * div_invariant(x, k.d, k.rd); }

// accumulate + write the scaled 65x65 matrix into a row-major LDS tile (row stride 65)
template <bool F16 = false>
__device__ __forceinline__ void cost65_to_tile(const float* __restrict__ A, const float* __restrict__ B,
                                               int D, float* tile, int lane) {
    constexpr int NT = C65_NT, NB = C65_NB;
    Cost65Acc c;
    if (F16) cost65_accumulate_f16x2(A, B, D, tile, lane, c);
    else cost65_accumulate(A, B, D, tile, lane, c);
    const int li = lane & 31, lk = lane >> 5;
    const Cost65Scale sq(D);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rc = (r & 3) + 8 * (r >> 2) + 4 * lk;
        tile[(2 * rc) * NT + 2 * li] = cost65_scale(c.c00[r], sq);
        tile[(2 * rc) * NT + 2 * li + 1] = cost65_scale(c.c01[r], sq);
        tile[(2 * rc + 1) * NT + 2 * li] = cost65_scale(c.c10[r], sq);
        tile[(2 * rc + 1) * NT + 2 * li + 1] = cost65_scale(c.c11[r], sq);
    }
    if (lk == 0) {
        tile[NB * NT + 2 * li] = cost65_scale(c.er0, sq);
        tile[NB * NT + 2 * li + 1] = cost65_scale(c.er1, sq);
        tile[(2 * li) * NT + NB] = cost65_scale(c.ec0, sq);
        tile[(2 * li + 1) * NT + NB] = cost65_scale(c.ec1, sq);
    }
    if (lane == 0) tile[NB * NT + NB] = cost65_scale(c.cn, sq);
}

}  // namespace pats
