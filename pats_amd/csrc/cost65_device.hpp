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
    __syncthreads();                                     // eA / eB visible
    for (int blk = 0; blk < nblk; blk += 4) {
        load_blk(k0_of(blk + 3), r3); compute_blk(k0_of(blk + 0), r0);
        load_blk(k0_of(blk + 4), r0); compute_blk(k0_of(blk + 1), r1);
        load_blk(k0_of(blk + 5), r1); compute_blk(k0_of(blk + 2), r2);
        load_blk(k0_of(blk + 6), r2); compute_blk(k0_of(blk + 3), r3);
    }
    __syncthreads();                                     // done with eA / eB: the caller may reuse the LDS
    o.er0 = er0 + __shfl_xor(er0, 32); o.er1 = er1 + __shfl_xor(er1, 32);
    o.ec0 = ec0 + __shfl_xor(ec0, 32); o.ec1 = ec1 + __shfl_xor(ec1, 32);
    o.cn = cn + __shfl_xor(cn, 32);
    o.c00 = c00; o.c01 = c01; o.c10 = c10; o.c11 = c11;
}

struct Cost65Scale {
    float d, rd;                 // D**.5 and its IEEE reciprocal
    __device__ __forceinline__ explicit Cost65Scale(int D) : d(sqrtf((float)D)), rd(1.0f / sqrtf((float)D)) {}
};
__device__ __forceinline__ float cost65_scale(float x, const Cost65Scale& k) { return 0.1f * div_invariant(x, k.d, k.rd); }

// accumulate + write the scaled 65x65 matrix into a row-major LDS tile (row stride 65)
__device__ __forceinline__ void cost65_to_tile(const float* __restrict__ A, const float* __restrict__ B,
                                               int D, float* tile, int lane) {
    constexpr int NT = C65_NT, NB = C65_NB;
    Cost65Acc c;
    cost65_accumulate(A, B, D, tile, lane, c);
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
