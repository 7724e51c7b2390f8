// Transposed reductions for an 8x8 lane grid (lane = 8*I + J) on a 64-lane wave.
//
// Each lane brings 8 partial values p[0..7]; value k must be reduced over the 8 lanes that share
// I (consecutive lanes 8I..8I+7) or over the 8 lanes that share J (lanes J, J+8, ..., J+56), and
// the 8 results are to end up one per lane.  A butterfly that halves the number of live values at
// every level does this in 4+2+1 = 7 exchange-and-combine steps instead of 8 separate 3-level
// reductions:
//   reduce8_consecutive: result for index J lands in lane (I, J)      (row_half_mirror, quad xor 2, xor 1)
//   reduce8_strided    : result for index I lands in lane (I, J)      (permlane32_swap, permlane16_swap, row_ror:8)
#pragma once
#include "common.hpp"

namespace pats {

// experiment hook (tools/third_determinism.py): wait states on both sides of every v_permlane*_swap
#ifdef PATS_EXP_SWAP_NOPS
#define PATS_SWAP_FENCE(x, y) asm volatile("s_nop 1" : "+v"(x), "+v"(y))
#else
#define PATS_SWAP_FENCE(x, y) do { } while (0)
#endif

constexpr int DPP_ROW_ROR8 = 0x128;      // row_ror:8  == lane ^ 8 inside a 16-lane row

template <class Op>
__device__ __forceinline__ float reduce8_consecutive(const float (&p)[8], Op op, int lane) {
    // Each level forms BOTH candidate sums with the DPP operand taken from a register that was
    // written a level earlier, then selects - so no DPP instruction reads a just-written VGPR (each
    // such read costs two wait states) and the independent ops of a level interleave freely.
    const bool hi = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float lo4[4], hi4[4], r[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) lo4[t] = op(p[t], dpp_f<DPP_ROW_HALF_MIRROR>(p[t]));          // partner = lane 7 - J
#pragma unroll
    for (int t = 0; t < 4; ++t) hi4[t] = op(p[4 + t], dpp_f<DPP_ROW_HALF_MIRROR>(p[4 + t]));
#pragma unroll
    for (int t = 0; t < 4; ++t) r[t] = hi ? hi4[t] : lo4[t];
    float lo2[2], hi2[2], q[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) lo2[s] = op(r[s], dpp_f<DPP_QUAD_XOR2>(r[s]));
#pragma unroll
    for (int s = 0; s < 2; ++s) hi2[s] = op(r[2 + s], dpp_f<DPP_QUAD_XOR2>(r[2 + s]));
#pragma unroll
    for (int s = 0; s < 2; ++s) q[s] = b1 ? hi2[s] : lo2[s];
    const float x = op(q[0], dpp_f<DPP_QUAD_XOR1>(q[0])), y = op(q[1], dpp_f<DPP_QUAD_XOR1>(q[1]));
    return b0 ? y : x;
}

// in-place exchange of two registers between wave halves / 16-lane rows (common.hpp: lane_swap32 / lane_swap16)
__device__ __forceinline__ void swap32(float& x, float& y) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = __builtin_bit_cast(unsigned, y);
    PATS_SWAP_FENCE(a, b);
    lane_swap32(a, b);
    PATS_SWAP_FENCE(a, b);
    x = __builtin_bit_cast(float, a);
    y = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ void swap16(float& x, float& y) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = __builtin_bit_cast(unsigned, y);
    PATS_SWAP_FENCE(a, b);
    lane_swap16(a, b);
    PATS_SWAP_FENCE(a, b);
    x = __builtin_bit_cast(float, a);
    y = __builtin_bit_cast(float, b);
}

template <class Op>
__device__ __forceinline__ float reduce8_strided(const float (&p)[8], Op op, int lane) {
    float r[4], q[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {           // lanes < 32 end with index t, lanes >= 32 with index 4 + t
        float x = p[t], y = p[4 + t];
        swap32(x, y);
        r[t] = op(x, y);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {           // even 16-rows: index s, odd 16-rows: index 2 + s
        float x = r[s], y = r[2 + s];
        swap16(x, y);
        q[s] = op(x, y);
    }
    const bool b3 = lane & 8;
    const float keep = b3 ? q[1] : q[0], send = b3 ? q[0] : q[1];
    return op(keep, dpp_f<DPP_ROW_ROR8>(send));
}

// value of lane (lane ^ X), X < 32: ds_swizzle bit-mask mode (and = 0x1f, or = 0, xor = X)
template <int X>
__device__ __forceinline__ float swz_xor(float v) {
    float r = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (X << 10) | 0x1f));
#ifdef PATS_EXP_XBAR_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r));      // experiment: every crossbar result is waited for on its own
#endif
    return r;
}

// 64-lane sum into every lane with the two cross-row steps on the LDS crossbar: four in-row DPP adds, lane ^ 16 by
// ds_swizzle_b32, lane ^ 32 by ds_bpermute_b32 (neither touches LDS memory nor takes a VALU issue slot) - six VALU
// instructions where the row-broadcast form (wave_sum_uniform) needs twelve.  Every step is a symmetric exchange, so all
// 64 lanes end with the same bits.
__device__ __forceinline__ float wave_sum_xbar(float v, int lane) {
    v += dpp_f<DPP_QUAD_XOR1>(v);
    v += dpp_f<DPP_QUAD_XOR2>(v);
    v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f<DPP_ROW_MIRROR>(v);
    v += swz_xor<16>(v);
    float r = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
#ifdef PATS_EXP_XBAR_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r));
#endif
    v += r;
    return v;
}

}  // namespace pats
