// 160 x 160 output tile of a channel-major product  C[i][j] = sum_k A[k][i] B[k][j]  on the matrix cores of gfx950,
// shared by the cost build (cost.hip: A, B = the two descriptor sets) and the 1x1 convolutions of the GNN layers
// (gnn.hip: A = transposed weights, B = activations over the flattened (batch, token) axis).
//
// One 256-thread workgroup = 5 x 5 MFMA tiles of 32 x 32: wave w computes tile row w (five tiles) plus tile (4, w),
// wave 0 also (4, 4) - one wave per SIMD, two workgroups per CU.  `row4 == false` (fewer than 129 output rows in
// this tile) skips the fifth tile row.  Two contraction paths:
//
//  * tile_split.  Every fp32 operand is split into two fp16 halves, x * 2^6 = hi + lo (both round-to-nearest, so the
//    pair carries 22 mantissa bits), and each tile takes THREE exact-product passes of v_mfma_f32_32x32x16_f16 per
//    16 channels (lo.hi + hi.lo + hi.hi, fp32 accumulation; lo.lo <= 2^-22 |x y| is dropped): 6 matrix-pipe cycles
//    per channel and tile instead of the 32 of v_mfma_f32_32x32x2_f32, whose multiplies run on the vector FMA lanes.
//    A chunk of 16 channels x 160 columns of each side is fetched global -> registers by the operand source (four
//    consecutive channels of one column per item, coalesced along the columns), split there (v_mul,
//    v_cvt_pk_f16_f32, v_fma_mix_f32, v_cvt_pk_f16_f32: six VALU instructions per pair) and parked in LDS already
//    in MFMA fragment order ([channel / 4][column] x 8 bytes: conflict-free 8-byte writes and fragment reads),
//    double-buffered, one barrier per chunk.  Measured against float64 the split products are closer than the fp32
//    fma chain's (tools/cost_ab.py).  The fp16 range is the price: |x| > 1023 makes hi infinite and with it every
//    output of that row / column inf or NaN - tile_split returns false (workgroup-uniform) when it finds a
//    non-finite value among the outputs the caller is about to store, and the caller REDOES the tile with tile_f32
//    (no flag buffer, no second launch, no host involvement; inputs that are themselves inf / NaN take the same
//    route and come out as the fp32 chain has them).  fp16 subnormals are flushed by the matrix pipe; the 2^6
//    prescale keeps `lo` normal down to |x| = 0.004, below which an operand is carried with an absolute error
//    <= 2^-20 (at most 1e-6 |y| in one product).  Accumulators come out scaled by 2^12 (UNSCALE undoes it exactly).
//
//  * tile_f32 (v_mfma_f32_32x32x2_f32; bitwise a k-ordered fmaf chain): slabs of 8 channels x 160 columns per
//    side, staged global -> registers -> LDS, double-buffered.
//
// Operands come from an operand source, one object per thread (CmSrc below: two channel-major tensors with a
// per-column base offset; the staging maps are fixed here, the column addressing is the caller's):
//   int  K                                                       channels of this source
//   void fetch_f32(int k0, float (&ra)[CPT], float (&rb)[CPT])   element q of a slab = (channel k0 + (t + 256 q) / 160,
//                                                                tile column (t + 256 q) % 160); zero beyond K
//   void fetch_split(int k0, float (&r)[SQ][4])                  item q: id = t + 256 q, tile column id % 160,
//                                                                side (id / 160) >> 2, channels k0 + 4 ((id / 160) & 3) + e
//   bool row_stored(int r) / col_stored(int c)                   tile row / column inside the output
// Calls arrive with k0 = 0, step, 2 step, ... (k0 = 0 restarts), so a source keeps running pointers.  tile() takes
// one or two sources: the reduction may run over two tensors one after the other (x | message in the GNN's MLP -
// `cat` is never materialised; the first source's ragged last chunk is zero-filled, so any channel count works).
//
// Tried and measured slower for the cost build (tools/cost_ab.py, 20 736 x [264,145]^2: tile_f32 3.04 ms, tile_split
// 1.98 ms): 320-thread workgroups with one tile row per wave (balanced, 80 accumulator registers, but two 5-wave
// workgroups do not pack onto four SIMDs: 2.5-3.1 ms with dword loads, 16-byte loads or 32-channel chunks alike).
#pragma once
#include "common.hpp"

namespace pats {
namespace mt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h2c __attribute__((ext_vector_type(2)));
typedef _Float16 h8c __attribute__((ext_vector_type(8)));
typedef float f2c __attribute__((ext_vector_type(2)));
typedef unsigned u4c __attribute__((ext_vector_type(4)));

constexpr int CT = 160, KC = 8, CPT = KC * CT / 256;      // workgroup tile edge; fp32 path: k-chunk, floats per thread per slab
constexpr int SKC = 16, SQ = 2 * SKC * CT / (4 * 256);    // split path: channels per chunk, 4-channel items per thread (5)
constexpr float PRESCALE = 64.0f;                         // 1023 * 64 < 65504, the largest fp16
constexpr float UNSCALE = 1.0f / (PRESCALE * PRESCALE);   // exact power of two

struct __attribute__((aligned(16))) LdsF32 {
    float a[2][KC][CT];
    float b[2][KC][CT];
};
struct __attribute__((aligned(16))) LdsSplit {
    uint2 v[2][2][2][4][CT];          // [buffer][side][hi | lo][channel / 4][column]: four fp16 (channels 4 q .. 4 q + 3)
};
union Lds {
    LdsF32 f;
    LdsSplit s;
};

// staging-map helpers for the sources
__device__ __forceinline__ int f32_row(int t, int q) { return (t + 256 * q) / CT; }
__device__ __forceinline__ int f32_col(int t, int q) { return (t + 256 * q) % CT; }
__device__ __forceinline__ int item_col(int t, int q) { return (t + 256 * q) % CT; }
__device__ __forceinline__ int item_side(int t, int q) { return ((t + 256 * q) / CT) >> 2; }
__device__ __forceinline__ int item_quad(int t, int q) { return ((t + 256 * q) / CT) & 3; }

__device__ __forceinline__ void clear(f32x16 (&acc)[7]) {
#pragma unroll
    for (int q = 0; q < 7; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
}

// fp32 contraction of one 160 x 160 tile over the channels of one source, ADDED to acc
template <bool ROW4, class Src>
__device__ __forceinline__ void tile_f32(Src& src, LdsF32& lds, f32x16 (&acc)[7], bool row4_, int t, int wave) {
    const bool row4 = ROW4 && row4_;
    const int K = src.K;
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    float ra[CPT], rb[CPT];
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int e = t + 256 * q;
            (&lds.a[buf][0][0])[e] = ra[q];
            (&lds.b[buf][0][0])[e] = rb[q];
        }
    };
    const int nchunk = (K + KC - 1) / KC;
    src.fetch_f32(0, ra, rb);
    stash(0);
    wg_barrier();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) src.fetch_f32((c + 1) * KC, ra, rb);
#pragma unroll
        for (int kp = 0; kp < KC / 2; ++kp) {
            const float* ar = &lds.a[buf][2 * kp + lk][0];
            const float* br = &lds.b[buf][2 * kp + lk][0];
            const float aw = ar[32 * wave + li], a4 = ar[128 + li], bw = br[32 * wave + li];
            float bf[5];
#pragma unroll
            for (int tj = 0; tj < 5; ++tj) bf[tj] = br[32 * tj + li];
#pragma unroll
            for (int tj = 0; tj < 5; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bf[tj], acc[tj], 0, 0, 0);
            if (row4) {                                 // workgroup-uniform
                acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4, bw, acc[5], 0, 0, 0);
                if (wave == 0) acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4, bf[4], acc[6], 0, 0, 0);
            }
        }
        if (c + 1 < nchunk) stash(buf ^ 1);        // last read in chunk c - 1; every wave is past that barrier
        wg_barrier();
    }
}

// x * 2^6 = hi + lo for two consecutive channels of one column.  Scalar multiplies / fmas on purpose (build with
// -fno-slp-vectorize): packed fp32 math would first have to move the operands into adjacent registers.
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const h2c h = __builtin_convertvector(f2c{x0 * PRESCALE, x1 * PRESCALE}, h2c);                 // v_cvt_pk_f16_f32, RNE
    const float r0 = fmaf(x0, PRESCALE, -(float)h.x), r1 = fmaf(x1, PRESCALE, -(float)h.y);        // exact
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2c{r0, r1}, h2c));
}

// fp16-split contraction of one tile over the channels of one source, ADDED to acc
template <bool ROW4, class Src>
__device__ __forceinline__ void split_accumulate(Src& src, LdsSplit& lds, f32x16 (&acc)[7], bool row4_, int t, int wave) {
    const bool row4 = ROW4 && row4_;
    const int K = src.K;
    const int lane = t & 63, li = lane & 31, kg = lane >> 5;
    uint2* dst[SQ];
#pragma unroll
    for (int q = 0; q < SQ; ++q) dst[q] = &lds.v[0][item_side(t, q)][0][item_quad(t, q)][item_col(t, q)];
    constexpr int BUF = sizeof(lds.v[0]) / sizeof(uint2), HL = sizeof(lds.v[0][0][0]) / sizeof(uint2);
    auto stash = [&](int buf, const float (&r)[SQ][4]) {
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
            uint2 hi, lo;
            split2(r[q][0], r[q][1], hi.x, lo.x);
            split2(r[q][2], r[q][3], hi.y, lo.y);
            dst[q][buf * BUF] = hi;
            dst[q][buf * BUF + HL] = lo;
        }
    };
    auto frag = [&](int buf, int side, int hl, int col) {
        const uint2 e0 = lds.v[buf][side][hl][2 * kg][col], e1 = lds.v[buf][side][hl][2 * kg + 1][col];
        return __builtin_bit_cast(h8c, u4c{e0.x, e0.y, e1.x, e1.y});
    };
    auto mfma_chunk = [&](int buf) {
        // Phases kept apart as in cost65_accumulate_f16x2 (cost65_device.hpp): the VALU-heavy split of the next chunk
        // starts one instruction's worth of wait states after the last MFMA - the compiler's own wait states did not
        // cover a VALU write into an operand register of an MFMA still queueing on the matrix pipe (measured there).
        __builtin_amdgcn_sched_barrier(0);
        const h8c awh = frag(buf, 0, 0, 32 * wave + li), awl = frag(buf, 0, 1, 32 * wave + li);
#pragma unroll
        for (int tj = 0; tj < 5; ++tj) {
            const h8c bh = frag(buf, 1, 0, 32 * tj + li), bl = frag(buf, 1, 1, 32 * tj + li);
            acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(awl, bh, acc[tj], 0, 0, 0);
            acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(awh, bl, acc[tj], 0, 0, 0);
            acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(awh, bh, acc[tj], 0, 0, 0);
            if (tj == 4) {
                if (row4 && wave == 0) {
                    const h8c a4h = frag(buf, 0, 0, 128 + li), a4l = frag(buf, 0, 1, 128 + li);
                    acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4l, bh, acc[6], 0, 0, 0);
                    acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, bl, acc[6], 0, 0, 0);
                    acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, bh, acc[6], 0, 0, 0);
                    asm volatile("" :: "v"(a4h), "v"(a4l));
                }
                asm volatile("" :: "v"(bh), "v"(bl));
            }
        }
        if (row4) {
            const h8c a4h = frag(buf, 0, 0, 128 + li), a4l = frag(buf, 0, 1, 128 + li);
            const h8c bh = frag(buf, 1, 0, 32 * wave + li), bl = frag(buf, 1, 1, 32 * wave + li);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4l, bh, acc[5], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, bl, acc[5], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, bh, acc[5], 0, 0, 0);
            asm volatile("" :: "v"(a4h), "v"(a4l), "v"(bh), "v"(bl));
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        asm volatile("" :: "v"(awh), "v"(awl));
        __builtin_amdgcn_sched_barrier(0);
    };
    const int nchunk = (K + SKC - 1) / SKC;
    // (a second register set with the fetch running two chunks ahead was measured 2 % slower on the GNN products and
    // does not fit the register budget of the cost build)
    float r[SQ][4];
    src.fetch_split(0, r);
    stash(0, r);
    wg_barrier();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) src.fetch_split((c + 1) * SKC, r);
        mfma_chunk(buf);
        if (c + 1 < nchunk) stash(buf ^ 1, r);       // last read in chunk c - 1; every wave is past that barrier
        wg_barrier();
    }
}

// An operand beyond the fp16 range became an infinite hi half: every output of its row / column is then inf or NaN
// (inf - inf from the lo.hi pass, or inf * 0).  Only outputs that will be stored count.  Workgroup-uniform result.
template <bool ROW4, class Src>
__device__ __forceinline__ bool split_finite(const Src& src, const f32x16 (&acc)[7], bool row4_, int t, int wave) {
    const bool row4 = ROW4 && row4_;
    const int lane = t & 63, li = lane & 31, kg = lane >> 5;
    bool bad = false;
    auto scan = [&](const f32x16& c, int ti, int tj) {
        const bool colin = src.col_stored(32 * tj + li);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const bool rowin = src.row_stored(32 * ti + (q & 3) + 8 * (q >> 2) + 4 * kg);
            bad |= colin && rowin && !(fabsf(c[q]) <= 3.0e38f);
        }
    };
#pragma unroll
    for (int tj = 0; tj < 5; ++tj) scan(acc[tj], wave, tj);
    if (row4) {
        scan(acc[5], 4, wave);
        if (wave == 0) scan(acc[6], 4, 4);
    }
    return !wg_barrier_or(bad);
}

// Two channel-major operands with caller-defined column offsets: element (channel k, tile column c) of side A is
// A[cols.a_off(c) + k * ldA], of side B  B[cols.b_off(c) + k * ldB]; optionally x <- max(0, x * scale[k] + shift[k]) on
// side B while staging.  `Cols` also says which tile rows / columns are stored.  Running pointers, advanced per chunk;
// only a ragged last chunk takes the zero-filling fetch (a valid channel is read, then zeroed: no predicated loads).
template <class Cols>
struct CmSrc {
    const float *A, *B;
    int ldA, ldB, K, t;
    const Cols& cols;
    const float *scale, *shift;
    const float* p[SQ];                // split items, or side-A slab elements of the fp32 path
    const float* pq[CPT];              // side-B slab elements of the fp32 path

    __device__ __forceinline__ CmSrc(const float* A_, int ldA_, const float* B_, int ldB_, int K_, const Cols& c, int t_,
                                     const float* scale_ = nullptr, const float* shift_ = nullptr)
        : A(A_), B(B_), ldA(ldA_), ldB(ldB_), K(K_), t(t_), cols(c), scale(scale_), shift(shift_) {}
    __device__ __forceinline__ bool row_stored(int r) const { return cols.row_stored(r); }
    __device__ __forceinline__ bool col_stored(int c) const { return cols.col_stored(c); }
    __device__ __forceinline__ float affine(float x, int k) const { return fmaxf(fmaf(x, scale[k], shift[k]), 0.f); }

    __device__ __forceinline__ void fetch_f32(int k0, float (&ra)[CPT], float (&rb)[CPT]) {
        if (k0 == 0) {
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                p[q] = A + cols.a_off(f32_col(t, q)) + (int64_t)f32_row(t, q) * ldA;
                pq[q] = B + cols.b_off(f32_col(t, q)) + (int64_t)f32_row(t, q) * ldB;
            }
        }
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int k = k0 + f32_row(t, q);
            const bool kin = k0 + KC <= K || k < K;
            ra[q] = kin ? *p[q] : 0.f;
            float x = kin ? *pq[q] : 0.f;
            if (scale && kin) x = affine(x, k);
            rb[q] = x;
            p[q] += (int64_t)KC * ldA;
            pq[q] += (int64_t)KC * ldB;
        }
    }

    __device__ __forceinline__ void fetch_split(int k0, float (&r)[SQ][4]) {
        int ld[SQ];
#pragma unroll
        for (int q = 0; q < SQ; ++q) ld[q] = item_side(t, q) ? ldB : ldA;
        if (k0 == 0) {
#pragma unroll
            for (int q = 0; q < SQ; ++q) {
                const int col = item_col(t, q);
                p[q] = (item_side(t, q) ? B + cols.b_off(col) : A + cols.a_off(col)) + (int64_t)(4 * item_quad(t, q)) * ld[q];
            }
        }
        if (k0 + SKC <= K) {
#pragma unroll
            for (int q = 0; q < SQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) r[q][e] = Cols::stream ? __builtin_nontemporal_load(p[q] + e * ld[q]) : p[q][e * ld[q]];
        } else {
#pragma unroll
            for (int q = 0; q < SQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + 4 * item_quad(t, q) + e, back = min(k, K - 1) - (k - e);
                    const float v = p[q][back * ld[q]];
                    r[q][e] = k < K ? v : 0.f;
                }
        }
        if (scale) {
#pragma unroll
            for (int q = 0; q < SQ; ++q)
                if (item_side(t, q)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = k0 + 4 * item_quad(t, q) + e;
                        if (k < K) r[q][e] = affine(r[q][e], k);
                    }
                }
        }
#pragma unroll
        for (int q = 0; q < SQ; ++q) p[q] += SKC * ld[q];
    }
};

// Both paths behind one call: accumulators and the factor the epilogue has to apply to them.  `s1` (may be null): a
// second source whose channels continue the reduction.
// ROW4 = false: the caller never has a fifth tile row (at most 128 output rows) - acc[5], acc[6] cost no registers
template <bool SPLIT, bool ROW4, class Src>
__device__ __forceinline__ float tile(Src& s0, Src* s1, Lds& lds, f32x16 (&acc)[7], bool row4, int t, int wave) {
    clear(acc);
    if (SPLIT) {
        split_accumulate<ROW4>(s0, lds.s, acc, row4, t, wave);
        if (s1) split_accumulate<ROW4>(*s1, lds.s, acc, row4, t, wave);
        if (split_finite<ROW4>(s0, acc, row4, t, wave)) return UNSCALE;
        clear(acc);
    }
    tile_f32<ROW4>(s0, lds.f, acc, row4, t, wave);
    if (s1) tile_f32<ROW4>(*s1, lds.f, acc, row4, t, wave);
    return 1.0f;
}

}  // namespace mt
}  // namespace pats
