// Descriptor x descriptor cost build on the matrix cores of gfx950.
//
// Replaces  scores = einsum('bdn,bdm->bnm', mdesc0, mdesc1) / D**.5 ;  0.1 * scores
//   models/first_layer.py:110-111,114 (D=448, 300x300)   second_layer.py:100-101,104 (D=264,
//   145x145 x B)   third_layer.py:156-158 (D=128, 65x65 x P)
//
// One 256-thread workgroup owns a 160 x 160 output tile = 5 x 5 MFMA tiles of 32 x 32: wave w computes tile
// row w (five tiles) plus tile (4, w), wave 0 also (4, 4) - so a fine-level problem (145 x 145) is exactly one
// workgroup with no idle wave, and a workgroup is one wave per SIMD (two workgroups per CU).  Operands are
// channel-major ([D][n]); columns past the matrix edge are CLAMPED, not predicated (they only feed output rows /
// columns that are never stored, and the main loop stays free of branches).  Scale and the reference's two-step
// `/ sqrt(D)`, `* 0.1` rounding are applied in the epilogue.
//
// Two contraction paths share the tile map and the epilogue:
//
//  * split path (default).  Every fp32 operand is split into two fp16 halves, x * 2^6 = hi + lo (both
//    round-to-nearest, so the pair carries 22 mantissa bits), and each tile takes THREE exact-product passes of
//    v_mfma_f32_32x32x16_f16 per 16 channels (lo.hi + hi.lo + hi.hi, fp32 accumulation; lo.lo <= 2^-22 |x y| is
//    dropped): 6 matrix-pipe cycles per channel and tile instead of the 32 of v_mfma_f32_32x32x2_f32, whose
//    multiplies run on the vector FMA lanes.  A chunk of 16 channels x 160 columns of each side is fetched
//    global -> registers (four consecutive channels of one column per item, coalesced along the columns),
//    split there (v_mul, v_cvt_pk_f16_f32, v_fma_mix_f32, v_cvt_pk_f16_f32: six VALU instructions per pair) and
//    parked in LDS already in MFMA fragment order ([channel / 4][column] x 8 bytes: conflict-free 8-byte writes
//    and fragment reads), double-buffered, one barrier per chunk.  Measured against float64 the split scores are
//    closer than the fp32 fma chain's (tools/cost_ab.py).  The fp16 range is the price: |x| > 1023 makes hi
//    infinite and with it every output of that row / column inf or NaN - a workgroup that finds a non-finite value
//    among the outputs it is about to store REDOES its tile on the fp32 path below (workgroup-uniform branch; no
//    flag buffer, no second launch, no host involvement; inputs that are themselves inf / NaN take the same route
//    and come out as the fp32 chain has them).  fp16 subnormals are flushed by the matrix pipe; the 2^6 prescale
//    keeps `lo` normal down to |x| = 0.004, below which an operand is carried with an absolute error <= 2^-20
//    (at most 1e-6 |y| in one product).
//
//  * fp32 path (v_mfma_f32_32x32x2_f32; bitwise a k-ordered fmaf chain): the fallback above, and the whole kernel
//    under PATS_COST_F32=1 (ablation / A-B runs).  Slabs of 8 descriptor rows x 160 columns per side, staged
//    global -> registers -> LDS, double-buffered.
//
// Tried and measured slower (tools/cost_ab.py, 20 736 x [264,145]^2: fp32 path 3.04 ms, this kernel 2.15 ms):
// 320-thread workgroups with one tile row per wave (balanced, 80 accumulator registers, but two 5-wave workgroups
// do not pack onto four SIMDs: 2.5-3.1 ms with dword loads, 16-byte loads or 32-channel chunks alike).
#include "common.hpp"

#include <cstdlib>

namespace pats {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h2c __attribute__((ext_vector_type(2)));
typedef _Float16 h8c __attribute__((ext_vector_type(8)));
typedef float f2c __attribute__((ext_vector_type(2)));
typedef unsigned u4c __attribute__((ext_vector_type(4)));

constexpr int CT = 160, KC = 8, CPT = KC * CT / 256;      // workgroup tile edge; fp32 path: k-chunk, floats per thread per slab
constexpr int SKC = 16, SQ = 2 * SKC * CT / (4 * 256);    // split path: channels per chunk, 4-channel items per thread (5)
constexpr float COST_PRESCALE = 64.0f;                    // 1023 * 64 < 65504, the largest fp16

struct __attribute__((aligned(16))) CostLdsF32 {
    float a[2][KC][CT];
    float b[2][KC][CT];
};
struct __attribute__((aligned(16))) CostLdsSplit {
    uint2 v[2][2][2][4][CT];          // [buffer][side][hi | lo][channel / 4][column]: four fp16 (channels 4 q .. 4 q + 3)
};
union CostLds {
    CostLdsF32 f;
    CostLdsSplit s;
};

struct CostTile {
    const float *A, *B;
    int D, n, m, i0, j0;
};

// fp32 contraction of one 160 x 160 tile
__device__ __forceinline__ void cost_tile_f32(const CostTile& g, CostLdsF32& lds, f32x16 (&acc)[7], int t, int wave) {
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    // staging map: element e = t + 256 q of a slab is (row e / 160, column e % 160).  Pointers are fixed per thread
    // and advance by KC descriptor rows per chunk; only a ragged last chunk (D % 8) needs the zero-filling fetch.
    float ra[CPT], rb[CPT];
    const float* pa[CPT];
    const float* pb[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int e = t + 256 * q, kk = e / CT, col = e % CT;
        pa[q] = g.A + (int64_t)kk * g.n + min(g.i0 + col, g.n - 1);
        pb[q] = g.B + (int64_t)kk * g.m + min(g.j0 + col, g.m - 1);
    }
    const int64_t stepa = (int64_t)KC * g.n, stepb = (int64_t)KC * g.m;
    auto fetch = [&](int k0) {
        if (k0 + KC <= g.D) {
#pragma unroll
            for (int q = 0; q < CPT; ++q) { ra[q] = *pa[q]; rb[q] = *pb[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const bool kin = k0 + (t + 256 * q) / CT < g.D;
                ra[q] = kin ? *pa[q] : 0.f;
                rb[q] = kin ? *pb[q] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < CPT; ++q) { pa[q] += stepa; pb[q] += stepb; }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int e = t + 256 * q;
            (&lds.a[buf][0][0])[e] = ra[q];
            (&lds.b[buf][0][0])[e] = rb[q];
        }
    };
    const int nchunk = (g.D + KC - 1) / KC;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) fetch((c + 1) * KC);
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
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4, bw, acc[5], 0, 0, 0);
            if (wave == 0) acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4, bf[4], acc[6], 0, 0, 0);
        }
        if (c + 1 < nchunk) stash(buf ^ 1);        // last read in chunk c - 1; every wave is past that barrier
        __syncthreads();
    }
}

// x * 2^6 = hi + lo for four consecutive channels of one column.  Scalar multiplies / fmas on purpose (the file is built
// with -fno-slp-vectorize): packed fp32 math would first have to move the operands into adjacent registers.
__device__ __forceinline__ void cost_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const h2c h = __builtin_convertvector(f2c{x0 * COST_PRESCALE, x1 * COST_PRESCALE}, h2c);       // v_cvt_pk_f16_f32, RNE
    const float r0 = fmaf(x0, COST_PRESCALE, -(float)h.x), r1 = fmaf(x1, COST_PRESCALE, -(float)h.y);   // exact
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2c{r0, r1}, h2c));
}

// fp16-split contraction of one tile; returns false (workgroup-uniform) if an operand left the fp16 range
__device__ __forceinline__ bool cost_tile_split(const CostTile& g, CostLdsSplit& lds, f32x16 (&acc)[7], int t, int wave) {
    const int lane = t & 63, li = lane & 31, kg = lane >> 5;
    // item id = t + 256 q -> column id % 160 and (side, channel quad) = id / 160: four consecutive channels of one column
    float r[SQ][4];
    const float* p[SQ];
    int ld[SQ], krow[SQ];
    uint2* dst[SQ];
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
        const int id = t + 256 * q, col = id % CT, rest = id / CT;
        const int side = rest >> 2, gq = rest & 3;
        krow[q] = 4 * gq;
        ld[q] = side ? g.m : g.n;
        p[q] = (side ? g.B + min(g.j0 + col, g.m - 1) : g.A + min(g.i0 + col, g.n - 1)) + (int64_t)krow[q] * ld[q];
        dst[q] = &lds.v[0][side][0][gq][col];
    }
    auto fetch = [&](int k0) {
        if (k0 + SKC <= g.D) {
#pragma unroll
            for (int q = 0; q < SQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) r[q][e] = p[q][e * ld[q]];
        } else {
#pragma unroll
            for (int q = 0; q < SQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {      // ragged last chunk: read a valid row, then zero (no predicated loads)
                    const int row = k0 + krow[q] + e, back = min(row, g.D - 1) - (k0 + krow[q]);
                    const float v = p[q][back * ld[q]];
                    r[q][e] = row < g.D ? v : 0.f;
                }
        }
#pragma unroll
        for (int q = 0; q < SQ; ++q) p[q] += SKC * ld[q];
    };
    constexpr int BUF = sizeof(lds.v[0]) / sizeof(uint2), HL = sizeof(lds.v[0][0][0]) / sizeof(uint2);
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
            uint2 hi, lo;
            cost_split2(r[q][0], r[q][1], hi.x, lo.x);
            cost_split2(r[q][2], r[q][3], hi.y, lo.y);
            dst[q][buf * BUF] = hi;
            dst[q][buf * BUF + HL] = lo;
        }
    };
    auto frag = [&](int buf, int side, int hl, int col) {
        const uint2 e0 = lds.v[buf][side][hl][2 * kg][col], e1 = lds.v[buf][side][hl][2 * kg + 1][col];
        return __builtin_bit_cast(h8c, u4c{e0.x, e0.y, e1.x, e1.y});
    };
    const int nchunk = (g.D + SKC - 1) / SKC;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) fetch((c + 1) * SKC);
        // Phases kept apart as in cost65_accumulate_f16x2 (cost65_device.hpp): the VALU-heavy split of the next chunk
        // starts one instruction's worth of wait states after the last MFMA - the compiler's own wait states did not
        // cover a VALU write into an operand register of an MFMA still queueing on the matrix pipe (measured there).
        __builtin_amdgcn_sched_barrier(0);
        const h8c awh = frag(buf, 0, 0, 32 * wave + li), awl = frag(buf, 0, 1, 32 * wave + li);
        const h8c a4h = frag(buf, 0, 0, 128 + li), a4l = frag(buf, 0, 1, 128 + li);
#pragma unroll
        for (int tj = 0; tj < 5; ++tj) {
            const h8c bh = frag(buf, 1, 0, 32 * tj + li), bl = frag(buf, 1, 1, 32 * tj + li);
            acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(awl, bh, acc[tj], 0, 0, 0);
            acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(awh, bl, acc[tj], 0, 0, 0);
            acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(awh, bh, acc[tj], 0, 0, 0);
            if (tj == 4) {
                if (wave == 0) {
                    acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4l, bh, acc[6], 0, 0, 0);
                    acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, bl, acc[6], 0, 0, 0);
                    acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, bh, acc[6], 0, 0, 0);
                }
                asm volatile("" :: "v"(bh), "v"(bl));
            }
        }
        {
            const h8c bh = frag(buf, 1, 0, 32 * wave + li), bl = frag(buf, 1, 1, 32 * wave + li);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4l, bh, acc[5], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, bl, acc[5], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4h, bh, acc[5], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            asm volatile("" :: "v"(awh), "v"(awl), "v"(a4h), "v"(a4l), "v"(bh), "v"(bl));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < nchunk) stash(buf ^ 1);        // last read in chunk c - 1; every wave is past that barrier
        __syncthreads();
    }
    // an operand beyond the fp16 range became an infinite hi half: every output of its row / column is then inf or NaN
    // (inf - inf from the lo.hi pass, or inf * 0).  Only outputs that will be stored count.
    bool bad = false;
    auto scan = [&](const f32x16& c, int ti, int tj) {
        const bool colin = g.j0 + 32 * tj + li < g.m;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const bool rowin = g.i0 + 32 * ti + (q & 3) + 8 * (q >> 2) + 4 * kg < g.n;
            bad |= colin && rowin && !(fabsf(c[q]) <= 3.0e38f);
        }
    };
#pragma unroll
    for (int tj = 0; tj < 5; ++tj) scan(acc[tj], wave, tj);
    scan(acc[5], 4, wave);
    if (wave == 0) scan(acc[6], 4, 4);
    return !__syncthreads_or(bad);
}

template <bool SPLIT>
__global__ void __launch_bounds__(256, 2)
cost_mfma_kernel(const float* __restrict__ d0, const float* __restrict__ d1, int D, int n, int m,
                 float rsqrtD, float sqrtD, float* __restrict__ out) {
    __shared__ CostLds lds;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int tiles_j = (m + CT - 1) / CT, tiles = tiles_j * ((n + CT - 1) / CT);
    const int64_t b = blockIdx.x / tiles;
    const int tt = (int)(blockIdx.x - b * tiles);
    const int i0 = (tt / tiles_j) * CT, j0 = (tt % tiles_j) * CT;
    const CostTile g{d0 + b * (int64_t)D * n, d1 + b * (int64_t)D * m, D, n, m, i0, j0};
    float* O = out + b * (int64_t)n * m;

    f32x16 acc[7];
    auto clear = [&]() {
#pragma unroll
        for (int q = 0; q < 7; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    };
    clear();
    float unscale = 1.0f;
    if (SPLIT) {
        if (cost_tile_split(g, lds.s, acc, t, wave)) {
            unscale = 1.0f / (COST_PRESCALE * COST_PRESCALE);     // exact power of two
        } else {
            clear();
            cost_tile_f32(g, lds.f, acc, t, wave);
        }
    } else {
        cost_tile_f32(g, lds.f, acc, t, wave);
    }

    // C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    auto store_tile = [&](const f32x16& cacc, int ti, int tj) {
        const int col = j0 + 32 * tj + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < n && col < m) {
                const float sc = div_invariant(cacc[r] * unscale, sqrtD, rsqrtD);     // `scores / D ** .5`
                O[(int64_t)row * m + col] = 0.1f * sc;                                // `0.1 * scores`
            }
        }
    };
#pragma unroll
    for (int tj = 0; tj < 5; ++tj) store_tile(acc[tj], wave, tj);
    store_tile(acc[5], 4, wave);
    if (wave == 0) store_tile(acc[6], 4, 4);
}

}  // namespace pats

namespace pats { int launch_cost65(const float*, const float*, int, int64_t, float*, hipStream_t); }
using namespace pats;

extern "C" int pats_cost_f32(const float* d0, const float* d1, int64_t batch, int D, int n, int m,
                             float* out, pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && D > 0 && n > 0 && m > 0, "cost: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(d0 && d1 && out, "cost: null pointer");
    if (n == 65 && m == 65 && (D % 32) == 0 && D <= 512)      // third level: one wave per problem, see sinkhorn.hip
        return launch_cost65(d0, d1, D, batch, out, as_stream(stream));
    const int64_t tiles = (int64_t)((n + CT - 1) / CT) * ((m + CT - 1) / CT);
    PATS_REQUIRE(tiles * batch < (1ll << 31), "cost: grid too large (split the call)");
    static const bool fp32_only = [] { const char* e = getenv("PATS_COST_F32"); return e && atoi(e) != 0; }();
    const float sq = (float)sqrt((double)D);
    const dim3 grid((unsigned)(tiles * batch)), block(256);
    if (fp32_only)
        hipLaunchKernelGGL(cost_mfma_kernel<false>, grid, block, 0, as_stream(stream), d0, d1, D, n, m, 1.0f / sq, sq, out);
    else
        hipLaunchKernelGGL(cost_mfma_kernel<true>, grid, block, 0, as_stream(stream), d0, d1, D, n, m, 1.0f / sq, sq, out);
    return check_launch("cost_mfma_kernel");
}
