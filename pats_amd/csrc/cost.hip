// Descriptor x descriptor cost build on the matrix cores of gfx950.
//
// Replaces  scores = einsum('bdn,bdm->bnm', mdesc0, mdesc1) / D**.5 ;  0.1 * scores
//   models/first_layer.py:110-111,114 (D=448, 300x300)   second_layer.py:100-101,104 (D=264,
//   145x145 x B)   third_layer.py:156-158 (D=128, 65x65 x P)
//
// The contraction is the 160 x 160 tile of mfma_tile.hpp - one 256-thread workgroup per tile, so a fine-level problem
// (145 x 145) is exactly one workgroup - in its fp16-split form: fp32 operands as fp16 hi + lo pairs, three
// exact-product MFMA passes, fp32 accumulation, closer to float64 than the fp32 fma chain; a workgroup whose operands
// leave the fp16 range redoes its tile with the fp32 MFMA in the same launch.  PATS_COST_F32=1 selects the fp32 MFMA
// for the whole kernel (ablation / A-B runs, tools/cost_ab.py).  Operands are channel-major ([D][n]); columns past
// the matrix edge are CLAMPED, not predicated (they only feed output rows / columns that are never stored, and the
// main loop stays free of branches).  Scale and the reference's two-step `/ sqrt(D)`, `* 0.1` rounding are applied
// in the epilogue.
#include "mfma_tile.hpp"

#include <cstdlib>

namespace pats {

namespace {

// operand source of mfma_tile.hpp: A = d0 [D][n] at columns i0.., B = d1 [D][m] at columns j0..
struct CostSrc {
    const float *A, *B;
    int D, n, m, i0, j0, t;
    const float* p[mt::SQ];            // running pointers (split items, or fp32 slab elements: pa in p[0..4], pb in pq)
    const float* pq[mt::CPT];

    __device__ __forceinline__ CostSrc(const float* A_, const float* B_, int D_, int n_, int m_, int i0_, int j0_, int t_)
        : A(A_), B(B_), D(D_), n(n_), m(m_), i0(i0_), j0(j0_), t(t_) {}
    __device__ __forceinline__ bool row_stored(int r) const { return i0 + r < n; }
    __device__ __forceinline__ bool col_stored(int c) const { return j0 + c < m; }
    __device__ __forceinline__ void rewind() {}

    // only a ragged last chunk (D % 8) needs the zero-filling fetch
    __device__ __forceinline__ void fetch_f32(int k0, float (&ra)[mt::CPT], float (&rb)[mt::CPT]) {
        if (k0 == 0) {
#pragma unroll
            for (int q = 0; q < mt::CPT; ++q) {
                const int kk = mt::f32_row(t, q), col = mt::f32_col(t, q);
                p[q] = A + (int64_t)kk * n + min(i0 + col, n - 1);
                pq[q] = B + (int64_t)kk * m + min(j0 + col, m - 1);
            }
        }
        if (k0 + mt::KC <= D) {
#pragma unroll
            for (int q = 0; q < mt::CPT; ++q) { ra[q] = *p[q]; rb[q] = *pq[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < mt::CPT; ++q) {
                const bool kin = k0 + mt::f32_row(t, q) < D;
                ra[q] = kin ? *p[q] : 0.f;
                rb[q] = kin ? *pq[q] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < mt::CPT; ++q) { p[q] += (int64_t)mt::KC * n; pq[q] += (int64_t)mt::KC * m; }
    }

    __device__ __forceinline__ void fetch_split(int k0, float (&r)[mt::SQ][4]) {
        if (k0 == 0) {
#pragma unroll
            for (int q = 0; q < mt::SQ; ++q) {
                const int col = mt::item_col(t, q), side = mt::item_side(t, q);
                p[q] = (side ? B + min(j0 + col, m - 1) : A + min(i0 + col, n - 1)) + (int64_t)(4 * mt::item_quad(t, q)) * (side ? m : n);
            }
        }
        int ld[mt::SQ];
#pragma unroll
        for (int q = 0; q < mt::SQ; ++q) ld[q] = mt::item_side(t, q) ? m : n;
        if (k0 + mt::SKC <= D) {
#pragma unroll
            for (int q = 0; q < mt::SQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) r[q][e] = p[q][e * ld[q]];
        } else {
#pragma unroll
            for (int q = 0; q < mt::SQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {      // ragged last chunk: read a valid row, then zero (no predicated loads)
                    const int row = k0 + 4 * mt::item_quad(t, q) + e, back = min(row, D - 1) - (row - e);
                    const float v = p[q][back * ld[q]];
                    r[q][e] = row < D ? v : 0.f;
                }
        }
#pragma unroll
        for (int q = 0; q < mt::SQ; ++q) p[q] += mt::SKC * ld[q];
    }
};

}  // namespace

template <bool SPLIT>
__global__ void __launch_bounds__(256, 2)
cost_mfma_kernel(const float* __restrict__ d0, const float* __restrict__ d1, int D, int n, int m,
                 float rsqrtD, float sqrtD, float* __restrict__ out) {
    __shared__ mt::Lds lds;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int tiles_j = (m + mt::CT - 1) / mt::CT, tiles = tiles_j * ((n + mt::CT - 1) / mt::CT);
    const int64_t b = blockIdx.x / tiles;
    const int tt = (int)(blockIdx.x - b * tiles);
    const int i0 = (tt / tiles_j) * mt::CT, j0 = (tt % tiles_j) * mt::CT;
    CostSrc src(d0 + b * (int64_t)D * n, d1 + b * (int64_t)D * m, D, n, m, i0, j0, t);
    float* O = out + b * (int64_t)n * m;

    mt::f32x16 acc[7];
    const float unscale = mt::tile<SPLIT, true>(src, lds, acc, D, true, t, wave);

    // C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    auto store_tile = [&](const mt::f32x16& cacc, int ti, int tj) {
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
    const int64_t tiles = (int64_t)((n + mt::CT - 1) / mt::CT) * ((m + mt::CT - 1) / mt::CT);
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
