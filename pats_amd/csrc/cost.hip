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

// column addressing for mfma_tile.hpp's CmSrc: A = d0 [D][n] at columns i0.., B = d1 [D][m] at columns j0.. (clamped)
// STREAM: the main loop's operand loads non-temporal (an experiment switch, PATS_COST_NT: see launch_cost)
template <bool STREAM>
struct CostColsT {
    static constexpr bool stream = STREAM;
    int n, m, i0, j0;
    __device__ __forceinline__ int64_t a_off(int c) const { return min(i0 + c, n - 1); }
    __device__ __forceinline__ int64_t b_off(int c) const { return min(j0 + c, m - 1); }
    __device__ __forceinline__ bool row_stored(int r) const { return i0 + r < n; }
    __device__ __forceinline__ bool col_stored(int c) const { return j0 + c < m; }
};

}  // namespace

template <bool SPLIT, bool STREAM = false>
__global__ void __launch_bounds__(256, 2)
cost_mfma_kernel(const float* __restrict__ d0, const float* __restrict__ d1, int D, int n, int m,
                 float rsqrtD, float sqrtD, float* __restrict__ out, const int64_t* __restrict__ live) {
    __shared__ mt::Lds lds;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int tiles_j = (m + mt::CT - 1) / mt::CT, tiles = tiles_j * ((n + mt::CT - 1) / mt::CT);
    const int64_t b = blockIdx.x / tiles;
    if (live && b >= *live) return;            // counted launch (throughput mode): problems past the device-side count
    const int tt = (int)(blockIdx.x - b * tiles);
    const int i0 = (tt / tiles_j) * mt::CT, j0 = (tt % tiles_j) * mt::CT;
    typedef CostColsT<STREAM> CostCols;
    const CostCols cols{n, m, i0, j0};
    mt::CmSrc<CostCols> src(d0 + b * (int64_t)D * n, n, d1 + b * (int64_t)D * m, m, D, cols, t);
    float* O = out + b * (int64_t)n * m;

    mt::f32x16 acc[7];
    const float unscale = mt::tile<SPLIT, true>(src, (mt::CmSrc<CostCols>*)nullptr, lds, acc, true, t, wave);

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

namespace pats {
int launch_cost(const float* d0, const float* d1, int64_t batch, int D, int n, int m, float* out, pats_stream_t stream,
                const int64_t* live);
}

extern "C" int pats_cost_f32(const float* d0, const float* d1, int64_t batch, int D, int n, int m,
                             float* out, pats_stream_t stream) {
    return launch_cost(d0, d1, batch, D, n, m, out, stream, nullptr);
}

// live: optional device-side problem count (<= batch) - the MFMA-tile kernel's workgroups past it return at once
int pats::launch_cost(const float* d0, const float* d1, int64_t batch, int D, int n, int m, float* out, pats_stream_t stream,
                      const int64_t* live) {
    PATS_REQUIRE(batch >= 0 && D > 0 && n > 0 && m > 0, "cost: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(d0 && d1 && out, "cost: null pointer");
    if (n == 65 && m == 65 && (D % 32) == 0 && D <= 512)      // third level: one wave per problem, see sinkhorn.hip
        return launch_cost65(d0, d1, D, batch, out, as_stream(stream));
    const int64_t tiles = (int64_t)((n + mt::CT - 1) / mt::CT) * ((m + mt::CT - 1) / mt::CT);
    PATS_REQUIRE(tiles * batch < (1ll << 31), "cost: grid too large (split the call)");
    const bool fp32_only = cost_f32_only();
    const float sq = (float)sqrt((double)D);
    const dim3 grid((unsigned)(tiles * batch)), block(256);
    static const bool nt = [] { const char* e = diag_env("PATS_COST_NT"); return e && atoi(e) != 0; }();
    if (fp32_only)
        hipLaunchKernelGGL(cost_mfma_kernel<false>, grid, block, 0, as_stream(stream), d0, d1, D, n, m, 1.0f / sq, sq, out, live);
    else if (nt)
        hipLaunchKernelGGL((cost_mfma_kernel<true, true>), grid, block, 0, as_stream(stream), d0, d1, D, n, m, 1.0f / sq, sq, out, live);
    else
        hipLaunchKernelGGL(cost_mfma_kernel<true>, grid, block, 0, as_stream(stream), d0, d1, D, n, m, 1.0f / sq, sq, out, live);
    return check_launch("cost_mfma_kernel");
}
