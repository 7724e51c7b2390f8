// Descriptor x descriptor cost build on the fp32 matrix cores of gfx950.
//
// Replaces  scores = einsum('bdn,bdm->bnm', mdesc0, mdesc1) / D**.5 ;  0.1 * scores
//   models/first_layer.py:110-111,114 (D=448, 300x300)   second_layer.py:100-101,104 (D=264,
//   145x145 x B)   third_layer.py:156-158 (D=128, 65x65 x P)
//
// fp32-input MFMA only (v_mfma_f32_32x32x2_f32): it is bitwise a k-ordered fmaf chain, which keeps
// the 1e-4 transport-mass and exact-argmax gates; bf16 would not.  Operands are channel-major
// ([D][n]): for the 32x32x2 shape lane l supplies A[i = l&31][k = l>>5] = d0[k][i0 + (l&31)].
//
// One 256-thread workgroup owns a 160 x 160 output tile = 5 x 5 MFMA tiles: wave w computes tile row
// w (five tiles) plus tile (4, w), wave 0 also (4, 4) - so a fine-level problem (145 x 145) is exactly
// one workgroup with no idle wave.  The operand slabs of a k-chunk (8 descriptor rows x 160 columns
// of each side, 10 KB) are staged global -> registers -> LDS once per workgroup, double-buffered,
// one barrier per chunk; fragments are then read from LDS (conflict-free: a half-wave reads 32
// consecutive floats of one row).  The first version had every wave fetch both fragments from
// global memory for every MFMA (two 256-byte requests per 64-cycle MFMA and wave - vector-L1 bound,
// 40 TF/s on the 145-wide batch); here each descriptor element reaches the CU once per workgroup.
// Scale and the reference's two-step `/ sqrt(D)`, `* 0.1` rounding are applied in the epilogue.
#include "common.hpp"

namespace pats {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CT = 160, KC = 8, CPT = KC * CT / 256;      // workgroup tile edge, k-chunk, floats per thread per slab

struct __attribute__((aligned(16))) CostLds {
    float a[2][KC][CT];
    float b[2][KC][CT];
};

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
    const float* A = d0 + b * (int64_t)D * n;
    const float* B = d1 + b * (int64_t)D * m;
    float* O = out + b * (int64_t)n * m;

    // staging map: element e = t + 256 q of a slab is (row e / 160, column e % 160).  Columns past the
    // matrix edge are CLAMPED, not predicated: they only feed output rows / columns that are never
    // stored, and the main loop stays free of branches (guarded loads and per-tile predicates put every
    // MFMA in its own basic block - measured slower than the kernel this one replaces).  Pointers are
    // fixed per thread and advance by KC descriptor rows per chunk; only a ragged last chunk (D % 8)
    // needs the zero-filling fetch.
    float ra[CPT], rb[CPT];
    const float* pa[CPT];
    const float* pb[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int e = t + 256 * q, kk = e / CT, col = e % CT;
        pa[q] = A + (int64_t)kk * n + min(i0 + col, n - 1);
        pb[q] = B + (int64_t)kk * m + min(j0 + col, m - 1);
    }
    const int64_t stepa = (int64_t)KC * n, stepb = (int64_t)KC * m;
    auto fetch = [&](int k0) {
        if (k0 + KC <= D) {
#pragma unroll
            for (int q = 0; q < CPT; ++q) { ra[q] = *pa[q]; rb[q] = *pb[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const bool kin = k0 + (t + 256 * q) / CT < D;
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

    f32x16 acc[7];
#pragma unroll
    for (int q = 0; q < 7; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int nchunk = (D + KC - 1) / KC;
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

    // C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    auto store_tile = [&](const f32x16& cacc, int ti, int tj) {
        const int col = j0 + 32 * tj + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < n && col < m) {
                const float sc = div_invariant(cacc[r], sqrtD, rsqrtD);     // `scores / D ** .5`
                O[(int64_t)row * m + col] = 0.1f * sc;                      // `0.1 * scores`
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
    hipLaunchKernelGGL(cost_mfma_kernel, dim3((unsigned)(tiles * batch)), dim3(256), 0,
                       as_stream(stream), d0, d1, D, n, m, 1.0f / (float)sqrt((double)D), (float)sqrt((double)D), out);
    return check_launch("cost_mfma_kernel");
}
