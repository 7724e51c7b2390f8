// Descriptor x descriptor cost build on the fp32 matrix cores of gfx950.
//
// Replaces  scores = einsum('bdn,bdm->bnm', mdesc0, mdesc1) / D**.5 ;  0.1 * scores
//   models/first_layer.py:110-111,114 (D=448, 300x300)   second_layer.py:100-101,104 (D=264,
//   145x145 x B)   third_layer.py:156-158 (D=128, 65x65 x P)
//
// fp32-input MFMA only (v_mfma_f32_32x32x2_f32): it is bitwise a k-ordered fmaf chain, which keeps
// the 1e-4 transport-mass and exact-argmax gates; bf16 would not.  Operands are channel-major
// ([D][n]): for the 32x32x2 shape lane l supplies A[i = l&31][k = l>>5] = d0[k][i0 + (l&31)], i.e.
// each half-wave reads one 128-byte line of a descriptor row - fragment-shaped loads are already
// fully coalesced here, so the operands go global -> VGPR with no LDS stage.
// One wave = one 32x32 output tile; a 256-thread workgroup = a 64x64 block (operand lines are
// shared through L1 between the block's four waves).  Scale and the reference's two-step
// `/ sqrt(D)`, `* 0.1` rounding are applied in the epilogue.
#include "common.hpp"

namespace pats {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256)
cost_mfma_kernel(const float* __restrict__ d0, const float* __restrict__ d1, int D, int n, int m,
                 float rsqrtD, float sqrtD, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_j = (m + 63) / 64, tiles = tiles_j * ((n + 63) / 64);
    const int64_t b = blockIdx.x / tiles;
    const int t = (int)(blockIdx.x - b * tiles);
    const int bi = t / tiles_j, bj = t - bi * tiles_j;
    const int i0 = bi * 64 + (wave >> 1) * 32, j0 = bj * 64 + (wave & 1) * 32;
    if (i0 >= n || j0 >= m) return;
    const float* A = d0 + b * (int64_t)D * n;
    const float* B = d1 + b * (int64_t)D * m;
    float* O = out + b * (int64_t)n * m;
    // One-wide edge tiles (the dustbin row / column of 65 = 64 + 1 and 145 = 144 + 1 shapes): a
    // 32x32 MFMA tile would spend the whole K loop on one valid row, so these are plain fp32 dot
    // products instead - one output per lane, fmaf chain in k order (same numerics as the MFMA).
    if (n - i0 == 1 || m - j0 == 1) {
        const bool row_edge = (n - i0 == 1);
        const int cnt = row_edge ? min(32, m - j0) : min(32, n - i0);
        for (int e = lane; e < cnt; e += 64) {
            const int i = row_edge ? i0 : i0 + e, j = row_edge ? j0 + e : j0;
            float acc = 0.f;
            for (int k = 0; k < D; ++k) acc = fmaf(A[(int64_t)k * n + i], B[(int64_t)k * m + j], acc);
            O[(int64_t)i * m + j] = 0.1f * div_invariant(acc, sqrtD, rsqrtD);
        }
        return;
    }
    const int li = lane & 31, lk = lane >> 5;
    const int ia = i0 + li, jb = j0 + li;
    const bool va = ia < n, vb = jb < m;
    const float* pa = A + (int64_t)lk * n + (va ? ia : 0);
    const float* pb = B + (int64_t)lk * m + (vb ? jb : 0);
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= D; k += 8) {
        float a[4], bb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = va ? pa[(int64_t)(k + 2 * q) * n] : 0.f;
            bb[q] = vb ? pb[(int64_t)(k + 2 * q) * m] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bb[q], acc, 0, 0, 0);
    }
    for (; k < D; k += 2) {
        const bool vk = (k + lk) < D;
        const float a = (va && vk) ? pa[(int64_t)k * n] : 0.f;
        const float bv = (vb && vk) ? pb[(int64_t)k * m] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
    // C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int col = j0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < n && col < m) {
            float s = div_invariant(acc[r], sqrtD, rsqrtD);       // `scores / D ** .5`
            O[(int64_t)row * m + col] = 0.1f * s;   // `0.1 * scores`
        }
    }
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
    const int64_t tiles = (int64_t)((n + 63) / 64) * ((m + 63) / 64);
    PATS_REQUIRE(tiles * batch < (1ll << 31), "cost: grid too large (split the call)");
    hipLaunchKernelGGL(cost_mfma_kernel, dim3((unsigned)(tiles * batch)), dim3(256), 0,
                       as_stream(stream), d0, d1, D, n, m, 1.0f / (float)sqrt((double)D), (float)sqrt((double)D), out);
    return check_launch("cost_mfma_kernel");
}
