// attention(query, key, value) of the GNN layers on gfx950.
//   models/modules.py:84-88:   scores = einsum('bdhn,bdhm->bhnm', q, k) / dim**.5
//                              prob   = softmax(scores, dim=-1)
//                              out    = einsum('bhnm,bdhm->bdhn', prob, v)
// q [b,dim,heads,n], k / v [b,dim,heads,m] exactly as MultiHeadedAttention views its projections
// (:101-102): for one (batch, head) the operand q[:, h, :] is a [dim][n] matrix with row stride
// heads*n - channel-major like the descriptors of the cost build, so Q^T K is the same fp32 MFMA
// product (v_mfma_f32_32x32x2_f32, lane l supplies q[d = k0 + (l >> 5)][i0 + (l & 31)], 128-byte lines).
//
// One 256-thread workgroup = (batch, head, 32 query rows):
//   1. S = Q^T K / sqrt(dim) for the 32 rows against all m keys, one 32x32 MFMA tile per wave at a time,
//      into a [32][m_pad] LDS slab (odd stride: conflict-free both by rows and by columns);
//   2. row softmax in place (one wave per row at a time; max, exp, sum, IEEE divide like ATen), padded
//      columns zeroed; the probabilities go to HBM only if the caller asks for them (the reference's own
//      caller discards them, modules.py:103);
//   3. out^T = V P^T: wave w owns channel tile w; V chunks [32 channels][32 keys] are staged through a
//      private LDS tile (coalesced along keys, read back channel-major), P fragments come straight
//      from the slab; the [channel][query] result tile stores 128-byte lines.
// The n x m score matrix never reaches HBM (the stock path writes and re-reads it three times).
// Tokens on this path: 65 / 145 / 300 per side, dim = 32 / 66 / 112, 4 heads; m <= 640 (LDS slab).
#include "common.hpp"

namespace pats {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AR = 32;                    // query rows per workgroup
constexpr int VST = 33;                   // stride of the private V tile

struct AttnArgs {
    const float* q; const float* k; const float* v;
    int dim, heads, n, m, mp;             // mp: slab stride (odd, >= 32 * ceil(m / 32))
    float sq, rsq;                        // dim**.5 and its reciprocal
    float* out; float* prob;
};
}  // namespace

__global__ void __launch_bounds__(256)
attention_kernel(AttnArgs g) {
    extern __shared__ __attribute__((aligned(16))) float slab[];          // [AR][mp], then 4 x [32][VST]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int slabs = (g.n + AR - 1) / AR;
    const int64_t bh = blockIdx.x / slabs;
    const int i0 = (int)(blockIdx.x - bh * slabs) * AR;
    const int64_t bi = bh / g.heads;
    const int h = (int)(bh - bi * g.heads);
    const int64_t rsn = (int64_t)g.heads * g.n, rsm = (int64_t)g.heads * g.m;      // channel strides
    const float* Q = g.q + (bi * g.dim * g.heads + h) * (int64_t)g.n;
    const float* K = g.k + (bi * g.dim * g.heads + h) * (int64_t)g.m;
    const float* V = g.v + (bi * g.dim * g.heads + h) * (int64_t)g.m;
    const int mt = (g.m + 31) / 32;
    float* vt = slab + AR * g.mp + wave * (32 * VST);

    // ---- 1. scores -------------------------------------------------------------------------------
    const bool qv = i0 + li < g.n;
    const float* qp = Q + (qv ? i0 + li : 0);
    for (int tj = wave; tj < mt; tj += 4) {
        const bool kv = 32 * tj + li < g.m;
        const float* kp = K + (kv ? 32 * tj + li : 0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int d0 = 0; d0 < g.dim; d0 += 2) {
            const bool dv = d0 + lk < g.dim;
            const float a = (qv && dv) ? qp[(int64_t)(d0 + lk) * rsn] : 0.f;
            const float b = (kv && dv) ? kp[(int64_t)(d0 + lk) * rsm] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
            slab[row * g.mp + 32 * tj + li] = div_invariant(acc[r], g.sq, g.rsq);      // `/ dim**.5`
        }
    }
    __syncthreads();

    // ---- 2. softmax over the keys, one wave per row ------------------------------------------------
    for (int row = wave; row < AR; row += 4) {
        float* s = slab + row * g.mp;
        if (i0 + row >= g.n) {                              // rows past the end: zero probabilities
            for (int j = lane; j < 32 * mt; j += 64) s[j] = 0.f;
            continue;
        }
        float mx = -INFINITY;
        for (int j = lane; j < g.m; j += 64) mx = fmaxf(mx, s[j]);
        mx = wave_max(mx);
        float den = 0.f;
        for (int j = lane; j < g.m; j += 64) {
            const float e = expf(s[j] - mx);
            s[j] = e;
            den += e;
        }
        den = wave_sum(den);
        float* pr = g.prob ? g.prob + ((bh * g.n + i0 + row) * (int64_t)g.m) : nullptr;
        for (int j = lane; j < 32 * mt; j += 64) {
            const float p = j < g.m ? s[j] / den : 0.f;
            s[j] = p;
            if (pr && j < g.m) pr[j] = p;
        }
    }
    __syncthreads();

    // ---- 3. out^T[d][i] = sum_j V[d][j] P[i][j] -------------------------------------------------------
    const int dt = (g.dim + 31) / 32;
    for (int td = wave; td < dt; td += 4) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int j0 = 0; j0 < 32 * mt; j0 += 32) {
            // stage V[32 td .. +31][j0 .. +31]: a half-wave reads one 128-byte line of a channel row
#pragma unroll 4
            for (int qd = 0; qd < 16; ++qd) {
                const int dd = 2 * qd + lk, d = 32 * td + dd, j = j0 + li;
                vt[dd * VST + li] = (d < g.dim && j < g.m) ? V[(int64_t)d * rsm + j] : 0.f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same wave: LDS is in order, only the compiler needs telling
#pragma unroll 4
            for (int jj = 0; jj < 32; jj += 2) {
                const float a = vt[li * VST + jj + lk];                      // A[d = li][k = jj + lk]
                const float b = slab[li * g.mp + j0 + jj + lk];              // B[k][i = li] = P[i][j]
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // tile consumed before the next chunk overwrites it
        }
        float* O = g.out + (bi * g.dim * g.heads + h) * (int64_t)g.n;
        const int i = i0 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = 32 * td + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (d < g.dim && i < g.n) O[(int64_t)d * rsn + i] = acc[r];
        }
    }
}

}  // namespace pats

using namespace pats;

extern "C" int pats_attention_f32(const float* query, const float* key, const float* value, int64_t batch,
                                  int dim, int heads, int n, int m, float* out, float* prob,
                                  pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && dim > 0 && heads > 0 && n > 0 && m > 0, "attention: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(query && key && value && out, "attention: null pointer");
    const int mt = (m + 31) / 32, mp = 32 * mt + 1;
    const size_t lds = (size_t)(AR * mp + 4 * 32 * VST) * sizeof(float);
    if (lds > 100 * 1024) {
        set_error("attention: m=%d keys exceed the LDS slab (m <= 640)", m);
        return PATS_ERR_UNSUPPORTED;
    }
    const int64_t blocks = batch * heads * ((n + AR - 1) / AR);
    PATS_REQUIRE(blocks < (1ll << 31), "attention: grid too large (split the batch)");
    static bool optin = false;
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute((const void*)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) != hipSuccess)
            return check_launch("attention (LDS opt-in)");
        optin = true;
    }
    const float sq = (float)sqrt((double)dim);
    AttnArgs g{query, key, value, dim, heads, n, m, mp, sq, 1.0f / sq, out, prob};
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)blocks), dim3(256), lds, as_stream(stream), g);
    return check_launch("attention_kernel");
}
