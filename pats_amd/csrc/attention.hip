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
//   2. row softmax in place (one wave per row at a time; max, exp2 of the pre-scaled difference, sum, one
//      reciprocal per row), padded
//      columns zeroed; the probabilities go to HBM only if the caller asks for them (the reference's own
//      caller discards them, modules.py:103);
//   3. out^T = V P^T: wave w owns channel tile w; V chunks [32 channels][32 keys] are staged through a
//      private LDS tile (coalesced along keys, read back channel-major), P fragments come straight
//      from the slab; the [channel][query] result tile stores 128-byte lines.
// The n x m score matrix never reaches HBM (the stock path writes and re-reads it three times).
// Tokens on this path: 65 / 145 / 300 (768 at YFCC size) per side, dim = 32 / 66 / 112, 4 heads;
// the [rows][m] slab must fit the CU's 160 KB of LDS: 32 query rows per workgroup up to m = 1024, 16 / 8 / 4 beyond (m <= 8416).
#include "cost65_device.hpp"

namespace pats {

namespace {
constexpr int AR = 32;                    // query rows per workgroup
constexpr int VST = 33;                   // stride of the private V tile

struct AttnArgs {
    const float* q; const float* k; const float* v;
    int dim, heads, n, m, mp;             // mp: slab stride (odd, >= 32 * ceil(m / 32))
    int rows;                             // query rows per workgroup: 32, or 16 / 8 / 4 when m is too large for a 32-row slab
    float sq, rsq;                        // dim**.5 and its reciprocal
    float* out; float* prob;
    const int* gate;                      // optional: the launch is a no-op unless *gate != 0 (fallback behind the fused GNN layer)
    int64_t total;                        // attention65_kernel: batch * heads work items (a gated launch covers them with a capped grid)
};
}  // namespace

__global__ void __launch_bounds__(256)
attention_kernel(AttnArgs g) {
    extern __shared__ __attribute__((aligned(16))) float slab[];          // [rows][mp], then 4 x [32][VST]
    if (g.gate && *g.gate == 0) return;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lk = lane >> 5;
    // beyond 1024 keys a 32-row slab no longer fits the CU's LDS: the workgroup then takes 16 / 8 / 4 query rows (the MFMA
    // tiles stay 32 wide, their unused rows carry zeros and are neither stored nor read back)
    const int R = g.rows;
    const int slabs = (g.n + R - 1) / R;
    const int64_t bh = blockIdx.x / slabs;
    const int i0 = (int)(blockIdx.x - bh * slabs) * R;
    const int64_t bi = bh / g.heads;
    const int h = (int)(bh - bi * g.heads);
    const int64_t rsn = (int64_t)g.heads * g.n, rsm = (int64_t)g.heads * g.m;      // channel strides
    const float* Q = g.q + (bi * g.dim * g.heads + h) * (int64_t)g.n;
    const float* K = g.k + (bi * g.dim * g.heads + h) * (int64_t)g.m;
    const float* V = g.v + (bi * g.dim * g.heads + h) * (int64_t)g.m;
    const int mt = (g.m + 31) / 32;
    float* vt = slab + R * g.mp + wave * (32 * VST);

    // ---- 1. scores -------------------------------------------------------------------------------
    const bool qv = li < R && i0 + li < g.n;
    const float* qp = Q + (qv ? i0 + li : 0);
    for (int tj = wave; tj < mt; tj += 4) {
        const bool kv = 32 * tj + li < g.m;
        const float* kp = K + (kv ? 32 * tj + li : 0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // blocks of 8 k-steps: all 16 operand loads of a block are issued before its MFMAs (one load
        // latency per block instead of one per MFMA)
        for (int d0 = 0; d0 < g.dim; d0 += 16) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + 2 * u + lk;
                const bool dv = d < g.dim;
                a[u] = (qv && dv) ? qp[(int64_t)d * rsn] : 0.f;
                b[u] = (kv && dv) ? kp[(int64_t)d * rsm] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < R) slab[row * g.mp + 32 * tj + li] = div_invariant(acc[r], g.sq, g.rsq);      // `/ dim**.5`
        }
    }
    wg_barrier();

    // ---- 2. softmax over the keys, one wave per row ------------------------------------------------
    for (int row = wave; row < R; row += 4) {
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
            const float e = fast_exp2((s[j] - mx) * LOG2E);     // v_exp_f32: ~1 ulp, far inside the 2e-6 gate on probabilities
            s[j] = e;
            den += e;
        }
        const float inv = 1.0f / wave_sum(den);
        float* pr = g.prob ? g.prob + ((bh * g.n + i0 + row) * (int64_t)g.m) : nullptr;
        for (int j = lane; j < 32 * mt; j += 64) {
            const float p = j < g.m ? s[j] * inv : 0.f;
            s[j] = p;
            if (pr && j < g.m) pr[j] = p;
        }
    }
    wg_barrier();

    // ---- 3. out^T[d][i] = sum_j V[d][j] P[i][j] -------------------------------------------------------
    const int dt = (g.dim + 31) / 32;
    for (int td = wave; td < dt; td += 4) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // V[32 td .. +31][j0 .. +31] chunks: a half-wave reads one 128-byte line of a channel row; the next
        // chunk is fetched into registers while the MFMAs of the current one run
        float vr[16];
        auto fetch = [&](int j0) {
#pragma unroll
            for (int qd = 0; qd < 16; ++qd) {
                const int d = 32 * td + 2 * qd + lk, j = j0 + li;
                vr[qd] = (d < g.dim && j < g.m) ? V[(int64_t)d * rsm + j] : 0.f;
            }
        };
        fetch(0);
        for (int j0 = 0; j0 < 32 * mt; j0 += 32) {
#pragma unroll
            for (int qd = 0; qd < 16; ++qd) vt[(2 * qd + lk) * VST + li] = vr[qd];
            if (j0 + 32 < 32 * mt) fetch(j0 + 32);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same wave: LDS is in order, only the compiler needs telling
#pragma unroll 4
            for (int jj = 0; jj < 32; jj += 2) {
                const float a = vt[li * VST + jj + lk];                      // A[d = li][k = jj + lk]
                const float b = li < R ? slab[li * g.mp + j0 + jj + lk] : 0.f;   // B[k][i = li] = P[i][j]
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // tile consumed before the next chunk overwrites it
        }
        float* O = g.out + (bi * g.dim * g.heads + h) * (int64_t)g.n;
        const int i = i0 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = 32 * td + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (d < g.dim && li < R && i < g.n) O[(int64_t)d * rsn + i] = acc[r];
        }
    }
}

// ---- the third-level shape: 65 tokens per side, dim 32 - one wave per (batch, head) ----------------------
// S^T = K^T Q is built with the in-wave cost machinery (cost65_device.hpp: 2x2 MFMA tiles over even / odd
// tokens, token 64 of either side as VALU chains), i.e. with KEYS as MFMA rows and QUERIES as MFMA columns:
// a lane then owns queries 2 li and 2 li + 1 and half of their keys sit in its accumulator registers (the
// other half in lane ^ 32).  The softmax is therefore in-lane plus one half-wave exchange, and the
// probabilities never move: accumulator register r of tile (key parity, query parity) IS the B operand of
// the second product out^T = V P^T (lane l supplies B[k = l >> 5][column = l & 31] = P[query][key(r, l >> 5)]),
// with the matching A operand V[channel = l & 31][key(r, l >> 5)] read from an 8.3 KB LDS copy of V.
// Token 64 on the key side is one more rank-1 MFMA step, on the query side a 65-term dot product per channel.
struct Attn65Lds {
    float edge[1024];            // cost65_accumulate's staging of token 64's channels
    float vt[32 * 65];           // V[channel][key]
    float p64[72];               // probabilities of query 64
};

__global__ void __launch_bounds__(64, 2)
attention65_kernel(AttnArgs g) {
    __shared__ Attn65Lds lds;
    if (g.gate && *g.gate == 0) return;
    const int lane = threadIdx.x, li = lane & 31, lk = lane >> 5;
    // (grid-stride: an ungated launch has one workgroup per item; a GATED one - the fp32 redo behind the fused GNN layer, which
    //  normally leaves at the gate - comes with a capped grid: an empty launch of 131 072 one-wave workgroups cost 28 us, 180 of them
    //  a step in the bench's with-GNN leg)
    for (int64_t bh = blockIdx.x; bh < g.total; bh += gridDim.x) {
    const int64_t bi = bh / g.heads;
    const int h = (int)(bh - bi * g.heads);
    const int ld = g.heads * 65;
    const int64_t base = (bi * 32 * g.heads + h) * 65;
    const float* Q = g.q + base;
    const float* K = g.k + base;
    const float* V = g.v + base;

    // V -> LDS early (independent of the scores): element e = lane + 64 s of the [32][65] matrix
    float vreg[33];
#pragma unroll
    for (int s = 0; s < 33; ++s) {
        const int e = lane + 64 * s, d = e / 65, j = e - d * 65;
        vreg[s] = e < 32 * 65 ? V[d * ld + j] : 0.f;
    }
    Cost65Acc c;
    cost65_accumulate(K, Q, 32, lds.edge, lane, c, ld);          // rows = keys, columns = queries
#pragma unroll
    for (int s = 0; s < 33; ++s) {
        const int e = lane + 64 * s;
        if (e < 32 * 65) lds.vt[e] = vreg[s];
    }
    // ---- scale, softmax per query --------------------------------------------------------------------
    f32x16* tiles[4] = {&c.c00, &c.c01, &c.c10, &c.c11};          // [key parity * 2 + query parity]
    float pd[2];                                                   // probability of key 64 for queries 2 li + {0, 1}
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
        f32x16& e0 = *tiles[tq];                                   // even keys
        f32x16& e1 = *tiles[2 + tq];                               // odd keys
        float sd = div_invariant(tq ? c.er1 : c.er0, g.sq, g.rsq); // score against key 64
        float mx = sd;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            e0[r] = div_invariant(e0[r], g.sq, g.rsq);
            e1[r] = div_invariant(e1[r], g.sq, g.rsq);
            mx = fmaxf(mx, fmaxf(e0[r], e1[r]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float den = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            e0[r] = fast_exp2((e0[r] - mx) * LOG2E);
            e1[r] = fast_exp2((e1[r] - mx) * LOG2E);
            den += e0[r] + e1[r];
        }
        den += __shfl_xor(den, 32);
        sd = fast_exp2((sd - mx) * LOG2E);
        const float inv = 1.0f / (den + sd);
#pragma unroll
        for (int r = 0; r < 16; ++r) { e0[r] *= inv; e1[r] *= inv; }
        pd[tq] = sd * inv;
    }
    // query 64: its scores against keys 2 li, 2 li + 1 (ec0, ec1: complete in both halves) and key 64 (cn)
    {
        const float s0 = div_invariant(c.ec0, g.sq, g.rsq), s1 = div_invariant(c.ec1, g.sq, g.rsq);
        const float sc = div_invariant(c.cn, g.sq, g.rsq);
        const float mx = fmaxf(wave_max(fmaxf(s0, s1)), sc);
        const float x0 = fast_exp2((s0 - mx) * LOG2E), x1 = fast_exp2((s1 - mx) * LOG2E), xc = fast_exp2((sc - mx) * LOG2E);
        const float den = wave_sum(lk == 0 ? x0 + x1 : 0.f) + xc;
        const float inv = 1.0f / den;
        if (lk == 0) { lds.p64[2 * li] = x0 * inv; lds.p64[2 * li + 1] = x1 * inv; }
        if (lane == 0) lds.p64[64] = xc * inv;
    }
    wg_barrier();                                               // vt and p64 visible (single wave: ordering only)
    // ---- out^T = V P^T -------------------------------------------------------------------------------------
    f32x16 o0, o1;                                                 // queries 2 li (o0) and 2 li + 1 (o1); rows = channels
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    const float* va = lds.vt + li * 65 + 8 * lk;                   // V[channel li][key 2 (rc0 + 4 lk) + parity]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j0 = 2 * ((r & 3) + 8 * (r >> 2));              // even key of register r for lk = 0
        const float a_even = va[j0], a_odd = va[j0 + 1];
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_even, c.c00[r], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_even, c.c01[r], o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_odd, c.c10[r], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_odd, c.c11[r], o1, 0, 0, 0);
    }
    {   // key 64: rank-1 step (k = 0 carries it, k = 1 is zero)
        const float a64 = lk == 0 ? lds.vt[li * 65 + 64] : 0.f;
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a64, lk == 0 ? pd[0] : 0.f, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a64, lk == 0 ? pd[1] : 0.f, o1, 0, 0, 0);
    }
    float* O = g.out + base;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * lk;
        *reinterpret_cast<f2u*>(O + d * ld + 2 * li) = f2u{o0[r], o1[r]};
    }
    // query 64: out[d][64] = sum_j V[d][j] p64[j], channel d = li in the lower half-wave
    if (lk == 0) {
        float acc0 = 0.f, acc1 = 0.f;
        const float* vr = lds.vt + li * 65;
#pragma unroll 8
        for (int j = 0; j < 64; j += 2) {
            acc0 = fmaf(vr[j], lds.p64[j], acc0);
            acc1 = fmaf(vr[j + 1], lds.p64[j + 1], acc1);
        }
        O[li * ld + 64] = fmaf(vr[64], lds.p64[64], acc0 + acc1);
    }
    wg_barrier();                                               // the LDS staging is free for the next item
    }
}

}  // namespace pats

using namespace pats;

namespace pats {
int launch_attention(const float* query, const float* key, const float* value, int64_t batch, int dim, int heads, int n, int m,
                     float* out, float* prob, pats_stream_t stream, const int* gate);
}

extern "C" int pats_attention_f32(const float* query, const float* key, const float* value, int64_t batch,
                                  int dim, int heads, int n, int m, float* out, float* prob,
                                  pats_stream_t stream) {
    return launch_attention(query, key, value, batch, dim, heads, n, m, out, prob, stream, nullptr);
}

int pats::launch_attention(const float* query, const float* key, const float* value, int64_t batch, int dim, int heads, int n,
                           int m, float* out, float* prob, pats_stream_t stream, const int* gate) {
    PATS_REQUIRE(batch >= 0 && dim > 0 && heads > 0 && n > 0 && m > 0, "attention: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(query && key && value && out, "attention: null pointer");
    const float sq0 = (float)sqrt((double)dim);
    static const bool general_only = diag_env("PATS_ATTN_GENERAL") != nullptr;     // A/B switch for benchmarking
    if (n == 65 && m == 65 && dim == 32 && !prob && !general_only) {            // the third-level shape
        PATS_REQUIRE(batch * heads < (1ll << 31), "attention: grid too large (split the batch)");
        AttnArgs g{query, key, value, dim, heads, n, m, 0, 32, sq0, 1.0f / sq0, out, nullptr, gate, batch * heads};
        const int64_t wgs = gate ? std::min<int64_t>(batch * heads, 4096) : batch * heads;
        hipLaunchKernelGGL(attention65_kernel, dim3((unsigned)wgs), dim3(64), 0, as_stream(stream), g);
        return check_launch("attention65_kernel");
    }
    const int mt = (m + 31) / 32, mp = 32 * mt + 1;
    // query rows per workgroup: 32 while the [rows][m] score slab fits the CU's LDS (m <= 1024), then 16 / 8 / 4
    // (m <= 2 080 / 4 192 / 8 416: the reference's demo runs 1 900 tokens, demo.py:36)
    int rows = AR;
    size_t lds = (size_t)(rows * mp + 4 * 32 * VST) * sizeof(float);
    while (lds > 152 * 1024 && rows > 4) {
        rows >>= 1;
        lds = (size_t)(rows * mp + 4 * 32 * VST) * sizeof(float);
    }
    if (lds > 152 * 1024) {
        set_error("attention: m=%d keys exceed the LDS slab even at 4 query rows per workgroup (m <= 8416)", m);
        return PATS_ERR_UNSUPPORTED;
    }
    const int64_t blocks = batch * heads * ((n + rows - 1) / rows);
    PATS_REQUIRE(blocks < (1ll << 31), "attention: grid too large (split the batch)");
    if (lds > 64 * 1024 &&      // per-device attribute: set whenever needed (cheap), never cached process-wide
        hipFuncSetAttribute((const void*)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
        return check_launch("attention (LDS opt-in)");
    const float sq = (float)sqrt((double)dim);
    AttnArgs g{query, key, value, dim, heads, n, m, mp, rows, sq, 1.0f / sq, out, prob, gate};
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)blocks), dim3(256), lds, as_stream(stream), g);
    return check_launch("attention_kernel");
}
