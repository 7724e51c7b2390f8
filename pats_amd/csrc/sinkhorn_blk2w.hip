// Fine-level Sinkhorn (145 x 145) on TWO waves per problem (round 4, last part): the same solve as sinkhorn_blk145_kernel
// (sinkhorn_blk.hip: models/modules.py:137-143 sweeps in the linear domain, :165-182 marginals, second_layer.py:107-112 bias) with
// a lane's block grown from 9 x 9 to 9 x 18.
//
// Why: the 16 x 16 lane grid of sinkhorn_blk145_kernel spends 133 of its 208 VALU instructions per wave and sweep on reductions
// and exchanges (profiles/r04_pmc_sq.md: 96 % VALU busy, 3 barriers per sweep), and that overhead is per WAVE - four of them per
// problem.  Here a wave is an 8 x 8 lane grid (lane = 8 Iw + J) over 72 rows x 144 columns: lane (Iw, J) of wave w holds
// K[9 I .. 9 I + 8][18 J .. 18 J + 17], I = 8 w + Iw, in 162 registers; a sweep is
//   rows   : 81 packed FMAs against b[18 J ..] (the wave's private LDS copy), nine partial sums reduced over the 8 lanes of J by the
//            transposed butterfly of lane_reduce.hpp (row 9 I + J lands in lane J; the ninth row is all-reduced, owner J = 0);
//            rows never leave the wave;
//   columns: 81 packed FMAs against a[9 I ..], eighteen partials reduced over the 8 lanes of Iw (permlane32 / 16 swaps, row_ror:8:
//            lane (Iw, J) ends with columns 18 J + Iw, 18 J + 8 + Iw and - Iw = 0 / 4 - 18 J + 16 / 17), the two waves' halves meet
//            in LDS behind the ONE barrier of the sweep, and BOTH waves form every b_j (same order, same bits) into their own copy.
//   dustbin row: its sum runs over all columns, every column is owned once per wave - each wave forms it alone; dustbin column:
//   partials through LDS with the column halves.
// 572 VALU instructions per problem and sweep instead of 832, one barrier instead of three, 72-byte instead of 36-byte runs of
// the score matrix; ~235 registers: two waves per SIMD.  Same guard, same flags, same epilogue arithmetic as the four-wave kernel;
// summation ORDER differs (results agree to rounding; the match flags of a plan are those of its own values).
// PATS_FINE_W2=0 selects the four-wave kernel (read once per process).
#include "lane_reduce.hpp"
#include <stdlib.h>

namespace pats {

namespace {

constexpr int N_ = 145, NB = 144, BR = 9, BC = 18;
constexpr int VA_S = 12, VB_S = 20;                     // padded strides of a lane group's 9 / 18 floats in LDS (16-byte reads)
constexpr float W2_GUARD = 1073741824.0f;               // 2^30, as sinkhorn.hip
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ bool ok_scale(float x) { return x <= W2_GUARD && x > 0.f; }
__device__ __forceinline__ float uni(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
__device__ __forceinline__ float mul_rcp(float num, float den) { return num * __builtin_amdgcn_rcpf(den); }

// (wave_lds_sync(): common.hpp - every intra-wave hand-over of this kernel sits behind it)

// wave priority up from the end of a half-sweep's FMA block to the hand-over of its result (reduction chain, reciprocal, LDS write,
// barrier, the other half's start): the wave nearest to the hand-over gets the issue slots (as in sinkhorn_blk.hip).  PATS_W2_NO_PRIO: A/B
#ifndef PATS_W2_NO_PRIO
#define W2_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define W2_PRIO(n) do { } while (0)
#endif

struct __attribute__((aligned(16))) W2Lds {
    float va[2][8 * VA_S];       // per wave: the row-indexed vector of ITS 72 rows, entry 9 Iw + r at [Iw * 12 + r]
    float vb[2][8 * VB_S];       // per wave: its own copy of the column-indexed vector, entry 18 J + c at [J * 20 + c]
    float cx[2][NB][2];          // [sweep parity][column][wave]: column partials over a wave's 72 rows
    float red[2][2];             // [parity][wave]: the waves' partials of the dustbin column's sum
    float misc[8];
};

__device__ __forceinline__ void load9(const float* v, float (&o)[BR]) {
    const f4v a = *reinterpret_cast<const f4v*>(v), b = *reinterpret_cast<const f4v*>(v + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    o[8] = v[8];
}
__device__ __forceinline__ void load18(const float* v, float (&o)[BC]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f4v a = *reinterpret_cast<const f4v*>(v + 4 * q);
        o[4 * q] = a.x; o[4 * q + 1] = a.y; o[4 * q + 2] = a.z; o[4 * q + 3] = a.w;
    }
    const f2v e = *reinterpret_cast<const f2v*>(v + 16);
    o[16] = e.x; o[17] = e.y;
}

// nine values over the 8 consecutive lanes that share Iw: value J ends in lane J, the ninth in all eight
template <class Op>
__device__ __forceinline__ float rows_reduce9(const float (&p)[BR], Op op, int lane, float& ninth) {
    float q[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) q[v] = p[v];
    float e = p[8];
    e = op(e, dpp_f<DPP_ROW_HALF_MIRROR>(e));
    const float r = reduce8_consecutive(q, op, lane);
    e = op(e, dpp_f<DPP_QUAD_XOR2>(e));
    e = op(e, dpp_f<DPP_QUAD_XOR1>(e));
    ninth = e;
    return r;
}

// eighteen values over the 8 lanes that share J (lanes J, J + 8, ..): value Iw ends in lane Iw (cA), value 8 + Iw too (cB); values
// 16 / 17 are all-reduced into the lower / upper half of the wave (cC: lanes < 32 hold value 16, lanes >= 32 value 17).  The
// butterfly of lane_reduce.hpp's reduce8_strided on the two sets of eight and the pair AT ONCE, level by level: the swaps are opaque
// (volatile) to the compiler and stay in program order - one set after the other left every level's latency exposed.
template <class Op>
__device__ __forceinline__ void cols_reduce18(const float (&q)[BC], Op op, int lane, float& cA, float& cB, float& cC) {
    float x[9], y[9];
#pragma unroll
    for (int t = 0; t < 4; ++t) { x[t] = q[t]; y[t] = q[4 + t]; x[4 + t] = q[8 + t]; y[4 + t] = q[12 + t]; }
    x[8] = q[16]; y[8] = q[17];
#pragma unroll
    for (int t = 0; t < 9; ++t) swap32(x[t], y[t]);     // lanes < 32 keep index t of their set, lanes >= 32 index 4 + t (pair: 16 / 17)
    float r[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) r[t] = op(x[t], y[t]);
    // level 2: even 16-lane rows keep index s, odd rows 2 + s; the pair exchanges a copy of itself (all-reduce)
    float u[5], v[5];
    u[0] = r[0]; v[0] = r[2]; u[1] = r[1]; v[1] = r[3]; u[2] = r[4]; v[2] = r[6]; u[3] = r[5]; v[3] = r[7]; u[4] = r[8]; v[4] = r[8];
#pragma unroll
    for (int t = 0; t < 5; ++t) swap16(u[t], v[t]);
    float z[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) z[t] = op(u[t], v[t]);
    const bool b3 = lane & 8;
    const float keepA = b3 ? z[1] : z[0], sendA = b3 ? z[0] : z[1], keepB = b3 ? z[3] : z[2], sendB = b3 ? z[2] : z[3];
    cA = op(keepA, dpp_f<DPP_ROW_ROR8>(sendA));
    cB = op(keepB, dpp_f<DPP_ROW_ROR8>(sendB));
    cC = op(z[4], dpp_f<DPP_ROW_ROR8>(z[4]));
}

}  // namespace

// MODE 0: log_mu_in / log_nu_in [batch, 145]; MODE 2: log_optimal_transport2's marginals from ns [batch, 144] (and *one)
template <int MODE>
__global__ void __launch_bounds__(128, 2)
sinkhorn_blk145w2_kernel(const float* __restrict__ Zin, const float* __restrict__ log_mu_in, const float* __restrict__ log_nu_in,
                         const float* __restrict__ ns, const float* __restrict__ one, int iters, float bias_k,
                         float* __restrict__ out, int* __restrict__ fail, uint8_t* __restrict__ col_nomatch,
                         const int64_t* __restrict__ live) {
    __shared__ W2Lds lds;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), J = lane & 7, Iw = lane >> 3, I = 8 * w + Iw;
    const int64_t p = blockIdx.x;
    if (live && p >= *live) {                  // counted launch: a padding row - no solve, and no redo behind it either
        if (t == 0 && fail) fail[p] = 0;
        return;
    }
    const float* Zp = Zin + p * (N_ * N_);
    // ownership: every lane one row (9 I + J), the lanes J = 0 also row 9 I + 8; columns 18 J + Iw (A), 18 J + 8 + Iw (B) and, the
    // lanes Iw = 0 / 4, 18 J + 16 / 17 (C) - the same in both waves
    const bool own8 = J == 0, ownC = (Iw & 3) == 0;
    const int rowA = BR * I + J, row8 = BR * I + 8;
    const int cAi = Iw, cBi = 8 + Iw, cCi = 16 + (Iw >> 2);
    const int colA = BC * J + cAi, colB = BC * J + cBi, colC = BC * J + cCi;

    // ---- the block, the dustbin entries ---------------------------------------------------------------------------------
    float kb[BR][BC];
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        const float* row = Zp + (BR * I + r) * N_ + BC * J;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4a x = *reinterpret_cast<const f4a*>(row + 4 * q);
            kb[r][4 * q] = x.x; kb[r][4 * q + 1] = x.y; kb[r][4 * q + 2] = x.z; kb[r][4 * q + 3] = x.w;
        }
        kb[r][16] = row[16];
        kb[r][17] = row[17];
    }
    // Z[row][144]: the lane's own row, and row 9 I + 8 in all eight lanes of the group (one address: a broadcast load) so that its
    // scaling is formed identically in all of them; only J = 0 counts it in sums and stores it
    const float zdcA = Zp[rowA * N_ + NB], zdc8 = Zp[row8 * N_ + NB];
    const float zdrA = Zp[NB * N_ + colA], zdrB = Zp[NB * N_ + colB], zdrC = ownC ? Zp[NB * N_ + colC] : -INFINITY;   // Z[144][col]
    const float zcorner = uni(Zp[NB * N_ + NB]);

    // ---- marginals (modules.py:169-179) -------------------------------------------------------------------------------------
    float lmuA, lmu8, lnuA, lnuB, lnuC, lmu_d, lnu_d, norm = 0.f;
    if (MODE == 0) {
        lmuA = log_mu_in[p * N_ + rowA];
        lmu8 = log_mu_in[p * N_ + row8];
        lnuA = log_nu_in[p * N_ + colA];
        lnuB = log_nu_in[p * N_ + colB];
        lnuC = log_nu_in[p * N_ + (ownC ? colC : colA)];
        lmu_d = uni(log_mu_in[p * N_ + NB]);
        lnu_d = uni(log_nu_in[p * N_ + NB]);
    } else {
        const float nA = ns[p * NB + colA], nB = ns[p * NB + colB], nC = ownC ? ns[p * NB + colC] : 0.f;
        const float ns_sum = uni(wave_sum_xbar((nA + nB) + nC, lane));        // every column is owned once per wave
        const float ms = (float)NB * (one ? *one : 1.0f);
        norm = uni(-logf(ms + ns_sum));
        lmuA = norm;
        lmu8 = norm;
        lmu_d = uni(logf(ns_sum) + norm);
        lnuA = logf(nA) + norm;
        lnuB = logf(nB) + norm;
        lnuC = ownC ? logf(nC) + norm : 0.f;
        lnu_d = uni(logf(ms) + norm);
    }

    float* va = lds.va[w];
    float* vb = lds.vb[w];
    float* myA = &va[Iw * VA_S + J];
    float* my8 = &va[Iw * VA_S + 8];
    float* vbA = &vb[J * VB_S + cAi];
    float* vbB = &vb[J * VB_S + cBi];
    float* vbC = &vb[J * VB_S + cCi];

    // ---- stabilisers: r_i = max_j Z_ij, c_j = max_i (Z_ij - r_i), both over all 145 entries -------------------------------------
    float rA, r8, r_d, cA, cB, cC, c_d;
    float rloc[BR], cloc[BC];
    {
        float m[BR];
#pragma unroll
        for (int r = 0; r < BR; ++r) {
            float x = kb[r][0];
#pragma unroll
            for (int c = 1; c < BC; ++c) x = fmaxf(x, kb[r][c]);
            m[r] = x;
        }
        float e;
        rA = fmaxf(rows_reduce9(m, OpMax(), lane, e), zdcA);
        r8 = fmaxf(e, zdc8);
        *myA = rA;
        if (own8) *my8 = r8;
        // dustbin row: max over its 144 entries (every column owned once per wave) and the corner - each wave alone
        r_d = uni(fmaxf(wave_max(fmaxf(fmaxf(zdrA, zdrB), zdrC)), zcorner));
        wave_lds_sync();
        load9(&va[Iw * VA_S], rloc);
        // columns: partial maxima of (Z - r) over this lane's 9 rows, over the wave's 8 row groups, then over the two waves
        float cm[BC];
#pragma unroll
        for (int c = 0; c < BC; ++c) {
            float x = kb[0][c] - rloc[0];
#pragma unroll
            for (int r = 1; r < BR; ++r) x = fmaxf(x, kb[r][c] - rloc[r]);
            cm[c] = x;
        }
        float pA, pB, pC;
        cols_reduce18(cm, OpMax(), lane, pA, pB, pC);
        lds.cx[0][colA][w] = pA;
        lds.cx[0][colB][w] = pB;
        if (ownC) lds.cx[0][colC][w] = pC;
        const float wd = wave_max(fmaxf(zdcA - rA, own8 ? zdc8 - r8 : -INFINITY));
        if (lane == 0) lds.red[0][w] = wd;
        wg_barrier();
        const f2v qa = *reinterpret_cast<const f2v*>(lds.cx[0][colA]), qb = *reinterpret_cast<const f2v*>(lds.cx[0][colB]);
        const f2v qc = *reinterpret_cast<const f2v*>(lds.cx[0][ownC ? colC : colA]);
        cA = fmaxf(fmaxf(qa.x, qa.y), zdrA - r_d);
        cB = fmaxf(fmaxf(qb.x, qb.y), zdrB - r_d);
        cC = fmaxf(fmaxf(qc.x, qc.y), zdrC - r_d);                      // (meaningless where !ownC; never stored)
        const f2v qd = *reinterpret_cast<const f2v*>(lds.red[0]);
        c_d = uni(fmaxf(fmaxf(qd.x, qd.y), zcorner - r_d));
        *vbA = cA;
        *vbB = cB;
        if (ownC) *vbC = cC;
        wave_lds_sync();
        load18(&vb[J * VB_S], cloc);
        wave_lds_sync();                         // (cloc is read before b takes the same slots below)
    }
    // ---- K = exp(Z - r - c) ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int c = 0; c < BC; ++c) kb[r][c] = fast_exp2(((kb[r][c] - rloc[r]) - cloc[c]) * LOG2E);
    const float kdcA = fast_exp2(((zdcA - rA) - c_d) * LOG2E), kdc8 = fast_exp2(((zdc8 - r8) - c_d) * LOG2E);
    const float kdrA = fast_exp2(((zdrA - r_d) - cA) * LOG2E), kdrB = fast_exp2(((zdrB - r_d) - cB) * LOG2E);
    const float kdrC = ownC ? fast_exp2(((zdrC - r_d) - cC) * LOG2E) : 0.f;
    const float kcorner = uni(fast_exp2(((zcorner - r_d) - c_d) * LOG2E));
    const float muA = expf(lmuA), mu8 = expf(lmu8), nuA = expf(lnuA), nuB = expf(lnuB), nuC = ownC ? expf(lnuC) : 1.f;
    const float mu_d = uni(expf(lmu_d)), nu_d = uni(expf(lnu_d));
    float aA = 0.f, a8 = 0.f, a_d = 0.f, bA = expf(cA), bB = expf(cB), bC = ownC ? expf(cC) : 1.f, b_d = uni(expf(c_d));
    *vbA = bA;                                   // (this wave's copy: read by this wave only, LDS operations of a wave are in order)
    *vbB = bB;
    if (ownC) *vbC = bC;

    for (int it = 0; it < iters; ++it) {
        const int par = (it + 1) & 1;            // (parity 0 carried the stabilisers' exchange)
        {   // ---- a_i = mu_i / sum_j K_ij b_j ---------------------------------------------------------------------------------
            float bl[BC], part[BR];
            wave_lds_sync();                     // this wave's b - written by its owner lanes above / at the end of the last sweep
            load18(&vb[J * VB_S], bl);
            f2v acc[BR];
#pragma unroll
            for (int r = 0; r < BR; ++r) acc[r] = f2v{kb[r][0], kb[r][1]} * f2v{bl[0], bl[1]};
#pragma unroll
            for (int cc = 1; cc < BC / 2; ++cc)
#pragma unroll
                for (int r = 0; r < BR; ++r)
                    acc[r] = __builtin_elementwise_fma(f2v{kb[r][2 * cc], kb[r][2 * cc + 1]}, f2v{bl[2 * cc], bl[2 * cc + 1]}, acc[r]);
#pragma unroll
            for (int r = 0; r < BR; ++r) part[r] = acc[r].x + acc[r].y;
            W2_PRIO(1);
            // the dustbin row's sum over all columns: this wave owns every column once
            const float dsum = wave_sum_xbar(fmaf(kdrC, bC, fmaf(kdrB, bB, kdrA * bA)), lane);
            float e;
            const float s = fmaf(kdcA, b_d, rows_reduce9(part, OpSum(), lane, e));
            aA = mul_rcp(muA, s);
            a8 = mul_rcp(mu8, fmaf(kdc8, b_d, e));
            wave_lds_sync();                     // (every lane has read the last sweep's a)
            *myA = aA;
            if (own8) *my8 = a8;
            a_d = mul_rcp(mu_d, fmaf(kcorner, b_d, dsum));
            W2_PRIO(0);
        }
        {   // ---- b_j = nu_j / sum_i K_ij a_i ---------------------------------------------------------------------------------
            float al[BR];
            wave_lds_sync();                     // this wave's a
            load9(&va[Iw * VA_S], al);
            f2v q2[BC / 2];
#pragma unroll
            for (int cc = 0; cc < BC / 2; ++cc) q2[cc] = f2v{kb[0][2 * cc], kb[0][2 * cc + 1]} * f2v{al[0], al[0]};
#pragma unroll
            for (int r = 1; r < BR; ++r) {
                const f2v ar = {al[r], al[r]};
#pragma unroll
                for (int cc = 0; cc < BC / 2; ++cc) q2[cc] = __builtin_elementwise_fma(f2v{kb[r][2 * cc], kb[r][2 * cc + 1]}, ar, q2[cc]);
            }
            float q[BC];
#pragma unroll
            for (int cc = 0; cc < BC / 2; ++cc) { q[2 * cc] = q2[cc].x; q[2 * cc + 1] = q2[cc].y; }
            float pA, pB, pC;
            W2_PRIO(1);
            cols_reduce18(q, OpSum(), lane, pA, pB, pC);
            lds.cx[par][colA][w] = pA;
            lds.cx[par][colB][w] = pB;
            if (ownC) lds.cx[par][colC][w] = pC;
            const float dpart = wave_sum_xbar(fmaf(kdcA, aA, own8 ? kdc8 * a8 : 0.f), lane);
            if (lane == 0) lds.red[par][w] = dpart;
        }
        wg_barrier();                              // the ONE barrier of a sweep: both waves' column partials are in LDS
        {
            const f2v qa = *reinterpret_cast<const f2v*>(lds.cx[par][colA]), qb = *reinterpret_cast<const f2v*>(lds.cx[par][colB]);
            const f2v qc = *reinterpret_cast<const f2v*>(lds.cx[par][ownC ? colC : colA]);
            const f2v qd = *reinterpret_cast<const f2v*>(lds.red[par]);
            bA = mul_rcp(nuA, fmaf(kdrA, a_d, qa.x + qa.y));
            bB = mul_rcp(nuB, fmaf(kdrB, a_d, qb.x + qb.y));
            if (ownC) bC = mul_rcp(nuC, fmaf(kdrC, a_d, qc.x + qc.y));
            b_d = mul_rcp(nu_d, fmaf(kcorner, a_d, qd.x + qd.y));
            *vbA = bA;
            *vbB = bB;
            if (ownC) *vbC = bC;
            W2_PRIO(0);
        }
    }

    // ---- guard: every scaling finite, positive, <= 2^30 ----------------------------------------------------------------------------
    const bool okl = ok_scale(aA) && (!own8 || ok_scale(a8)) && ok_scale(bA) && ok_scale(bB) && (!ownC || ok_scale(bC)) && ok_scale(a_d) &&
                     ok_scale(b_d);
    const bool okw = __all(okl);
    wg_barrier();                                  // (the last sweep's reads of red[] are done)
    if (lane == 0) lds.misc[w] = okw ? 1.f : 0.f;
    wg_barrier();
    if (lds.misc[0] * lds.misc[1] < 0.5f) {
        if (t == 0) fail[p] = 1;
        return;
    }
    if (t == 0) fail[p] = 0;
    // ---- duals back to log space, Z_out = ((Z + u) + v) - norm (+ bias) from the original Z -------------------------------------------
    const float u_d = uni(logf(a_d) - r_d), v_d = uni(logf(b_d) - c_d);
    const float uA = logf(aA) - rA, u8 = logf(a8) - r8;
    const float vA = logf(bA) - cA, vB = logf(bB) - cB, vC = ownC ? logf(bC) - cC : 0.f;
    *myA = uA;
    if (own8) *my8 = u8;
    *vbA = vA;
    *vbB = vB;
    if (ownC) *vbC = vC;
    wave_lds_sync();
    float ul[BR], vl[BC];
    load9(&va[Iw * VA_S], ul);
    load18(&vb[J * VB_S], vl);
    const float lb = bias_k > 0.f ? logf(bias_k) : 0.f;
    float* Op = out + p * (N_ * N_);
    float cmx[BC];                     // column maxima of the OUTPUT over this lane's nine (real) rows
#pragma unroll
    for (int c = 0; c < BC; ++c) cmx[c] = -INFINITY;
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        const int e0 = (BR * I + r) * N_ + BC * J;
        float o[BC];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4a x = *reinterpret_cast<const f4a*>(Zp + e0 + 4 * q);
            o[4 * q] = x.x; o[4 * q + 1] = x.y; o[4 * q + 2] = x.z; o[4 * q + 3] = x.w;
        }
        o[16] = Zp[e0 + 16];
        o[17] = Zp[e0 + 17];
#pragma unroll
        for (int c = 0; c < BC; ++c) {
            o[c] = ((o[c] + ul[r]) + vl[c]) - norm;
            cmx[c] = fmaxf(cmx[c], o[c]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f4a*>(Op + e0 + 4 * q) = f4a{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
        Op[e0 + 16] = o[16];
        Op[e0 + 17] = o[17];
    }
    {   // the dustbin column: every lane its own row, J = 0 also row 9 I + 8
        float z = ((zdcA + uA) + v_d) - norm;
        if (bias_k > 0.f) z += lb;
        Op[rowA * N_ + NB] = z;
        if (own8) {
            float z8 = ((zdc8 + u8) + v_d) - norm;
            if (bias_k > 0.f) z8 += lb;
            Op[row8 * N_ + NB] = z8;
        }
    }
    // the dustbin row: both waves hold every column - wave 0 stores
    float zrA = ((zdrA + u_d) + vA) - norm, zrB = ((zdrB + u_d) + vB) - norm, zrC = ((zdrC + u_d) + vC) - norm;
    if (bias_k > 0.f) { zrA += lb; zrB += lb; zrC += lb; }
    if (w == 0) {
        Op[NB * N_ + colA] = zrA;
        Op[NB * N_ + colB] = zrB;
        if (ownC) Op[NB * N_ + colC] = zrC;
    }
    if (col_nomatch) {
        // est_position's if_nomatching2 = (scores.max(1).indices == 144), second_layer.py:243,248: the dustbin row strictly above
        // every real row of the column (first index wins ties)
        float pA, pB, pC;
        cols_reduce18(cmx, OpMax(), lane, pA, pB, pC);
        wg_barrier();                              // (cx[0] was last read before the guard's barriers; this keeps the pattern plain)
        lds.cx[0][colA][w] = pA;
        lds.cx[0][colB][w] = pB;
        if (ownC) lds.cx[0][colC][w] = pC;
        wg_barrier();
        if (w == 0) {
            const f2v qa = *reinterpret_cast<const f2v*>(lds.cx[0][colA]), qb = *reinterpret_cast<const f2v*>(lds.cx[0][colB]);
            col_nomatch[p * NB + colA] = zrA > fmaxf(qa.x, qa.y);
            col_nomatch[p * NB + colB] = zrB > fmaxf(qb.x, qb.y);
            if (ownC) {
                const f2v qc = *reinterpret_cast<const f2v*>(lds.cx[0][colC]);
                col_nomatch[p * NB + colC] = zrC > fmaxf(qc.x, qc.y);
            }
        }
    }
    if (t == 0) {
        float z = ((zcorner + u_d) + v_d) - norm;
        if (bias_k > 0.f) { z += lb; z += lb; }
        Op[NB * N_ + NB] = z;
    }
}

// PATS_FINE_W2 (read once per process): 0 = the four-wave kernel of sinkhorn_blk.hip
bool fine_w2_enabled() {
    static const bool on = [] { const char* e = env_switch("PATS_FINE_W2"); return !(e && atoi(e) == 0); }();
    return on;
}

int launch_blk145_w2(int mode, const float* Z, int64_t batch, const float* log_mu, const float* log_nu, const float* ns,
                     const float* one, int iters, float bias_k, float* out, int* fail, uint8_t* col_nomatch, hipStream_t st,
                     const int64_t* live) {
    if (mode == 0)
        hipLaunchKernelGGL((sinkhorn_blk145w2_kernel<0>), dim3((unsigned)batch), dim3(128), 0, st, Z, log_mu, log_nu,
                           (const float*)nullptr, (const float*)nullptr, iters, 0.f, out, fail, col_nomatch, live);
    else
        hipLaunchKernelGGL((sinkhorn_blk145w2_kernel<2>), dim3((unsigned)batch), dim3(128), 0, st, Z, (const float*)nullptr,
                           (const float*)nullptr, ns, one, iters, bias_k, out, fail, col_nomatch, live);
    return check_launch("sinkhorn_blk145w2_kernel");
}

}  // namespace pats
