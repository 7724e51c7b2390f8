// Fine-level Sinkhorn (145 x 145 = 12 x 12 cells + dustbin) with the coupling matrix held in
// registers as 9 x 9 blocks.   models/modules.py:137-143 (sweeps), :165-182 (marginals of
// log_optimal_transport2), second_layer.py:107-112 (dustbin bias folded into the epilogue).
//
// One 256-thread workgroup per problem = a 16 x 16 grid of lanes; lane (I, J) (J = t & 15 is the
// position inside its 16-lane DPP row, I = t >> 4) owns K[9I..9I+8][9J..9J+8] in 81 VGPRs.  The
// dustbin column entry K[i][144] of row i = 9I + J sits in lane (I, J < 9), the dustbin row entry
// K[144][j] of column j = 9J + I in lane (I < 9, J); those lanes also own a_i resp. b_j.
// A sweep (linear domain, see sinkhorn.hip for the equivalence with the reference's log-sum-exp
// form and for the guard):
//   rows   : 81 FMAs on the block against b[9J..9J+8] (three broadcast ds_read_b128), then the 9
//            partials are reduced over the 16 lanes of the DPP row - one row_ror:8 step on all nine,
//            the 8-value butterfly of lane_reduce.hpp, three steps for the ninth;
//   columns: 81 FMAs against a[9I..9I+8], partials reduced over the four DPP rows of the wave with
//            v_permlane32/16_swap (two values per swap), then over the four waves through LDS;
//   the 145th row / column: one product per lane, wave sum, four partials through LDS.
// Against sinkhorn_rc_kernel (a full row or column per lane, 145 broadcast floats per lane per
// half-sweep, 85 KB of LDS => one workgroup per CU) this moves 12 instead of 145 floats per lane per
// half-sweep through LDS and needs 13 KB of LDS, so three workgroups share a CU and the sweeps are
// VALU-bound.  A problem whose scalings leave the guard band sets fail[p]; the host re-runs those
// with sinkhorn_rc_kernel's log-sum-exp sweeps.
#include "lane_reduce.hpp"
#include "mfma_tile.hpp"
#include <stdlib.h>

namespace pats {

namespace {

constexpr int N_ = 145, NB = 144, BS = 9, VS = 12;     // VS: padded stride of a 9-float group in LDS
constexpr float BLK_GUARD = 1073741824.0f;              // 2^30, as sinkhorn.hip
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool ok_scale(float x) { return x <= BLK_GUARD && x > 0.f; }
__device__ __forceinline__ float uni(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

struct __attribute__((aligned(16))) BlkLds {
    float va[16 * VS];           // row-indexed vector: entry 9I + r at [I * VS + r]
    float vb[16 * VS];           // column-indexed vector: entry 9J + c at [J * VS + c]
    float cpart[NB * 4];         // cross-wave column partials, [column][wave]
    float red_r[4], red_c[4];    // per-wave partials of the dustbin row / column sums
    float misc[8];
    float tmp[NB * 17];          // one-time: partial maxima for the stabilisers
};

// experiment hooks (libpats_amd_diag<suffix>.so, tools/fine_determinism.py)
// Wave priority up for the dependent reduction tail of the row and column phases (DPP / swap chains, reciprocal, the LDS
// write the other waves wait for at the barrier), back down for the FMA blocks: the waves closest to a barrier get the issue
// slots.  Measured 5.58 -> 5.50 ms per 20 224 problems for cost + OT, levels 1 / 2 / 3 alike; no effect
// on results.  -DPATS_EXPB_NO_PRIO: the A/B partner.
#ifndef PATS_EXPB_NO_PRIO
#define BLK_PRIO(n) __builtin_amdgcn_s_setprio((n) ? 1 : 0)
#else
#define BLK_PRIO(n) do { } while (0)
#endif
#ifdef PATS_EXPB_WSUM
#define BLK_WSUM(x) wave_sum(x)                 // crossbar all-reduce instead of row_bcast DPP + v_readlane
#else
#define BLK_WSUM(x) wave_sum_uniform(x)
#endif
typedef float f2e __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2e blk_fma2(f2e a, f2e b, f2e c) {
#ifdef PATS_EXPB_NOPK
    return f2e{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)};      // two v_fma_f32 instead of v_pk_fma_f32 (build with -fno-slp-vectorize)
#else
    return __builtin_elementwise_fma(a, b, c);
#endif
}
__device__ __forceinline__ f2e blk_mul2(f2e a, f2e b) {
#ifdef PATS_EXPB_NOPK
    return f2e{a.x * b.x, a.y * b.y};
#else
    return a * b;
#endif
}

__device__ __forceinline__ float mul_rcp(float num, float den) { return num * __builtin_amdgcn_rcpf(den); }

__device__ __forceinline__ void load9(const float* v, float (&o)[BS]) {
    const f4v a = *reinterpret_cast<const f4v*>(v), b = *reinterpret_cast<const f4v*>(v + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    o[8] = v[8];
}

// sum over the 16 lanes of a DPP row of nine values; value J lands in lane J (J < 9)
__device__ __forceinline__ float row16_reduce9(const float (&p)[BS], int lane) {
    float q[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) q[v] = p[v] + dpp_f<DPP_ROW_ROR8>(p[v]);
    float e = p[8] + dpp_f<DPP_ROW_ROR8>(p[8]);
    const float r = reduce8_consecutive(q, OpSum(), lane);
    e += dpp_f<DPP_ROW_HALF_MIRROR>(e);
    e += dpp_f<DPP_QUAD_XOR2>(e);
    e += dpp_f<DPP_QUAD_XOR1>(e);
    return (lane & 15) == 8 ? e : r;
}

// all-reduce max over the 16 lanes of a DPP row
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    return v;
}

// ---- FUSED: the score matrix built in the kernel (second_layer.py:100-101,104 -> :105 in one launch) ---------------------
// The workgroup first runs the 160 x 160 MFMA tile of mfma_tile.hpp on the problem's two [D,145] descriptor blocks (the
// same code, the same bits as cost_mfma_kernel), then turns the accumulator fragments into the 9 x 9 register blocks of
// the sweeps through a 32-row band buffer in LDS (the staging area of the cost build, free by then): tile row b of the
// five is written by the wave(s) that hold it, every lane picks up the rows of its block that lie in the band.  The scores
// never reach HBM; the lane keeps its block (Z, 81 registers) beside K for the epilogue's ((Z + u) + v) - norm, which is why
// this variant is built for two workgroups per CU (256 VGPRs): tools/fine_fusion_probe.py measured the sweeps of the
// unfused kernel at that occupancy within 1-3 % of three workgroups per CU.
struct FineCols {
    static constexpr bool stream = false;
    __device__ __forceinline__ int64_t a_off(int c) const { return c < N_ ? c : N_ - 1; }
    __device__ __forceinline__ int64_t b_off(int c) const { return c < N_ ? c : N_ - 1; }
    __device__ __forceinline__ bool row_stored(int r) const { return r < N_; }
    __device__ __forceinline__ bool col_stored(int c) const { return c < N_; }
};
constexpr int BAND_LD = 164;
union __attribute__((aligned(16))) FusedLds {
    mt::Lds cost;
    float band[32 * BAND_LD];
};

__device__ __forceinline__ void fine_cost_block(const float* __restrict__ A, const float* __restrict__ B, int D, float rsqrtD,
                                                float sqrtD, FusedLds& fl, float (&zb)[BS][BS], float& zdc, float& zdr,
                                                float& zcorner, int t) {
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lk = lane >> 5;
    const int J = t & 15, I = t >> 4;
    const bool rown = J < BS, cown = I < BS;
    const int rowi = BS * I + (rown ? J : 0), colj = BS * J + (cown ? I : 0);
    const FineCols cols;
    mt::CmSrc<FineCols> src(A, N_, B, N_, D, cols, t);
    mt::f32x16 acc[7];
    const float unscale = mt::tile<true, true>(src, (mt::CmSrc<FineCols>*)nullptr, fl.cost, acc, true, t, wave);
    // `scores / D ** .5`, then `0.1 * scores`, exactly as cost_mfma_kernel stores them
    auto scaled = [&](float x) { return 0.1f * div_invariant(x * unscale, sqrtD, rsqrtD); };
    zdc = -INFINITY;
    zdr = -INFINITY;
    zcorner = 0.f;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        wg_barrier();                                  // the band buffer (b == 0: the cost build's staging area) is free
        if (b < 4) {
            if (wave == b) {
#pragma unroll
                for (int tj = 0; tj < 5; ++tj)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        fl.band[((r & 3) + 8 * (r >> 2) + 4 * lk) * BAND_LD + 32 * tj + li] = scaled(acc[tj][r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) fl.band[((r & 3) + 8 * (r >> 2) + 4 * lk) * BAND_LD + 32 * wave + li] = scaled(acc[5][r]);
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) fl.band[((r & 3) + 8 * (r >> 2) + 4 * lk) * BAND_LD + 128 + li] = scaled(acc[6][r]);
            }
        }
        wg_barrier();
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            const int row = BS * I + r;
            if ((row >> 5) == b) {
                const float* src_row = &fl.band[(row - 32 * b) * BAND_LD + BS * J];
#pragma unroll
                for (int c = 0; c < BS; ++c) zb[r][c] = src_row[c];
            }
        }
        if (rown && (rowi >> 5) == b) zdc = fl.band[(rowi - 32 * b) * BAND_LD + NB];
        if (b == 4) {
            if (cown) zdr = fl.band[(NB - 128) * BAND_LD + colj];
            zcorner = fl.band[(NB - 128) * BAND_LD + NB];
        }
    }
    wg_barrier();
}

}  // namespace

// MODE 0: log_mu / log_nu given (a6)      MODE 2: ns given, log_optimal_transport2 marginals (a5)
// FUSED: Zin unused, the scores come from the descriptor blocks d0, d1 [batch, D, 145] (see fine_cost_block)
#ifndef PATS_BLK_WAVES
#define PATS_BLK_WAVES 3         // workgroups per CU the register budget is cut for (diagnostic builds: 4 = 128 VGPRs, 25 of them spilled)
#endif
template <int MODE, bool FUSED = false>
__global__ void __launch_bounds__(256, FUSED ? 2 : PATS_BLK_WAVES)
sinkhorn_blk145_kernel(const float* __restrict__ Zin, const float* __restrict__ log_mu_in,
                       const float* __restrict__ log_nu_in, const float* __restrict__ ns,
                       const float* __restrict__ one, int iters, float bias_k, float* __restrict__ out,
                       int* __restrict__ fail, uint8_t* __restrict__ col_nomatch,
                       const float* __restrict__ d0 = nullptr, const float* __restrict__ d1 = nullptr, int D = 0,
                       float rsqrtD = 0.f, float sqrtD = 0.f, const int64_t* __restrict__ live = nullptr) {
    __shared__ BlkLds lds;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, J = t & 15, I = t >> 4, rho = lane >> 4;
    const int64_t p = blockIdx.x;
    if (live && p >= *live) {                  // counted launch: a padding row - no solve, and no redo behind it either
        if (t == 0 && fail) fail[p] = 0;
        return;
    }
    const float* Zp = Zin + p * (N_ * N_);
    const bool rown = J < BS, cown = I < BS;          // owns a row entry a_(9I+J) / a column entry b_(9J+I)
    const int rowi = BS * I + (rown ? J : 0), colj = BS * J + (cown ? I : 0);

    // ---- the block, the dustbin entries -----------------------------------------------------------
    // a block row is 36 contiguous bytes: 16 + 16 + 4 byte accesses (only 4-byte aligned).  With nine 4-byte accesses
    // per row every instruction of a wave touched 64 separate 4-byte pieces at a 36-byte stride: in isolation the
    // 84 KB of a problem then read at a third and wrote at half the rate (tools/store_patterns.hip, patterns g / g4)
    typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));
    float kb[BS][BS];
    float zb[FUSED ? BS : 1][FUSED ? BS : 1];          // FUSED: the scores of this lane's block, kept for the epilogue
    float zdc, zdr, zcorner;                            // Z[9I+J][144], Z[144][9J+I], Z[144][144]
    if constexpr (FUSED) {
        __shared__ FusedLds fl;
        float zc;
        fine_cost_block(d0 + p * (int64_t)D * N_, d1 + p * (int64_t)D * N_, D, rsqrtD, sqrtD, fl, zb, zdc, zdr, zc, t);
        zcorner = uni(zc);
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c) kb[r][c] = zb[r][c];
    } else {
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            const float* row = Zp + (BS * I + r) * N_ + BS * J;
            const f4a x0 = *reinterpret_cast<const f4a*>(row), x1 = *reinterpret_cast<const f4a*>(row + 4);
            kb[r][0] = x0.x; kb[r][1] = x0.y; kb[r][2] = x0.z; kb[r][3] = x0.w;
            kb[r][4] = x1.x; kb[r][5] = x1.y; kb[r][6] = x1.z; kb[r][7] = x1.w;
            kb[r][8] = row[8];
        }
        zdc = rown ? Zp[rowi * N_ + NB] : -INFINITY;
        zdr = cown ? Zp[NB * N_ + colj] : -INFINITY;
        zcorner = uni(Zp[NB * N_ + NB]);
    }

    // ---- marginals (modules.py:169-179) -------------------------------------------------------------
    float lmu, lnu, lmu_d, lnu_d, norm = 0.f;
    if (MODE == 0) {
        lmu = log_mu_in[p * N_ + rowi];
        lnu = log_nu_in[p * N_ + colj];
        lmu_d = uni(log_mu_in[p * N_ + NB]);
        lnu_d = uni(log_nu_in[p * N_ + NB]);
    } else {
        const float part = wave_sum_uniform(t < NB ? ns[p * NB + t] : 0.f);
        if (lane == 0) lds.misc[wave] = part;
        wg_barrier();
        const float ns_sum = (lds.misc[0] + lds.misc[1]) + (lds.misc[2] + lds.misc[3]);
        const float ms = (float)NB * (one ? *one : 1.0f);
        norm = uni(-logf(ms + ns_sum));
        lmu = norm;
        lmu_d = uni(logf(ns_sum) + norm);
        lnu = logf(ns[p * NB + colj]) + norm;
        lnu_d = uni(logf(ms) + norm);
    }

    // ---- stabilisers: r_i = max_j Z_ij, c_j = max_i (Z_ij - r_i), both over all 145 entries ---------
    float r_own, c_own, r_d, c_d;
    float rloc[BS], cloc[BS];
    {
        float m[BS];
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            float x = kb[r][0];
#pragma unroll
            for (int c = 1; c < BS; ++c) x = fmaxf(x, kb[r][c]);
            m[r] = row16_max(x);                       // every lane of the DPP row now holds the row maximum
        }
        // this lane's own row is m[J]: pick it without dynamic register indexing
        float mine = m[0];
#pragma unroll
        for (int r = 1; r < BS; ++r) mine = (J == r) ? m[r] : mine;
        r_own = fmaxf(mine, zdc);
        // dustbin row: max over its 144 entries and the corner
        const float wm = wave_max(zdr);
        if (lane == 0) lds.misc[4 + wave] = wm;
        if (rown) lds.va[I * VS + J] = r_own;
        wg_barrier();
        r_d = uni(fmaxf(fmaxf(fmaxf(lds.misc[4], lds.misc[5]), fmaxf(lds.misc[6], lds.misc[7])), zcorner));
        load9(&lds.va[I * VS], rloc);
        // columns: partial maxima of (Z - r) over this lane's 9 rows, reduced over the 16 values of I via LDS
#pragma unroll
        for (int c = 0; c < BS; ++c) {
            float x = kb[0][c] - rloc[0];
#pragma unroll
            for (int r = 1; r < BS; ++r) x = fmaxf(x, kb[r][c] - rloc[r]);
            lds.tmp[(BS * J + c) * 17 + I] = x;
        }
        const float wd = wave_max(rown ? zdc - r_own : -INFINITY);
        wg_barrier();
        if (lane == 0) lds.misc[wave] = wd;
        float x = -INFINITY;
        if (cown) {
#pragma unroll
            for (int k = 0; k < 16; ++k) x = fmaxf(x, lds.tmp[colj * 17 + k]);
            x = fmaxf(x, zdr - r_d);
            lds.vb[J * VS + I] = x;
        }
        c_own = x;
        wg_barrier();
        c_d = uni(fmaxf(fmaxf(fmaxf(lds.misc[0], lds.misc[1]), fmaxf(lds.misc[2], lds.misc[3])), zcorner - r_d));
        load9(&lds.vb[J * VS], cloc);
    }
    // ---- K = exp(Z - r - c) ---------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c) kb[r][c] = fast_exp2(((kb[r][c] - rloc[r]) - cloc[c]) * LOG2E);
    const float kdc = rown ? fast_exp2(((zdc - r_own) - c_d) * LOG2E) : 0.f;
    const float kdr = cown ? fast_exp2(((zdr - r_d) - c_own) * LOG2E) : 0.f;
    const float kcorner = uni(fast_exp2(((zcorner - r_d) - c_d) * LOG2E));
    const float mu = expf(lmu), nu = expf(lnu), mu_d = uni(expf(lmu_d)), nu_d = uni(expf(lnu_d));
    float a = 0.f, b = cown ? expf(c_own) : 0.f, a_d = 0.f, b_d = uni(expf(c_d));
    wg_barrier();                                  // everyone has read the stabilisers
    if (cown) lds.vb[J * VS + I] = b;

    for (int it = 0; it < iters; ++it) {
        wg_barrier();                              // b visible
        {   // ---- a_i = mu_i / sum_j K_ij b_j -----------------------------------------------------
            float bl[BS], part[BS];
            load9(&lds.vb[J * VS], bl);
            const f2v b01 = {bl[0], bl[1]}, b23 = {bl[2], bl[3]}, b45 = {bl[4], bl[5]}, b67 = {bl[6], bl[7]};
            f2v acc[BS];
#pragma unroll
            for (int r = 0; r < BS; ++r) acc[r] = blk_mul2(f2v{kb[r][0], kb[r][1]}, b01);
#pragma unroll
            for (int r = 0; r < BS; ++r) acc[r] = blk_fma2(f2v{kb[r][2], kb[r][3]}, b23, acc[r]);
#pragma unroll
            for (int r = 0; r < BS; ++r) acc[r] = blk_fma2(f2v{kb[r][4], kb[r][5]}, b45, acc[r]);
#pragma unroll
            for (int r = 0; r < BS; ++r) acc[r] = blk_fma2(f2v{kb[r][6], kb[r][7]}, b67, acc[r]);
#pragma unroll
            for (int r = 0; r < BS; ++r) part[r] = fmaf(kb[r][8], bl[8], acc[r].x) + acc[r].y;
            BLK_PRIO(2);
            const float dsum = BLK_WSUM(kdr * b);
            const float s = fmaf(kdc, b_d, row16_reduce9(part, lane));
            a = mul_rcp(mu, s);
            if (rown) lds.va[I * VS + J] = a;
            if (lane == 0) lds.red_r[wave] = dsum;
            BLK_PRIO(0);
        }
        wg_barrier();                              // a and the dustbin-row partials visible
        {   // ---- b_j = nu_j / sum_i K_ij a_i -----------------------------------------------------
            const f4v dr = *reinterpret_cast<const f4v*>(lds.red_r);
            a_d = mul_rcp(mu_d, fmaf(kcorner, b_d, (dr.x + dr.y) + (dr.z + dr.w)));
            float al[BS];
            load9(&lds.va[I * VS], al);
            f2v q01 = {0.f, 0.f}, q23 = {0.f, 0.f}, q45 = {0.f, 0.f}, q67 = {0.f, 0.f};
            float q8 = 0.f;
#pragma unroll
            for (int r = 0; r < BS; ++r) {
                const f2v ar = {al[r], al[r]};
                q01 = blk_fma2(f2v{kb[r][0], kb[r][1]}, ar, q01);
                q23 = blk_fma2(f2v{kb[r][2], kb[r][3]}, ar, q23);
                q45 = blk_fma2(f2v{kb[r][4], kb[r][5]}, ar, q45);
                q67 = blk_fma2(f2v{kb[r][6], kb[r][7]}, ar, q67);
                q8 = fmaf(kb[r][8], al[r], q8);
            }
            // over the wave's four DPP rows: permlane32 pairs (0,1)(2,3)(4,5)(6,7), then permlane16 pairs;
            // row rho ends with columns {0,2,1,3}[rho] and that + 4; column 8 is summed in every row
            BLK_PRIO(2);
            float w0, w1, w2, w3, w4;
            { float x = q01.x, y = q01.y; swap32(x, y); w0 = x + y; }
            { float x = q23.x, y = q23.y; swap32(x, y); w1 = x + y; }
            { float x = q45.x, y = q45.y; swap32(x, y); w2 = x + y; }
            { float x = q67.x, y = q67.y; swap32(x, y); w3 = x + y; }
            { float x = q8, y = q8; swap32(x, y); w4 = x + y; }
            float z0, z1, z2;
            { float x = w0, y = w1; swap16(x, y); z0 = x + y; }
            { float x = w2, y = w3; swap16(x, y); z1 = x + y; }
            { float x = w4, y = w4; swap16(x, y); z2 = x + y; }
            const int cA = ((rho & 1) << 1) | (rho >> 1);        // 0, 2, 1, 3
            lds.cpart[(BS * J + cA) * 4 + wave] = z0;
            lds.cpart[(BS * J + cA + 4) * 4 + wave] = z1;
            if (rho == 0) lds.cpart[(BS * J + 8) * 4 + wave] = z2;
            const float dsum = BLK_WSUM(kdc * a);
            if (lane == 0) lds.red_c[wave] = dsum;
            BLK_PRIO(0);
        }
        wg_barrier();                              // column partials visible
        {
            const f4v dc = *reinterpret_cast<const f4v*>(lds.red_c);
            b_d = mul_rcp(nu_d, fmaf(kcorner, a_d, (dc.x + dc.y) + (dc.z + dc.w)));
            const f4v cp = *reinterpret_cast<const f4v*>(&lds.cpart[colj * 4]);
            const float tsum = fmaf(kdr, a_d, (cp.x + cp.y) + (cp.z + cp.w));
            b = mul_rcp(nu, tsum);
            if (cown) lds.vb[J * VS + I] = b;
        }
    }

    // ---- guard: every scaling finite, positive, <= 2^30 ---------------------------------------------------
    const bool okl = (!rown || ok_scale(a)) && (!cown || ok_scale(b)) && ok_scale(a_d) && ok_scale(b_d);
    const bool okw = __all(okl);
    wg_barrier();
    if (lane == 0) lds.misc[wave] = okw ? 1.f : 0.f;
    wg_barrier();
    if ((lds.misc[0] * lds.misc[1]) * (lds.misc[2] * lds.misc[3]) < 0.5f) {
        if (t == 0) fail[p] = 1;
        if constexpr (FUSED) {
            // the log-domain kernel redoes this problem from its scores: they exist only here, so they go where the plan would
            // have gone (sinkhorn_rc_kernel stages the whole matrix in LDS before it writes: in place is safe)
            float* Sp = out + p * (N_ * N_);
#pragma unroll
            for (int r = 0; r < BS; ++r)
#pragma unroll
                for (int c = 0; c < BS; ++c) Sp[(BS * I + r) * N_ + BS * J + c] = zb[r][c];
            if (rown) Sp[rowi * N_ + NB] = zdc;
            if (cown) Sp[NB * N_ + colj] = zdr;
            if (t == 0) Sp[NB * N_ + NB] = zcorner;
        }
        return;
    }
    if (t == 0) fail[p] = 0;
    // ---- duals back to log space, Z_out = ((Z + u) + v) - norm (+ bias) from the original Z -----------------
    const float u_d = uni(logf(a_d) - r_d), v_d = uni(logf(b_d) - c_d);
    if (rown) lds.va[I * VS + J] = logf(a) - r_own;
    if (cown) lds.vb[J * VS + I] = logf(b) - c_own;
    wg_barrier();
    float ul[BS], vl[BS];
    load9(&lds.va[I * VS], ul);
    load9(&lds.vb[J * VS], vl);
    const float lb = bias_k > 0.f ? logf(bias_k) : 0.f;
    float* Op = out + p * (N_ * N_);
    float cm[BS];                      // column maxima of the OUTPUT over this lane's nine (real) rows
#pragma unroll
    for (int c = 0; c < BS; ++c) cm[c] = -INFINITY;
#pragma unroll
    for (int r = 0; r < BS; ++r) {
        const int e0 = (BS * I + r) * N_ + BS * J;
        float zin[BS];
        if constexpr (FUSED) {
#pragma unroll
            for (int c = 0; c < BS; ++c) zin[c] = zb[r][c];
        } else {
            const f4a x0 = *reinterpret_cast<const f4a*>(Zp + e0), x1 = *reinterpret_cast<const f4a*>(Zp + e0 + 4);
            zin[0] = x0.x; zin[1] = x0.y; zin[2] = x0.z; zin[3] = x0.w;
            zin[4] = x1.x; zin[5] = x1.y; zin[6] = x1.z; zin[7] = x1.w;
            zin[8] = Zp[e0 + 8];
        }
        float o[BS];
#pragma unroll
        for (int c = 0; c < BS; ++c) {
            o[c] = ((zin[c] + ul[r]) + vl[c]) - norm;
            cm[c] = fmaxf(cm[c], o[c]);
        }
        *reinterpret_cast<f4a*>(Op + e0) = f4a{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f4a*>(Op + e0 + 4) = f4a{o[4], o[5], o[6], o[7]};
        Op[e0 + 8] = o[8];
    }
    if (rown) {
        float z = ((zdc + (logf(a) - r_own)) + v_d) - norm;
        if (bias_k > 0.f) z += lb;
        Op[rowi * N_ + NB] = z;
    }
    float zrow = 0.f;
    if (cown) {
        float z = ((zdr + u_d) + (logf(b) - c_own)) - norm;
        if (bias_k > 0.f) z += lb;
        Op[NB * N_ + colj] = z;
        zrow = z;
    }
    if (col_nomatch) {
        // est_position's if_nomatching2 = (scores.max(1).indices == 144), second_layer.py:243,248: the dustbin row
        // strictly above every real row of the column (first index wins ties); partial maxima cross the 16 lanes
        // that share J through LDS, as for the stabilisers
        wg_barrier();
#pragma unroll
        for (int c = 0; c < BS; ++c) lds.tmp[(BS * J + c) * 17 + I] = cm[c];
        wg_barrier();
        if (cown) {
            float x = lds.tmp[colj * 17];
#pragma unroll
            for (int k = 1; k < 16; ++k) x = fmaxf(x, lds.tmp[colj * 17 + k]);
            col_nomatch[p * NB + colj] = zrow > x;
        }
    }
    if (t == 0) {
        float z = ((zcorner + u_d) + v_d) - norm;
        if (bias_k > 0.f) { z += lb; z += lb; }
        Op[NB * N_ + NB] = z;
    }
}

// fail must hold `batch` ints
int launch_blk145(int mode, const float* Z, int64_t batch, const float* log_mu, const float* log_nu,
                  const float* ns, const float* one, int iters, float bias_k, float* out, int* fail,
                  uint8_t* col_nomatch, hipStream_t st, const int64_t* live) {
    // libpats_amd_diag.so only: dynamic LDS the kernel never touches lowers the occupancy (13.7 KB per workgroup: three per
    // CU by registers; + 41 KB = the cost build's staging area -> two) - what the sweeps would cost inside a kernel that
    // also holds the cost build's LDS and registers ("fine-level fusion")
    unsigned pad = 0;
#ifdef PATS_DIAG
    if (const char* e = diag_env("PATS_BLK_LDS_PAD")) pad = (unsigned)atoi(e);
#endif
    if (mode == 0)
        hipLaunchKernelGGL((sinkhorn_blk145_kernel<0>), dim3((unsigned)batch), dim3(256), pad, st, Z, log_mu, log_nu,
                           (const float*)nullptr, (const float*)nullptr, iters, 0.f, out, fail, col_nomatch,
                           (const float*)nullptr, (const float*)nullptr, 0, 0.f, 0.f, live);
    else
        hipLaunchKernelGGL((sinkhorn_blk145_kernel<2>), dim3((unsigned)batch), dim3(256), pad, st, Z,
                           (const float*)nullptr, (const float*)nullptr, ns, one, iters, bias_k, out, fail, col_nomatch,
                           (const float*)nullptr, (const float*)nullptr, 0, 0.f, 0.f, live);
    return check_launch("sinkhorn_blk145_kernel");
}

// the fused fine-level step: descriptors [batch, D, 145] x 2 -> log-plan, log_optimal_transport2 marginals (MODE 2)
int launch_blk145_fused(const float* d0, const float* d1, int D, int64_t batch, const float* ns, const float* one, int iters,
                        float bias_k, float* out, int* fail, uint8_t* col_nomatch, hipStream_t st, const int64_t* live) {
    const float sq = (float)sqrt((double)D);
    hipLaunchKernelGGL((sinkhorn_blk145_kernel<2, true>), dim3((unsigned)batch), dim3(256), 0, st, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, ns, one, iters, bias_k, out, fail, col_nomatch, d0, d1, D,
                       1.0f / sq, sq, live);
    return check_launch("sinkhorn_blk145_kernel<2, fused>");
}

}  // namespace pats
