// The whole third-level step of PATS as ONE kernel in an 8x8 register-block layout (gfx950).
//
//   models/third_layer.py:153-170:  cost build (einsum, /sqrt(128), 0.1*) -> log_optimal_transport2
//   (100 sweeps) -> exp -> Compute_result (:184-217) -> label (:161-170);  one wave per problem.
//
// Why a second layout next to sinkhorn65_kernel (sinkhorn.hip): that kernel keeps a row AND a
// column of K per lane (128 VGPRs) and broadcasts the 64-entry scaling vector from LDS every
// half-sweep (32 ds_read_b128 per sweep per wave) with the 65x65 plan resident in 17 KB of LDS:
// 2 waves per SIMD, and measured latency-bound (waves phase-lock; HBM-, MFMA- and VALU-bound
// phases add instead of overlapping).  Here lane (I, J) = (lane >> 3, lane & 7) owns the 8x8 block
// K[8I..8I+7][8J..8J+7] - 64 VGPRs, ONE copy serving both half-sweeps:
//   row half-sweep:  8 partial dots with b[8J..8J+7] (two ds_read_b128), reduced over the 8 lanes
//                    that share I (reduce8_consecutive) -> lane (I,J) gets the sum of row 8I+J
//   col half-sweep:  8 partial dots with a[8I..8I+7], reduced over the 8 lanes that share J
//                    (reduce8_strided) -> lane (I,J) gets the sum of column 8J+I
// 8x less LDS traffic, half the registers, ~6 KB of LDS per wave (a 4.6 KB staging buffer for the
// MFMA-fragment -> block-layout redistribution, reused for the 16 centre rows of the plan that
// Compute_result reads): 3 waves per SIMD.  The dustbin row / column (index 64) stay as one
// value per lane plus a 64-lane reduction, as in sinkhorn65_kernel.  Same linear-domain iteration,
// same guard; a problem that trips the guard (or PATS_SINKHORN_LOG) redoes its cost build and runs
// max-subtracted log-sum-exp sweeps in the same layout.
#include "common.hpp"
#include "lane_reduce.hpp"
#include "third_device.hpp"
#include "cost65_device.hpp"
#include <stdlib.h>

namespace pats {

int sinkhorn_mode();

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

constexpr int SST = 36;                 // staging row stride (floats): 16-byte aligned rows
constexpr float GUARD = 1073741824.0f;  // 2^30
__device__ __forceinline__ bool sc_ok(float x) { return x <= GUARD && x > 0.f; }

struct __attribute__((aligned(16))) BlkLds {
    float va[72];                // row-indexed vector (r, then a / u); [64] = dustbin row's value
    float vb[72];                // column-indexed vector (c, then b / v); [64] = dustbin column's value
    float erow[72];              // Z[64][j]
    float ecol[72];              // Z[i][64]
    float stage[32 * SST];       // redistribution staging; later rows16[16][66]; first: cost edge columns
};


__device__ __forceinline__ float lse_fin(float s, float mI) { return (fast_log2(s) + mI) * LN2; }
__device__ __forceinline__ float uni(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }

__device__ __forceinline__ void third_v2_problem(const Fused65Args& g, const int64_t p, BlkLds& lds, const int lane) {
    const int I = lane >> 3, J = lane & 7;
    const int colj = 8 * J + I;              // the column this lane owns in the column half-sweep
    // ---- marginals of log_optimal_transport2 (modules.py:169-179) --------------------------------
    // Wave-uniform quantities are pinned to SGPRs (uni): the kernel sits at the 168-VGPR budget of
    // three waves per SIMD, and every uniform value left in a VGPR pushes a spill reload into the
    // sweep loop (tools/check_hot_loops.py keeps watch).
    const float ns_own = g.ns[p * 64 + colj];
    const float ns_sum = uni(wave_sum(g.ns[p * 64 + lane]));
    const float ms = uni(64.0f * (g.one ? *g.one : 1.0f));
    const float norm = uni(-logf(ms + ns_sum));
    const float lmu = norm, lmu64 = uni(logf(ns_sum) + norm);
    const float lnu = logf(ns_own) + norm, lnu64 = uni(logf(ms) + norm);

    float dual_r = 0.f, dual_c = 0.f, dual_r64 = 0.f, dual_c64 = 0.f;   // u_(8I+J), v_(8J+I), u_64, v_64
    float kb[8][8];                          // the block: Z, then K (linear) - or Z kept (log path)
    float zdrow = 0.f, zdcol = 0.f, zcorner = 0.f;
    float sc_r = 0.f, sc_c = 0.f, sc_c64 = 0.f;          // a_(8I+J), b_(8J+I), a_64, b_64
    bool linear_done = false;

#pragma unroll
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool use_log = (attempt == 1) || !g.linear;
        // ---- cost build (MFMA) and redistribution: fragment layout -> 8x8 blocks ----------------
        {
            Cost65Acc c;
            cost65_accumulate(g.d0 + p * (int64_t)g.D * 65, g.d1 + p * (int64_t)g.D * 65, g.D, lds.stage, lane, c);
            const int li = lane & 31, lk = lane >> 5;
            const Cost65Scale sq(g.D);
#pragma unroll
            for (int tile = 0; tile < 4; ++tile) {
                const int ti = tile >> 1, tj = tile & 1;
                const f32x16& acc = tile == 0 ? c.c00 : tile == 1 ? c.c01 : tile == 2 ? c.c10 : c.c11;
                wg_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    lds.stage[((r & 3) + 8 * (r >> 2) + 4 * lk) * SST + li] = cost65_scale(acc[r], sq);
                wg_barrier();
                // element (rc, li) of this tile is matrix (2 rc + ti, 2 li + tj): this lane's block needs
                // rc = 4I + k, li = 4J + m  ->  local (2k + ti, 2m + tj)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f4v v = *reinterpret_cast<const f4v*>(&lds.stage[(4 * I + k) * SST + 4 * J]);
                    kb[2 * k + ti][0 + tj] = v.x;
                    kb[2 * k + ti][2 + tj] = v.y;
                    kb[2 * k + ti][4 + tj] = v.z;
                    kb[2 * k + ti][6 + tj] = v.w;
                }
            }
            if (lk == 0) {
                lds.erow[2 * li] = cost65_scale(c.er0, sq);
                lds.erow[2 * li + 1] = cost65_scale(c.er1, sq);
                lds.ecol[2 * li] = cost65_scale(c.ec0, sq);
                lds.ecol[2 * li + 1] = cost65_scale(c.ec1, sq);
            }
            zcorner = cost65_scale(c.cn, sq);
            wg_barrier();
            zdrow = lds.erow[colj];          // Z[64][8J+I]
            zdcol = lds.ecol[lane];          // Z[8I+J][64]
        }
        if (g.iters == 0) { linear_done = false; break; }      // u = v = 0

        if (!use_log) {
            // ---- stabilisers r_i = max_j Z_ij, c_j = max_i (Z_ij - r_i); K = exp(Z - r - c) ------
            float part[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float m = kb[r][0];
#pragma unroll
                for (int c = 1; c < 8; ++c) m = fmaxf(m, kb[r][c]);
                part[r] = m;
            }
            const float r_own = fmaxf(reduce8_consecutive(part, OpMax(), lane), zdcol);
            const float r64 = uni(fmaxf(wave_max(zdrow), zcorner));
            wg_barrier();
            lds.va[lane] = r_own;
            wg_barrier();
            float rloc[8];
            {
                const f4v v0 = *reinterpret_cast<const f4v*>(&lds.va[8 * I]), v1 = *reinterpret_cast<const f4v*>(&lds.va[8 * I + 4]);
                rloc[0] = v0.x; rloc[1] = v0.y; rloc[2] = v0.z; rloc[3] = v0.w;
                rloc[4] = v1.x; rloc[5] = v1.y; rloc[6] = v1.z; rloc[7] = v1.w;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float m = kb[0][c] - rloc[0];
#pragma unroll
                for (int r = 1; r < 8; ++r) m = fmaxf(m, kb[r][c] - rloc[r]);
                part[c] = m;
            }
            const float c_own = fmaxf(reduce8_strided(part, OpMax(), lane), zdrow - r64);
            const float c64 = uni(fmaxf(wave_max(zdcol - r_own), zcorner - r64));
            wg_barrier();
            lds.vb[colj] = c_own;
            wg_barrier();
            float cloc[8];
            {
                const f4v v0 = *reinterpret_cast<const f4v*>(&lds.vb[8 * J]), v1 = *reinterpret_cast<const f4v*>(&lds.vb[8 * J + 4]);
                cloc[0] = v0.x; cloc[1] = v0.y; cloc[2] = v0.z; cloc[3] = v0.w;
                cloc[4] = v1.x; cloc[5] = v1.y; cloc[6] = v1.z; cloc[7] = v1.w;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) kb[r][c] = fast_exp2(((kb[r][c] - rloc[r]) - cloc[c]) * LOG2E);
            const float kdcol = fast_exp2(((zdcol - r_own) - c64) * LOG2E);     // K[8I+J][64]
            const float kdrow = fast_exp2(((zdrow - r64) - c_own) * LOG2E);     // K[64][8J+I]
            const float kcorner = fast_exp2(((zcorner - r64) - c64) * LOG2E);
            const float mu = uni(expf(lmu)), mu64 = uni(expf(lmu64)), nu = expf(lnu), nu64 = uni(expf(lnu64));
            float a = 0.f, a64 = 0.f, b = expf(c_own), b64 = expf(c64);
            wg_barrier();
            lds.vb[colj] = b;
            for (int it = 0; it < g.iters; ++it) {
                wg_barrier();                                 // b visible
                {   // a_i = mu_i / sum_j K_ij b_j
                    const f4v b0 = *reinterpret_cast<const f4v*>(&lds.vb[8 * J]), b1 = *reinterpret_cast<const f4v*>(&lds.vb[8 * J + 4]);
                    // eight independent accumulator pairs advance together (column pair outer, row
                    // inner): a row-by-row order compiles to one serial chain with a wait state
                    // between every dependent v_pk_fma_f32
                    f2v acc[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[r] = f2v{kb[r][0], kb[r][1]} * b0.xy;
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[r] = __builtin_elementwise_fma(f2v{kb[r][2], kb[r][3]}, b0.zw, acc[r]);
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[r] = __builtin_elementwise_fma(f2v{kb[r][4], kb[r][5]}, b1.xy, acc[r]);
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[r] = __builtin_elementwise_fma(f2v{kb[r][6], kb[r][7]}, b1.zw, acc[r]);
                    // row sums: transpose the 8 partials of each row through LDS (the LDS pipe idles
                    // here while the VALU is the bound): lane (I,J) posts its partial of row 8I+r at
                    // [I][r][J] (stride 72: bank = 8I+8r+J, conflict-free) and collects row 8I+J's
                    // eight as two b128 reads - odd I takes the halves in swapped order so that a
                    // 16-lane read group touches every bank once.  12 VALU slots against the 29 of
                    // x+y adds plus the DPP butterfly.
                    // (measured: transposing the row partials through LDS instead of this DPP butterfly
                    // saves 17 VALU slots per sweep but lengthens the wave's LDS latency chain - 3% slower)
#pragma unroll
                    for (int r = 0; r < 8; ++r) part[r] = acc[r].x + acc[r].y;
                    const float dsum = wave_sum_uniform(kdrow * b);
                    const float s = fmaf(kdcol, b64, reduce8_consecutive(part, OpSum(), lane));
                    a = mu * __builtin_amdgcn_rcpf(s);
                    a64 = mu64 * __builtin_amdgcn_rcpf(fmaf(kcorner, b64, dsum));
                    lds.va[lane] = a;
                }
                wg_barrier();                                 // a visible
                {   // b_j = nu_j / sum_i K_ij a_i
                    const f4v a0 = *reinterpret_cast<const f4v*>(&lds.va[8 * I]), a1 = *reinterpret_cast<const f4v*>(&lds.va[8 * I + 4]);
                    const float al[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    f2v q01 = {0.f, 0.f}, q23 = {0.f, 0.f}, q45 = {0.f, 0.f}, q67 = {0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const f2v ar = {al[r], al[r]};
                        q01 = __builtin_elementwise_fma(f2v{kb[r][0], kb[r][1]}, ar, q01);
                        q23 = __builtin_elementwise_fma(f2v{kb[r][2], kb[r][3]}, ar, q23);
                        q45 = __builtin_elementwise_fma(f2v{kb[r][4], kb[r][5]}, ar, q45);
                        q67 = __builtin_elementwise_fma(f2v{kb[r][6], kb[r][7]}, ar, q67);
                    }
                    const float qq[8] = {q01.x, q01.y, q23.x, q23.y, q45.x, q45.y, q67.x, q67.y};
                    const float t = fmaf(kdrow, a64, reduce8_strided(qq, OpSum(), lane));
                    b = nu * __builtin_amdgcn_rcpf(t);
                    b64 = nu64 * __builtin_amdgcn_rcpf(fmaf(kcorner, a64, wave_sum_uniform(kdcol * a)));
                    lds.vb[colj] = b;
                }
            }
            if (__all(sc_ok(a) && sc_ok(b)) && sc_ok(a64) && sc_ok(b64)) {
                sc_r = a; sc_c = b; sc_c64 = b64;
                // u = log a - r, v = log b - c; the plan only needs the products below
                dual_r = logf(a) - r_own; dual_c = logf(b) - c_own;
                dual_r64 = logf(a64) - r64; dual_c64 = logf(b64) - c64;
                (void)dual_r; (void)dual_c; (void)dual_r64; (void)dual_c64;
                zdcol = kdcol;                  // reuse: K[8I+J][64]
                linear_done = true;
                break;
            }
            if (lane == 0 && g.fallbacks) atomicAdd(g.fallbacks, 1ull);
            continue;                           // guard tripped: redo with log-sum-exp sweeps
        }

        // ---- max-subtracted log-sum-exp sweeps in the block layout (rare path) --------------------
        float u = 0.f, v = 0.f, u64 = 0.f, v64 = 0.f;
        for (int it = 0; it < g.iters; ++it) {
            float part[8], loc[8], mloc[8];
            // u = log_mu - lse_j(Z + v)
            wg_barrier();
            lds.vb[colj] = v;
            wg_barrier();
#pragma unroll
            for (int c = 0; c < 8; ++c) loc[c] = lds.vb[8 * J + c];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float m = kb[r][0] + loc[0];
#pragma unroll
                for (int c = 1; c < 8; ++c) m = fmaxf(m, kb[r][c] + loc[c]);
                part[r] = m;
            }
            float m = fmaxf(reduce8_consecutive(part, OpMax(), lane), zdcol + v64);
            if (m == -INFINITY || m == INFINITY) m = 0.f;
            const float mI = ceilf(m * LOG2E);
            wg_barrier();
            lds.va[lane] = mI;
            wg_barrier();
#pragma unroll
            for (int r = 0; r < 8; ++r) mloc[r] = lds.va[8 * I + r];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float sacc = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) sacc += fast_exp2(fmaf(kb[r][c] + loc[c], LOG2E, -mloc[r]));
                part[r] = sacc;
            }
            const float s = reduce8_consecutive(part, OpSum(), lane) + fast_exp2(fmaf(zdcol + v64, LOG2E, -mI));
            u = lmu - lse_fin(s, mI);
            {
                const float t = zdrow + v, tc = zcorner + v64;           // dustbin row: one element per lane
                float m2 = fmaxf(wave_max(t), tc);
                if (m2 == -INFINITY || m2 == INFINITY) m2 = 0.f;
                const float mI2 = ceilf(m2 * LOG2E);
                const float s2 = wave_sum(fast_exp2(fmaf(t, LOG2E, -mI2))) + fast_exp2(fmaf(tc, LOG2E, -mI2));
                u64 = lmu64 - lse_fin(s2, mI2);
            }
            // v = log_nu - lse_i(Z + u)
            wg_barrier();
            lds.va[lane] = u;
            wg_barrier();
#pragma unroll
            for (int r = 0; r < 8; ++r) loc[r] = lds.va[8 * I + r];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float mm = kb[0][c] + loc[0];
#pragma unroll
                for (int r = 1; r < 8; ++r) mm = fmaxf(mm, kb[r][c] + loc[r]);
                part[c] = mm;
            }
            float mc = fmaxf(reduce8_strided(part, OpMax(), lane), zdrow + u64);
            if (mc == -INFINITY || mc == INFINITY) mc = 0.f;
            const float mIc = ceilf(mc * LOG2E);
            wg_barrier();
            lds.vb[colj] = mIc;
            wg_barrier();
#pragma unroll
            for (int c = 0; c < 8; ++c) mloc[c] = lds.vb[8 * J + c];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float sacc = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) sacc += fast_exp2(fmaf(kb[r][c] + loc[r], LOG2E, -mloc[c]));
                part[c] = sacc;
            }
            const float sc = reduce8_strided(part, OpSum(), lane) + fast_exp2(fmaf(zdrow + u64, LOG2E, -mIc));
            v = lnu - lse_fin(sc, mIc);
            {
                const float t = zdcol + u, tc = zcorner + u64;
                float m2 = fmaxf(wave_max(t), tc);
                if (m2 == -INFINITY || m2 == INFINITY) m2 = 0.f;
                const float mI2 = ceilf(m2 * LOG2E);
                const float s2 = wave_sum(fast_exp2(fmaf(t, LOG2E, -mI2))) + fast_exp2(fmaf(tc, LOG2E, -mI2));
                v64 = lnu64 - lse_fin(s2, mI2);
            }
        }
        dual_r = u; dual_c = v; dual_r64 = u64; dual_c64 = v64;
        break;
    }

    // ---- the 16 centre rows of the plan exp(Z + u + v - norm) -> LDS, then Compute_result --------
    // centre row q = 4 (qy - 2) + (qx - 2) is matrix row 8 qy + qx: block row I = qy, local row qx.
    wg_barrier();
    if (linear_done) { lds.va[lane] = sc_r; lds.vb[colj] = sc_c; } else { lds.va[lane] = dual_r; lds.vb[colj] = dual_c; }
    wg_barrier();
    float* rows16 = lds.stage;              // [16][66]
    if (I >= 2 && I <= 5) {
        float cl[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) cl[c] = lds.vb[8 * J + c];
#pragma unroll
        for (int rr = 2; rr <= 5; ++rr) {
            const int q = 4 * (I - 2) + (rr - 2);
            const float rv = lds.va[8 * I + rr];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float pv;
                if (linear_done) pv = ((kb[rr][c] * rv) * cl[c]) * expf(-norm);     // K a b e^-norm
                else pv = expf(((kb[rr][c] + rv) + cl[c]) - norm);                 // exp(Z + u + v - norm)
                rows16[q * 66 + 8 * J + c] = pv;
            }
        }
        if (J >= 2 && J <= 5) {             // this lane owns matrix row 8I+J = a centre row: its dustbin entry
            const int q = 4 * (I - 2) + (J - 2);
            rows16[q * 66 + 64] = linear_done ? ((zdcol * sc_r) * sc_c64) * expf(-norm)
                                              : expf(((zdcol + dual_r) + dual_c64) - norm);
        }
    }
    wg_barrier();
    // scale_x == NULL: scale_x = scale_y = sqrt(ns + 1e-8) (third_layer.py:153-154) is formed in the kernel
    const bool area = g.scale_x == nullptr;
    compute_result_problem(rows16, 0, p, (area ? g.ns : g.scale_x) + p * 64, (area ? g.ns : g.scale_y) + p * 64,
                           (float)g.p_s[p * 2], (float)g.p_s[p * 2 + 1], (float)g.p_t[p * 2], (float)g.p_t[p * 2 + 1],
                           g.outdoor, g.cr, lane, 66, area);
}

// direct mode: one workgroup per problem.  scan mode (g.scan): the W workgroups of the launch share the problems INTERLEAVED -
// workgroup w looks at w, w + W, w + 2 W, ... (64 candidates per ballot) and re-solves those whose if_matching1 slot carries
// THIRD_REDO (left there by third_fused3_kernel's guard).  Flagged problems come in runs (the points of one wild fine row are
// consecutive): with contiguous blocks of 64 per workgroup - round 3 - one wave re-solved a whole run serially while the rest
// of the GPU idled (9.1 ms per step for 8 000 flagged problems, profiles/r04_wild10_step_kernel_stats.md); interleaved, a
// run of 64 goes to 64 different waves.
__global__ void __launch_bounds__(64, 3)
third_fused_kernel(Fused65Args g) {
    __shared__ BlkLds lds;
    const int lane = threadIdx.x;
    if (!g.scan) {
        const int64_t p = blockIdx.x;
        if (p >= live_problems(g)) return;
        // de-phase the first wave-front (see sinkhorn65_kernel)
        if (g.stagger > 0 && blockIdx.x < 8192u) {
            const unsigned slots = (blockIdx.x * 2654435761u) >> 29;
            for (unsigned q = 0; q < slots * (unsigned)g.stagger; ++q) __builtin_amdgcn_s_sleep(127);
        }
        third_v2_problem(g, p, lds, lane);
        return;
    }
    const int64_t W = gridDim.x, live = live_problems(g);
    for (int64_t first = blockIdx.x; first < live; first += 64 * W) {
        const int64_t cand = first + (int64_t)lane * W;
        const bool redo = cand < live && g.cr.ifm[cand * 16] == THIRD_REDO;
        unsigned long long todo = __ballot(redo);
        while (todo) {
            const int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            wg_barrier();
            third_v2_problem(g, first + (int64_t)k * W, lds, lane);
        }
    }
}

int launch_third_fused3(const Fused65Args& g0, hipStream_t st);      // third_fused3.hip

int launch_third_fused(const Fused65Args& g0, hipStream_t st) {
    Fused65Args g = g0;
    g.linear = sinkhorn_mode() != PATS_SINKHORN_LOG;
    g.fallbacks = fallback_counter();
    static const bool v2_only = diag_env("PATS_THIRD_V2") != nullptr;     // A/B switch for benchmarking
    if (g.linear && g.iters > 0 && !v2_only) {
        // linear-domain solve by the third-generation kernel; it flags the problems that leave the guard
        // band, and this file's kernel re-solves exactly those with log-sum-exp sweeps (scan mode)
        int rc = launch_third_fused3(g, st);
        if (rc) return rc;
        g.linear = 0;
        g.scan = 1;
        const int64_t waves = g.P < 6144 ? g.P : 6144;            // two rounds of the 3 072 wave slots: flagged runs spread out
        hipLaunchKernelGGL(third_fused_kernel, dim3((unsigned)(waves > 0 ? waves : 1)), dim3(64), 0, st, g);
        return check_launch("third_fused_kernel(scan)");
    }
    if (g.P >= 8192) g.stagger = (int)((30.0f + 0.6f * (float)g.iters) / 16.0f / 3.4f);
    if (const char* e = diag_env("PATS_STAGGER")) g.stagger = atoi(e);
    hipLaunchKernelGGL(third_fused_kernel, dim3((unsigned)g.P), dim3(64), 0, st, g);
    return check_launch("third_fused_kernel");
}

}  // namespace pats
