// Log-domain Sinkhorn optimal transport for PATS on gfx950.
//
// Replaces models/modules.py:137-182 (log_sinkhorn_iterations, log_optimal_transport,
// log_optimal_transport2): the reference issues >= 6 ATen kernels per iteration (>= 600 launches
// per call); here one launch runs all iterations with the problem held on chip.
//
// Kernels
//   sinkhorn65_kernel   one WAVE per 65x65 problem (third level, [P,65,65], P ~ 1e4 per pair).
//                       Lane i keeps row i AND column i of Z in VGPRs (2 x 65 registers), so both
//                       half-sweeps reduce lane-locally; the other side's dual vector is broadcast
//                       through 260 B of LDS; the dustbin row/column (index 64) is the only
//                       cross-lane reduction (4 DPP + 2 permlane-swap steps).
//   sinkhorn_wg_kernel  one WORKGROUP per problem of any size: wave-per-row sweeps over Z and
//                       over a transposed copy Zt kept in the workspace (both sweeps coalesced,
//                       served by L2), duals in LDS.
//   ot_prep_kernel      marginals of modules.py:157-159 / :176-179 (norm, log_mu, log_nu).
//
// log-sum-exp:  lse(t) = (log2(sum_j 2^(t_j*log2e - mI)) + mI) * ln2  with the integer
// stabiliser mI = ceil(max_j t_j * log2e): max-subtracted like ATen's logsumexp, exact stabiliser
// arithmetic, one v_exp_f32 per element.
#include "common.hpp"
#include "third_device.hpp"
#include "cost65_device.hpp"
#include <stdlib.h>

namespace pats {

// ------------------------------------------------------------------------------------------
// marginals
// ------------------------------------------------------------------------------------------
// variant 1: log_optimal_transport   (scores [m,n] -> M=m+1, N=n+1; ns has n entries)
// variant 2: log_optimal_transport2  (scores [m,n] -> M=m,   N=n;   ns has n-1 entries)
__global__ void __launch_bounds__(256)
ot_prep_kernel(const float* __restrict__ ns, int ns_len, int M, int N, float ms_base,
               const float* __restrict__ one, float* __restrict__ log_mu,
               float* __restrict__ log_nu, float* __restrict__ norm_out) {
    const int b = blockIdx.x;
    const float* nsb = ns + (int64_t)b * ns_len;
    __shared__ float red[4];
    float acc = 0.f;
    for (int j = threadIdx.x; j < ns_len; j += 256) acc += nsb[j];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    wg_barrier();
    const float ns_sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float ms = ms_base * (one ? *one : 1.0f);       // modules.py:150 / :169
    const float norm = -logf(ms + ns_sum);                // modules.py:157 / :176
    for (int j = threadIdx.x; j < N; j += 256)
        log_nu[(int64_t)b * N + j] = (j < ns_len ? logf(nsb[j]) : logf(ms)) + norm;   // :158 / :178
    for (int i = threadIdx.x; i < M; i += 256)
        log_mu[(int64_t)b * M + i] = (i < M - 1 ? 0.f : logf(ns_sum)) + norm;         // :159 / :179
    if (threadIdx.x == 0) norm_out[b] = norm;
}

// ------------------------------------------------------------------------------------------
// Linear-domain ("kernel matrix") form of the same iteration
// ------------------------------------------------------------------------------------------
// With row stabilisers r_i = max_j Z_ij and column stabilisers c_j = max_i (Z_ij - r_i):
//     K_ij = exp(Z_ij - r_i - c_j)  in (0, 1], every row and every column contains a 1,
//     a_i = exp(u_i + r_i),  b_j = exp(v_j + c_j)
// the reference's sweep  u = log_mu - lse_j(Z + v);  v = log_nu - lse_i(Z + u)  is EXACTLY
//     a_i = mu_i / sum_j K_ij b_j ;   b_j = nu_j / sum_i K_ij a_i          (b starts at exp(c_j))
// i.e. the same fixed-point sequence (same iterate after every sweep), but one FMA per element per
// half-sweep instead of add/max/sub/exp/add.  The duals return to log space once at the end:
// u_i = log a_i - r_i, v_j = log b_j - c_j, Z_out = ((Z + u) + v) - norm from the ORIGINAL Z.
// Guard: entries of K below 2^-126 flush to zero; their true mass is < 2^-126 * a_i * b_j, so the
// result is unaffected as long as the scalings stay below 2^30.  Each problem checks
// max(a, b) <= 2^30 and finiteness at the end and otherwise re-runs itself with the max-subtracted
// log-sum-exp sweeps (the code below each linear block), which have no range restriction.
constexpr float SCALE_GUARD = 1073741824.0f;   // 2^30

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ bool scaling_ok(float x) { return x <= SCALE_GUARD && x > 0.f; }

__device__ __forceinline__ float lse_finish(float s, float mI) {
    return (fast_log2(s) + mI) * LN2;
}

// ------------------------------------------------------------------------------------------
// 65 x 65, one wave per problem
// ------------------------------------------------------------------------------------------
constexpr int NB = 64;          // real rows / columns
constexpr int NT = NB + 1;      // + dustbin
constexpr int TILE = NT * NT;   // 4225

struct __attribute__((aligned(16))) Wave65Lds {
    float bc0[NT + 3];          // broadcast buffers (duals / scalings / stabilisers), 16-B aligned
    float bc1[NT + 3];
    float tile[TILE + 3];       // the problem's Z, kept for the epilogue
};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// dot of a lane's register-resident K entries with a vector broadcast from LDS.  All LDS reads of a
// batch are issued before the first FMA (a lone wave otherwise pays the ~64-cycle LDS latency per
// read), and the products pair up as (k[4q],k[4q+1])*(b.x,b.y), (k[4q+2],k[4q+3])*(b.z,b.w) so the
// compiler emits v_pk_fma_f32 without register shuffles.
template <int LEN, int BATCH>
__device__ __forceinline__ float dot_bcast(const float* k, const float* bc) {
    static_assert(LEN % 4 == 0, "dot_bcast: LEN must be a multiple of 4");
    const f4* b4 = reinterpret_cast<const f4*>(bc);
    f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
    for (int q0 = 0; q0 < LEN / 4; q0 += BATCH) {
        f4 bb[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; ++q)
            if (q0 + q < LEN / 4) bb[q] = b4[q0 + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BATCH; ++q)
            if (q0 + q < LEN / 4) {
                const int e = 4 * (q0 + q);
                const f2 k0 = {k[e + 0], k[e + 1]}, k1 = {k[e + 2], k[e + 3]};
                acc0 = __builtin_elementwise_fma(k0, bb[q].xy, acc0);
                acc1 = __builtin_elementwise_fma(k1, bb[q].zw, acc1);
            }
        // without this the scheduler hoists the NEXT batch's reads above these FMAs and every
        // batch is live at once (LEN extra VGPRs -> spills in the 145-wide kernel)
        // (sched_barrier alone does not do it - instruction selection has already clustered the
        // loads; a compiler memory barrier keeps the next batch's ds_reads behind this point)
        // and the accumulators pass through it so this batch's FMAs cannot sink below it
        if (q0 + BATCH < LEN / 4) asm volatile("" : "+v"(acc0), "+v"(acc1) : : "memory");
    }
    const f2 sacc = acc0 + acc1;
    return sacc.x + sacc.y;
}
__device__ __forceinline__ float dot64(const float (&k)[64], const float* bc) {
    return dot_bcast<64, 16>(k, bc);
}

struct Ot65Args {
    const float* Zin;        // SRC 0: [P,65,65] couplings
    const float* d0;         // SRC 1: [P,D,65] descriptors
    const float* d1;
    int D;
    int64_t P;
    const float* log_mu;     // MODE 0
    const float* log_nu;
    const float* ns;         // MODE 2: [P,64]
    const float* one;
    int iters;
    float bias_k;
    int linear;
    float* out;              // [P,65,65] log-plan, may be null when EPI == 1
    // EPI 1: Compute_result fused (third_layer.py:159-170,184-217)
    const float* scale_x;
    const float* scale_y;
    const int64_t* p_s;
    const int64_t* p_t;
    int outdoor;
    ComputeResultOut cr;
    int stagger;             // first-wave-front start delay unit, in s_sleep(127) periods (0 = off)
    unsigned long long* fallbacks;   // guard-trip counter or null
};

// MODE 0: log_mu/log_nu given (a6)      MODE 2: ns given, log_optimal_transport2 marginals (a5)
// SRC  0: couplings from HBM            SRC  1: cost build from descriptors (a3 fused)
// EPI  0: write the log-plan            EPI  1: Compute_result + label from the LDS-resident plan
template <int MODE, int SRC, int EPI>
__global__ void __launch_bounds__(64, 2)     // 2 waves/SIMD: VGPR + AGPR (MFMA accumulators) <= 256
sinkhorn65_kernel(Ot65Args g) {
    __shared__ Wave65Lds lds;
    const int lane = threadIdx.x;
    const int64_t p = blockIdx.x;
    if (p >= g.P) return;
    const float* __restrict__ log_mu_in = g.log_mu;
    const float* __restrict__ log_nu_in = g.log_nu;
    const float* __restrict__ ns = g.ns;
    const float* __restrict__ one = g.one;
    const int iters = g.iters, linear = g.linear;
    const float bias_k = g.bias_k;

    // Every problem takes the same time, so without this all resident waves stay phase-locked: the
    // chip alternates between "everyone streams descriptors" (HBM-bound, VALU idle) and "everyone
    // sweeps" (VALU/LDS-bound, HBM idle) and the phase times ADD.  A hashed one-off start delay for
    // the first wave-front spreads the phases; later blocks inherit the spread from the slot they
    // take over.  Purely a scheduling aid - no effect on results.
    if (g.stagger > 0 && blockIdx.x < 4096u) {
        const unsigned slots = (blockIdx.x * 2654435761u) >> 29;         // 0..7
        for (unsigned q = 0; q < slots * (unsigned)g.stagger; ++q) __builtin_amdgcn_s_sleep(127);
    }
    if (SRC == 0) {
        // ---- coalesced load of the 4225 floats into LDS ----------------------------------------
        const float* Zp = g.Zin + p * TILE;
#pragma unroll 11
        for (int k = 0; k < 66; ++k) lds.tile[k * 64 + lane] = Zp[k * 64 + lane];
        if (lane == 0) lds.tile[TILE - 1] = Zp[TILE - 1];
    } else {
        cost65_to_tile(g.d0 + p * (int64_t)g.D * NT, g.d1 + p * (int64_t)g.D * NT, g.D, lds.tile, lane);
    }

    // ---- marginals -----------------------------------------------------------------------
    float lmu, lmu64, lnu, lnu64, norm = 0.f;
    if (MODE == 0) {
        lmu = log_mu_in[p * NT + lane];
        lmu64 = log_mu_in[p * NT + NB];
        lnu = log_nu_in[p * NT + lane];
        lnu64 = log_nu_in[p * NT + NB];
    } else {
        const float nsj = ns[p * NB + lane];
        const float ns_sum = wave_sum(nsj);
        const float ms = (float)NB * (one ? *one : 1.0f);
        norm = -logf(ms + ns_sum);
        lnu = logf(nsj) + norm;
        lnu64 = logf(ms) + norm;
        lmu = norm;
        lmu64 = logf(ns_sum) + norm;
    }
    wg_barrier();
    const float* T = lds.tile;
    const float corner_z = T[TILE - 1];

    float u = 0.f, u64 = 0.f, v = 0.f, v64 = 0.f;
    bool solved = (iters == 0);

    if (linear && !solved) {
        // ---- stabilisers: r_i = max_j Z_ij ; c_j = max_i (Z_ij - r_i) ---------------------------
        float r = T[lane * NT + NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) r = fmaxf(r, T[lane * NT + j]);      // stride 65: conflict-free
        const float r64 = fmaxf(wave_max(T[NB * NT + lane]), corner_z);
        lds.bc0[lane] = r;
        wg_barrier();
        float c = T[NB * NT + lane] - r64;
#pragma unroll
        for (int i = 0; i < NB; ++i) c = fmaxf(c, T[i * NT + lane] - lds.bc0[i]);
        const float c64 = fmaxf(wave_max(T[lane * NT + NB] - r), corner_z - r64);
        lds.bc1[lane] = c;
        wg_barrier();
        // ---- K in both orientations (identical op order => kr/kc hold bitwise-equal entries) ----
        float kr[NB], kc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) kr[j] = fast_exp2(((T[lane * NT + j] - r) - lds.bc1[j]) * LOG2E);
#pragma unroll
        for (int i = 0; i < NB; ++i) kc[i] = fast_exp2(((T[i * NT + lane] - lds.bc0[i]) - c) * LOG2E);
        const float kr64 = fast_exp2(((T[lane * NT + NB] - r) - c64) * LOG2E);     // K[lane][64]
        const float kc64 = fast_exp2(((T[NB * NT + lane] - r64) - c) * LOG2E);     // K[64][lane]
        const float kcorner = fast_exp2(((corner_z - r64) - c64) * LOG2E);
        const float mu = expf(lmu), mu64 = expf(lmu64), nu = expf(lnu), nu64 = expf(lnu64);
        float a = 0.f, a64 = 0.f, b = expf(c), b64 = expf(c64);
        for (int it = 0; it < iters; ++it) {
            wg_barrier();
            lds.bc1[lane] = b;
            wg_barrier();
            a = mu * fast_rcp(fmaf(kr64, b64, dot64(kr, lds.bc1)));
            a64 = mu64 * fast_rcp(fmaf(kcorner, b64, wave_sum(kc64 * b)));
            wg_barrier();
            lds.bc0[lane] = a;
            wg_barrier();
            b = nu * fast_rcp(fmaf(kc64, a64, dot64(kc, lds.bc0)));
            b64 = nu64 * fast_rcp(fmaf(kcorner, a64, wave_sum(kr64 * a)));
        }
        // guard: every scaling finite, positive and <= 2^30 (comparisons are false for NaN)
        const bool ok_lane = scaling_ok(a) && scaling_ok(b);
        if (__all(ok_lane) && scaling_ok(a64) && scaling_ok(b64)) {
            u = logf(a) - r;
            u64 = logf(a64) - r64;
            v = logf(b) - c;
            v64 = logf(b64) - c64;
            solved = true;
        }
    }

    if (!solved) {
        if (g.linear && lane == 0 && g.fallbacks) atomicAdd(g.fallbacks, 1ull);
        // ---- max-subtracted log-sum-exp sweeps, Z rows and columns in registers -------------------
        float zr[NB], zc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) zr[j] = T[lane * NT + j];
#pragma unroll
        for (int i = 0; i < NB; ++i) zc[i] = T[i * NT + lane];
        const float zr64 = T[lane * NT + NB];    // Z[lane][64]   (dustbin column)
        const float zc64 = T[NB * NT + lane];    // Z[64][lane]   (dustbin row)
        u = u64 = v = v64 = 0.f;
        for (int it = 0; it < iters; ++it) {
            // u = log_mu - lse_j(Z + v)
            wg_barrier();
            lds.bc1[lane] = v;
            wg_barrier();
            {
                float m = zr64 + v64;
#pragma unroll
                for (int j = 0; j < NB; ++j) m = fmaxf(m, zr[j] + lds.bc1[j]);
                const float mI = ceilf(m * LOG2E);
                float s = fast_exp2(fmaf(zr64 + v64, LOG2E, -mI));
#pragma unroll
                for (int j = 0; j < NB; ++j) s += fast_exp2(fmaf(zr[j] + lds.bc1[j], LOG2E, -mI));
                u = lmu - lse_finish(s, mI);
                const float t = zc64 + v, tc = corner_z + v64;     // dustbin row, one element per lane
                const float mI2 = ceilf(fmaxf(wave_max(t), tc) * LOG2E);
                const float s2 = wave_sum(fast_exp2(fmaf(t, LOG2E, -mI2))) + fast_exp2(fmaf(tc, LOG2E, -mI2));
                u64 = lmu64 - lse_finish(s2, mI2);
            }
            // v = log_nu - lse_i(Z + u)
            wg_barrier();
            lds.bc0[lane] = u;
            wg_barrier();
            {
                float m = zc64 + u64;
#pragma unroll
                for (int i = 0; i < NB; ++i) m = fmaxf(m, zc[i] + lds.bc0[i]);
                const float mI = ceilf(m * LOG2E);
                float s = fast_exp2(fmaf(zc64 + u64, LOG2E, -mI));
#pragma unroll
                for (int i = 0; i < NB; ++i) s += fast_exp2(fmaf(zc[i] + lds.bc0[i], LOG2E, -mI));
                v = lnu - lse_finish(s, mI);
                const float t = zr64 + u, tc = corner_z + u64;
                const float mI2 = ceilf(fmaxf(wave_max(t), tc) * LOG2E);
                const float s2 = wave_sum(fast_exp2(fmaf(t, LOG2E, -mI2))) + fast_exp2(fmaf(tc, LOG2E, -mI2));
                v64 = lnu64 - lse_finish(s2, mI2);
            }
        }
    }

    // ---- Z + u + v - norm (+ the caller's dustbin bias) from the original Z -----------------------
    wg_barrier();
    lds.bc0[lane] = u;
    wg_barrier();
    const float lb = bias_k > 0.f ? logf(bias_k) : 0.f;
    float* Tw = lds.tile;
    if (EPI == 0) {
        float* Op = g.out + p * TILE;
#pragma unroll 8
        for (int i = 0; i < NB; ++i) Op[i * NT + lane] = ((T[i * NT + lane] + lds.bc0[i]) + v) - norm;
        // dustbin column entries (i, 64) and dustbin row entries (64, j)
        float cc = ((T[lane * NT + NB] + u) + v64) - norm;
        float rr = ((T[NB * NT + lane] + u64) + v) - norm;
        if (bias_k > 0.f) { cc += lb; rr += lb; }
        Op[lane * NT + NB] = cc;
        Op[NB * NT + lane] = rr;
        if (lane == 0) {
            float q = ((corner_z + u64) + v64) - norm;
            if (bias_k > 0.f) { q += lb; q += lb; }     // corner receives both in-place adds
            Op[TILE - 1] = q;
        }
    } else {
        // the log-plan replaces Z in the LDS tile; Compute_result reads it there
#pragma unroll 8
        for (int i = 0; i < NB; ++i) Tw[i * NT + lane] = ((T[i * NT + lane] + lds.bc0[i]) + v) - norm;
        float cc = ((T[lane * NT + NB] + u) + v64) - norm;
        float rr = ((T[NB * NT + lane] + u64) + v) - norm;
        if (bias_k > 0.f) { cc += lb; rr += lb; }
        float q = ((corner_z + u64) + v64) - norm;
        if (bias_k > 0.f) { q += lb; q += lb; }
        wg_barrier();
        Tw[lane * NT + NB] = cc;
        Tw[NB * NT + lane] = rr;
        if (lane == 0) Tw[TILE - 1] = q;
        wg_barrier();
        if (g.out) {
            float* Op = g.out + p * TILE;
#pragma unroll 11
            for (int k = 0; k < 66; ++k) Op[k * 64 + lane] = Tw[k * 64 + lane];
            if (lane == 0) Op[TILE - 1] = Tw[TILE - 1];
        }
        compute_result_problem(Tw, 1, p, g.scale_x + p * 64, g.scale_y + p * 64,
                               (float)g.p_s[p * 2], (float)g.p_s[p * 2 + 1], (float)g.p_t[p * 2],
                               (float)g.p_t[p * 2 + 1], g.outdoor, g.cr, lane);
    }
}

// standalone cost build for 65-wide problems (pats_cost_f32 fast path): one wave per problem
template <bool F16>
__global__ void __launch_bounds__(64)
cost65_kernel(const float* __restrict__ d0, const float* __restrict__ d1, int D, int64_t P,
              float* __restrict__ out) {
    __shared__ float tile[TILE + 3];
    const int lane = threadIdx.x;
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    cost65_to_tile<F16>(d0 + p * (int64_t)D * NT, d1 + p * (int64_t)D * NT, D, tile, lane);
    wg_barrier();
    float* Op = out + p * TILE;
#pragma unroll 11
    for (int k = 0; k < 66; ++k) Op[k * 64 + lane] = tile[k * 64 + lane];
    if (lane == 0) Op[TILE - 1] = tile[TILE - 1];
}

int launch_cost65(const float* d0, const float* d1, int D, int64_t P, float* out, hipStream_t st) {
    static const bool f16 = diag_env("PATS_COST65_F16") != nullptr;      // diagnostic: the fp16-split contraction of the fused kernel
    if (f16) hipLaunchKernelGGL(cost65_kernel<true>, dim3((unsigned)P), dim3(64), 0, st, d0, d1, D, P, out);
    else hipLaunchKernelGGL(cost65_kernel<false>, dim3((unsigned)P), dim3(64), 0, st, d0, d1, D, P, out);
    return check_launch("cost65_kernel");
}

// ------------------------------------------------------------------------------------------
// N x N with N <= 192 (the fine level: 145 x 145), one 384-thread workgroup per problem
// ------------------------------------------------------------------------------------------
// Waves 0-2 are ROW waves (lane i keeps row i of K in N VGPRs), waves 3-5 are COLUMN waves (lane j
// keeps column j).  Each half-sweep is a lane-local dot product of those registers with the other
// side's scaling vector, broadcast from LDS in batches that are fully in flight before the FMAs
// start; one barrier hands a (or b) to the other wave group, so two barriers per sweep (the
// Gauss-Seidel order leaves no work for the idle group anyway).  Splitting the orientations keeps
// each thread at N + prefetch registers (< 256, two waves per SIMD) instead of 2N.  The problem's
// Z stays in LDS (N*N*4 B = 84 KB at N = 145) for the stabilisers and the epilogue: HBM sees Z
// once in and once out.
template <int N_>
struct __attribute__((aligned(16))) WgLds {
    float bc0[(N_ + 7) & ~3];       // r, then a (and u for the epilogue)
    float bc1[(N_ + 7) & ~3];       // c, then b (and v)
    float red[12];                  // six per-wave slots x 2: the stabilised re-solve's drift flags alternate halves by sweep parity
    float tile[N_ * N_];
};

template <int N_>
__device__ __forceinline__ float dotN(const float (&k)[(N_ + 3) & ~3], const float* bc) {
    return dot_bcast<(N_ + 3) & ~3, 19>(k, bc);      // k and bc are zero-padded to a multiple of 4
}

template <int N_, int MODE>
__global__ void __launch_bounds__(384)
sinkhorn_rc_kernel(const float* Zin, int64_t P, const float* __restrict__ log_mu_in,
                   const float* __restrict__ log_nu_in, const float* __restrict__ ns,
                   const float* __restrict__ one, int iters, float bias_k, int linear,
                   float* out, unsigned long long* fallbacks, const int* __restrict__ only_if,
                   const int64_t* __restrict__ live = nullptr) {
    // Zin and out are NOT __restrict__: the fused fine-level redo (launch_fine145_fused) solves in place, Zin == out.  The
    // whole matrix is staged in LDS before the first store and Zin is not read again.
    __shared__ __attribute__((aligned(16))) WgLds<N_> lds;     // 85 KB at N = 145 (static: no opt-in)
    if (live && (int64_t)blockIdx.x >= *live) return;          // counted launch: a padding row
    if (only_if) {                                  // re-solve pass after sinkhorn_blk145_kernel: flagged problems only
        if (only_if[blockIdx.x] == 0) return;
        if (threadIdx.x == 0 && fallbacks) atomicAdd(fallbacks, 1ull);
    }
    constexpr int TH = 384, NN = N_ * N_, NP = (N_ + 3) & ~3;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool colw = __builtin_amdgcn_readfirstlane(wave) >= 3;   // wave-uniform orientation
    const int tl = t - (colw ? 192 : 0);                           // row (or column) index of this lane
    const bool act = tl < N_;
    const int tt = act ? tl : 0;
    const int64_t p = blockIdx.x;
    const float* Zp = Zin + p * NN;
    for (int k = t; k < NN; k += TH) lds.tile[k] = Zp[k];
    if (t < 2 * ((N_ + 7) & ~3)) lds.bc0[t] = 0.f;                 // bc0 and bc1 are adjacent: zero the pads

    // ---- marginals: row waves need log_mu of their row, column waves log_nu of their column ----
    float lmarg, norm = 0.f;
    if (MODE == 0) {
        lmarg = colw ? log_nu_in[p * N_ + tt] : log_mu_in[p * N_ + tt];
    } else {
        const float nsj = (!colw && t < N_ - 1) ? ns[p * (N_ - 1) + t] : 0.f;
        const float part = wave_sum(nsj);
        if (lane == 0) lds.red[wave] = part;
        wg_barrier();
        const float ns_sum = (lds.red[0] + lds.red[1]) + lds.red[2];
        const float ms = (float)(N_ - 1) * (one ? *one : 1.0f);
        norm = -logf(ms + ns_sum);
        if (colw) lmarg = (tl < N_ - 1 ? logf(ns[p * (N_ - 1) + tt]) : logf(ms)) + norm;
        else lmarg = (tl < N_ - 1 ? 0.f : logf(ns_sum)) + norm;
    }
    wg_barrier();
    const float* T = lds.tile;
    float dual = 0.f;                   // u_i in row waves, v_j in column waves
    bool solved = (iters == 0);

    if (linear && !solved) {
        // stabilisers r_i = max_j Z_ij (row waves), then c_j = max_i (Z_ij - r_i) (column waves)
        float stab = -INFINITY;
        if (!colw) {
#pragma unroll 5
            for (int j = 0; j < N_; ++j) stab = fmaxf(stab, T[tt * N_ + j]);     // stride N_ (odd): conflict-free
            if (act) lds.bc0[tl] = stab;
        }
        wg_barrier();
        if (colw) {
#pragma unroll 5
            for (int i = 0; i < N_; ++i) stab = fmaxf(stab, T[i * N_ + tt] - lds.bc0[i]);
            if (act) lds.bc1[tl] = stab;
        }
        wg_barrier();
        float kk[NP];
        // built in chunks of 8 with a compiler barrier between them: a free-running full unroll
        // keeps hundreds of LDS reads in flight and the allocator answers by spilling K entries
        // for the whole kernel
        if (!colw) {
#pragma unroll
            for (int j0 = 0; j0 < N_; j0 += 8) {
#pragma unroll
                for (int j = j0; j < j0 + 8 && j < N_; ++j)
                    kk[j] = fast_exp2(((T[tt * N_ + j] - stab) - lds.bc1[j]) * LOG2E);
                asm volatile("" ::: "memory");
            }
        } else {
#pragma unroll
            for (int i0 = 0; i0 < N_; i0 += 8) {
#pragma unroll
                for (int i = i0; i < i0 + 8 && i < N_; ++i)
                    kk[i] = fast_exp2(((T[i * N_ + tt] - lds.bc0[i]) - stab) * LOG2E);
                asm volatile("" ::: "memory");
            }
        }
#pragma unroll
        for (int q = N_; q < NP; ++q) kk[q] = 0.f;
        const float marg = expf(lmarg);
        float sc = colw ? expf(stab) : 1.f;        // b starts at exp(c_j); a is computed first
        wg_barrier();                           // everyone is done reading the stabilisers
        if (colw && act) lds.bc1[tl] = sc;
        // linear == 2 (the re-solve behind sinkhorn_blk145_kernel): STABILISED linear sweeps.  A scaling that drifts out of
        // [2^-20, 2^20] is absorbed into its stabiliser - r_i <- r_i - ln a_i, c_j <- c_j - ln b_j, a = b = 1 - and K is rebuilt
        // from the scores in LDS with the new stabilisers (exact exponents, nothing that was flushed stays flushed): the same
        // iterate as the reference's log-sum-exp form for any score range a sweep does not carry past fp32 in one step, at the
        // linear kernel's speed.  Round 3 sent these problems through the log-domain sweeps below, which read Z from LDS four
        // times per element and sweep with one or two waves per SIMD: 13 us per problem, 26 ms for the 2 000 wild problems of a
        // step with 10 % of its rows scaled by 32 (profiles/r04_wild10_step_kernel_stats.md).
        const bool absorb = linear == 2;
        if (absorb && t < 12) lds.red[t] = 0.f;
        for (int it = 0; it < iters; ++it) {
            wg_barrier();                       // b visible (and the drift flags of the previous sweep)
            // The flags of sweep it live in half (it & 1): this sweep READS the other half (written during sweep it - 1, complete
            // behind the barrier above) and WRITES its own, which every wave first clears for itself - so a fast wave raising its
            // flag later in this sweep can never be seen by a slow wave still evaluating the branch below (round-4 advice: with one
            // set of flags the two could disagree, and the branch holds barriers).
            const float* frd = lds.red + ((it + 1) & 1) * 6;
            float* fwr = lds.red + (it & 1) * 6;
            const bool drifted = absorb && it > 0 && ((frd[0] + frd[1] + frd[2]) + (frd[3] + frd[4] + frd[5])) > 0.5f;
            if (absorb && lane == 0) fwr[wave] = 0.f;
            if (drifted) {
                // workgroup-uniform branch: every wave re-bases its stabilisers and rebuilds its part of K
                const float lg = logf(sc);
                if (act && lg == lg && fabsf(lg) < 3.0e38f) { stab -= lg; sc = 1.f; }     // a dead scaling (0, inf, NaN) is left to the guard
                wg_barrier();                   // every wave has read the old b
                if (act) (colw ? lds.bc1 : lds.bc0)[tl] = stab;
                wg_barrier();
                if (!colw) {
#pragma unroll
                    for (int j0 = 0; j0 < N_; j0 += 8) {
#pragma unroll
                        for (int j = j0; j < j0 + 8 && j < N_; ++j)
                            kk[j] = fast_exp2(((T[tt * N_ + j] - stab) - lds.bc1[j]) * LOG2E);
                        asm volatile("" ::: "memory");
                    }
                } else {
#pragma unroll
                    for (int i0 = 0; i0 < N_; i0 += 8) {
#pragma unroll
                        for (int i = i0; i < i0 + 8 && i < N_; ++i)
                            kk[i] = fast_exp2(((T[i * N_ + tt] - lds.bc0[i]) - stab) * LOG2E);
                        asm volatile("" ::: "memory");
                    }
                }
                wg_barrier();                   // everyone is done reading the stabilisers
                if (colw && act) lds.bc1[tl] = sc;
                wg_barrier();                   // b visible
            }
            if (!colw) {
                sc = marg * fast_rcp(dotN<N_>(kk, lds.bc1));
                if (act) lds.bc0[tl] = sc;
                if (absorb && __any(act && !(sc <= 1048576.f && sc >= 9.5367431640625e-07f)) && lane == 0) fwr[wave] = 1.f;
            }
            wg_barrier();                       // a visible
            if (colw) {
                sc = marg * fast_rcp(dotN<N_>(kk, lds.bc0));
                if (act) lds.bc1[tl] = sc;
                if (absorb && __any(act && !(sc <= 1048576.f && sc >= 9.5367431640625e-07f)) && lane == 0) fwr[wave] = 1.f;
            }
        }
        const bool ok_wave = __all(!act || scaling_ok(sc));
        wg_barrier();
        if (lane == 0) lds.red[wave] = ok_wave ? 1.f : 0.f;
        wg_barrier();
        const float okp = (lds.red[0] * lds.red[1] * lds.red[2]) * (lds.red[3] * lds.red[4] * lds.red[5]);
        if (okp > 0.5f) {
            dual = logf(sc) - stab;
            solved = true;
        }
        wg_barrier();
    }

    if (!solved) {      // workgroup-uniform: max-subtracted log-sum-exp sweeps on Z itself.
        if (linear == 1 && t == 0 && fallbacks) atomicAdd(fallbacks, 1ull);      // (a re-solve pass has counted its problem on entry)
        // Rare path (guard tripped, or PATS_SINKHORN_LOG): Z is read from the LDS tile each time
        // rather than held in registers, so it does not raise the kernel's VGPR budget.
        const int zstride = colw ? N_ : 1;
        const float* zbase = colw ? T + tt : T + tt * N_;
        dual = 0.f;
        if (act) { lds.bc0[tl] = 0.f; lds.bc1[tl] = 0.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {          // h = 0: u from v (row waves); h = 1: v from u
                wg_barrier();
                if ((h == 1) == colw) {
                    const float* other = h == 0 ? lds.bc1 : lds.bc0;
                    float m = -INFINITY;
#pragma unroll 8
                    for (int q = 0; q < N_; ++q) m = fmaxf(m, zbase[q * zstride] + other[q]);
                    if (m == -INFINITY || m == INFINITY) m = 0.f;
                    const float mI = ceilf(m * LOG2E);
                    float sacc = 0.f;
#pragma unroll 8
                    for (int q = 0; q < N_; ++q)
                        sacc += fast_exp2(fmaf(zbase[q * zstride] + other[q], LOG2E, -mI));
                    dual = lmarg - lse_finish(sacc, mI);
                    if (act) (h == 0 ? lds.bc0 : lds.bc1)[tl] = dual;
                }
            }
        }
    }

    // ---- epilogue: ((Z + u) + v) - norm (+ bias) from the LDS copy of Z, coalesced ---------------
    wg_barrier();
    if (act) (colw ? lds.bc1 : lds.bc0)[tl] = dual;
    wg_barrier();
    const float lb = bias_k > 0.f ? logf(bias_k) : 0.f;
    float* Op = out + p * NN;
    for (int k = t; k < NN; k += TH) {
        const int i = k / N_, j = k - i * N_;
        float z = ((T[k] + lds.bc0[i]) + lds.bc1[j]) - norm;
        if (bias_k > 0.f) {
            if (j == N_ - 1) z += lb;
            if (i == N_ - 1) z += lb;
        }
        Op[k] = z;
    }
}

// ------------------------------------------------------------------------------------------
// any size, one workgroup per problem
// ------------------------------------------------------------------------------------------
// Virtual source matrix: plain [M,N], or the dustbin-augmented couplings of modules.py:152-156
// built on the fly from scores [M-1,N-1] + alpha.
struct SrcView {
    const float* base;     // per problem stride below
    int64_t stride;        // elements between problems
    int ld;                // leading dimension of the stored matrix
    int rows, cols;        // stored rows / cols (M-1,N-1 when augmenting)
    const float* alpha;    // non-null => augment
};
__device__ __forceinline__ float src_at(const SrcView& s, const float* b, int i, int j) {
    if (s.alpha && (i >= s.rows || j >= s.cols)) return *s.alpha;
    return b[(int64_t)i * s.ld + j];
}

// one wave reduces lse_k(row[k] + add[k]) over `len` contiguous elements
__device__ __forceinline__ float wave_lse(const float* __restrict__ row, const float* add, int len,
                                          int lane) {
    float m = -INFINITY;
    for (int k = lane; k < len; k += 64) m = fmaxf(m, row[k] + add[k]);
    m = wave_max(m);
    if (m == -INFINITY || m == INFINITY) m = 0.f;      // ATen: maxes.masked_fill_(|max| == inf, 0)
    const float mI = ceilf(m * LOG2E);
    float s = 0.f;
    for (int k = lane; k < len; k += 64) s += fast_exp2(fmaf(row[k] + add[k], LOG2E, -mI));
    s = wave_sum(s);
    return lse_finish(s, mI);
}

__global__ void sinkhorn_wg_kernel(SrcView src, int M, int N, const float* __restrict__ log_mu,
                                   const float* __restrict__ log_nu,
                                   const float* __restrict__ norm_in, int iters, float bias_k,
                                   float* __restrict__ out, float* __restrict__ wsZ,
                                   float* __restrict__ wsT, const int* __restrict__ only_if,
                                   unsigned long long* fallbacks) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if (only_if && only_if[blockIdx.x] == 0) return;     // re-solve pass: only flagged problems
    if (only_if && fallbacks && threadIdx.x == 0) atomicAdd(fallbacks, 1ull);
    float* u = sm;          // [M]
    float* v = sm + M;      // [N]
    const int b = blockIdx.x;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    const float* sb = src.base + (int64_t)b * src.stride;
    const int64_t MN = (int64_t)M * N;
    float* Zt = wsT + (int64_t)b * MN;
    float* Zw = wsZ ? wsZ + (int64_t)b * MN : nullptr;
    const float* Zr = Zw ? Zw : sb;     // row-major image the sweeps read (ld == N in both cases)

    // materialise the row-major image (only when augmenting) and the transposed image
    for (int64_t idx = tid; idx < MN; idx += nthr) {
        const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
        const float z = src_at(src, sb, i, j);
        if (Zw) Zw[idx] = z;
        Zt[(int64_t)j * M + i] = z;
    }
    for (int i = tid; i < M; i += nthr) u[i] = 0.f;
    for (int j = tid; j < N; j += nthr) v[j] = 0.f;
    wg_barrier_global();     // Zw / Zt went to GLOBAL memory from all waves and are read back wave-per-row below

    const float* lmu = log_mu + (int64_t)b * M;
    const float* lnu = log_nu + (int64_t)b * N;
    for (int it = 0; it < iters; ++it) {
        for (int i = wave; i < M; i += nwave) {
            const float l = wave_lse(Zr + (int64_t)i * N, v, N, lane);
            if (lane == 0) u[i] = lmu[i] - l;
        }
        wg_barrier();
        for (int j = wave; j < N; j += nwave) {
            const float l = wave_lse(Zt + (int64_t)j * M, u, M, lane);
            if (lane == 0) v[j] = lnu[j] - l;
        }
        wg_barrier();
    }
    const float norm = norm_in ? norm_in[b] : 0.f;
    const float lb = bias_k > 0.f ? logf(bias_k) : 0.f;
    float* ob = out + (int64_t)b * MN;
    for (int64_t idx = tid; idx < MN; idx += nthr) {
        const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
        float z = (Zr[idx] + u[i]) + v[j];
        if (norm_in) z = z - norm;
        if (bias_k > 0.f) {
            if (j == N - 1) z += lb;
            if (i == M - 1) z += lb;
        }
        ob[idx] = z;
    }
}

// ------------------------------------------------------------------------------------------
// up to 304 x 320 (the coarse level: 301 x 301), one 1024-thread workgroup = one CU per problem
// ------------------------------------------------------------------------------------------
// The whole kernel matrix lives in the register files of one CU: wave w owns rows w, w+16, ...
// (RPW = 19 of them), lane l owns columns l, l+64, ... (CPL = 5), so a lane holds a 19 x 5 block
// (95 VGPRs, 4 waves per SIMD).  Column scalings b_j are lane-local registers; row scalings a_i
// are wave-uniform (SGPR) values.  Row pass: 95 FMAs per lane, then ONE transposed all-reduce of
// the 19 partial sums (v_permlane32_swap pairs rows across half-waves, v_permlane16_swap across
// 16-lane rows, then 4 DPP steps: ~50 instructions instead of 19 separate 6-step reductions).
// Column pass: 95 FMAs per lane into 5 partial column sums, combined across the 16 waves through a
// double-buffered 20 KB LDS array (one barrier per sweep).  Z is re-read from L2 for the epilogue.
// A problem whose scalings leave the guard sets fail[b]; sinkhorn_wg_kernel re-solves only those.
typedef float f2p __attribute__((ext_vector_type(2)));

template <int RPW, class Op>
struct RowReduce {
    static constexpr int N1 = (RPW + 1) / 2, N2 = (N1 + 1) / 2;
    // slot s (row wave + 16 s) lives in red[slot_reg(s)], 16-lane group slot_grp(s)
    __host__ __device__ static constexpr int slot_reg(int s) { return (s < N1 ? s : s - N1) % N2; }
    __host__ __device__ static constexpr int slot_grp(int s) {
        return (s < N1 ? 0 : 2) + (((s < N1 ? s : s - N1) >= N2) ? 1 : 0);
    }
    // slot held by register i in 16-lane group g (may be >= RPW: padding)
    __device__ static int slot_of(int i, int g) { return i + N2 * (g & 1) + N1 * (g >> 1); }

    // pf(s) yields this lane's partial for slot s; it is evaluated right before the swap that
    // consumes it so the RPW partials are never all live at once
    template <class PF>
    __device__ static void run(PF pf, float (&red)[N2], Op op, float identity) {
        float a1[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            unsigned x = __builtin_bit_cast(unsigned, pf(i));
            unsigned y = __builtin_bit_cast(unsigned, (i + N1 < RPW) ? pf(i + N1 < RPW ? i + N1 : 0) : identity);
            lane_swap32(x, y);
            a1[i] = op(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
        }
#pragma unroll
        for (int i = 0; i < N2; ++i) {
            unsigned x = __builtin_bit_cast(unsigned, a1[i]);
            unsigned y = __builtin_bit_cast(unsigned, (i + N2 < N1) ? a1[i + N2] : identity);
            lane_swap16(x, y);
            float v = op(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
            v = op(v, dpp_f<DPP_QUAD_XOR1>(v));
            v = op(v, dpp_f<DPP_QUAD_XOR2>(v));
            v = op(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
            v = op(v, dpp_f<DPP_ROW_MIRROR>(v));
            red[i] = v;
        }
    }

    // the same reduction fed PAIR-wise: pf2(i) yields this lane's partials of slots i and i + N1 together (one packed FMA chain
    // over a register pair, sinkhorn_cu2_kernel) - exactly the two values the first exchange level consumes side by side
    template <class PF2>
    __device__ static void run_pairs(PF2 pf2, float (&red)[N2], Op op, float identity) {
        float a1[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            const f2p pr = pf2(i);
            unsigned x = __builtin_bit_cast(unsigned, pr.x);
            unsigned y = __builtin_bit_cast(unsigned, (i + N1 < RPW) ? pr.y : identity);
            lane_swap32(x, y);
            a1[i] = op(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
        }
#pragma unroll
        for (int i = 0; i < N2; ++i) {
            unsigned x = __builtin_bit_cast(unsigned, a1[i]);
            unsigned y = __builtin_bit_cast(unsigned, (i + N2 < N1) ? a1[i + N2] : identity);
            lane_swap16(x, y);
            float v = op(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
            v = op(v, dpp_f<DPP_QUAD_XOR1>(v));
            v = op(v, dpp_f<DPP_QUAD_XOR2>(v));
            v = op(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
            v = op(v, dpp_f<DPP_ROW_MIRROR>(v));
            red[i] = v;
        }
    }
};

template <int RPW, int CPL, int LSLOTS>
__global__ void __launch_bounds__(1024)
sinkhorn_cu_kernel(SrcView src, int M, int N, const float* __restrict__ log_mu,
                   const float* __restrict__ log_nu, const float* __restrict__ norm_in, int iters,
                   float* __restrict__ out, int* __restrict__ fail) {
    constexpr int NW = 16, CW = CPL * 64, RREG = RPW - LSLOTS;
    using RSum = RowReduce<RPW, OpSum>;
    using RMax = RowReduce<RPW, OpMax>;
    constexpr int N2 = RSum::N2;
    __shared__ float part[2][NW][CW];       // cross-wave column partials (double-buffered)
    __shared__ float kl[LSLOTS][NW][CW];    // the last LSLOTS row-slots of K (register budget: 128)
    __shared__ float rsave[NW][RPW + 1];    // row stabilisers r_i per (wave, slot)
    __shared__ float csave[CW];             // column stabilisers
    __shared__ int ok_s[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4;
    const int b = blockIdx.x;
    const float* sb = src.base + (int64_t)b * src.stride;
    const float* lmu = log_mu + (int64_t)b * M;
    const float* lnu = log_nu + (int64_t)b * N;

    float K[RREG][CPL];
#define KREF(s_, c) (*((s_) < RREG ? &K[(s_) < RREG ? (s_) : 0][c] : &kl[(s_) >= RREG ? (s_) - RREG : 0][wave][lane + 64 * (c)]))
    bool cval[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) cval[c] = lane + 64 * c < N;

    // ---- load Z block (rows wave + 16 s, columns lane + 64 c); -inf outside the matrix ----------
#pragma unroll
    for (int s_ = 0; s_ < RPW; ++s_) {
        const int i = wave + NW * s_;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            KREF(s_, c) = (i < M && cval[c]) ? src_at(src, sb, i, lane + 64 * c) : -INFINITY;
    }
    // ---- r_i = max_j Z_ij (transposed wave reduction) ------------------------------------------
    {
        float rred[N2];
        RMax::run([&](int s_) {
            float m = KREF(s_, 0);
#pragma unroll
            for (int c = 1; c < CPL; ++c) m = fmaxf(m, KREF(s_, c));
            return m; }, rred, OpMax(), -INFINITY);
        // keep one scalar per row slot: slot s sits in register slot_reg(s), 16-lane group slot_grp(s)
#pragma unroll
        for (int i = 0; i < N2; ++i)
            if ((lane & 15) == 0 && RSum::slot_of(i, grp) < RPW) rsave[wave][RSum::slot_of(i, grp)] = rred[i];
    }
    // ---- c_j = max_i (Z_ij - r_i): per-wave partial, then across waves through LDS ---------------
    // (same wave wrote rsave: LDS operations of one wave complete in order - and wave_lds_sync() tells the compiler, to which the
    //  reads below do not depend on another lane's store; round 5 audit, tools/lds_handover_audit.py)
    wave_lds_sync();
#define RS(s_) (rsave[wave][s_])
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        float m = -INFINITY;
#pragma unroll
        for (int s_ = 0; s_ < RPW; ++s_)
            if (s_ * NW < M) m = fmaxf(m, (wave + NW * s_ < M) ? KREF(s_, c) - RS(s_) : -INFINITY);
        part[0][wave][lane + 64 * c] = m;
    }
    wg_barrier();
    float bsc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        float m = part[0][0][lane + 64 * c];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, part[0][w][lane + 64 * c]);
        m = cval[c] ? m : 0.f;
        if (wave == 0) csave[lane + 64 * c] = m;
        bsc[c] = m;                                     // temporarily the stabiliser c_j
    }
    // ---- K = exp(Z - r - c), zero outside the matrix -------------------------------------------
#pragma unroll
    for (int s_ = 0; s_ < RPW; ++s_) {
        const float r_s = RS(s_);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const bool v = (wave + NW * s_ < M) && cval[c];
            KREF(s_, c) = v ? fast_exp2(((KREF(s_, c) - r_s) - bsc[c]) * LOG2E) : 0.f;
        }
    }
    // marginals: nu per owned column; mu in the reduced layout (register i, group g <-> slot)
    float nu[CPL], mured[N2], ared[N2];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        nu[c] = cval[c] ? expf(lnu[lane + 64 * c]) : 0.f;
        bsc[c] = cval[c] ? expf(bsc[c]) : 0.f;          // b starts at exp(c_j)
    }
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const int row = wave + NW * RSum::slot_of(i, grp);
        mured[i] = (RSum::slot_of(i, grp) < RPW && row < M) ? expf(lmu[row]) : 0.f;
        ared[i] = 0.f;
    }
    wg_barrier();        // part[0] is free again

    for (int it = 0; it < iters; ++it) {
        // ---- a_i = mu_i / sum_j K_ij b_j ----------------------------------------------------------
        float sred[N2];
        RSum::run([&](int s_) {
            float acc = KREF(s_, 0) * bsc[0];
#pragma unroll
            for (int c = 1; c < CPL; ++c) acc = fmaf(KREF(s_, c), bsc[c], acc);
            return acc; }, sred, OpSum(), 0.f);
#pragma unroll
        for (int i = 0; i < N2; ++i) ared[i] = mured[i] * fast_rcp(sred[i]);   // padding slots: never read
        // ---- b_j = nu_j / sum_i K_ij a_i ----------------------------------------------------------
        float t[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) t[c] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < RPW; ++s_) {
            const float a_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                __builtin_bit_cast(int, ared[RSum::slot_reg(s_)]), 16 * RSum::slot_grp(s_)));
            const float a_u = (wave + NW * s_ < M) ? a_s : 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) t[c] = fmaf(KREF(s_, c), a_u, t[c]);
        }
        float (*pb)[CW] = part[it & 1];
#pragma unroll
        for (int c = 0; c < CPL; ++c) pb[wave][lane + 64 * c] = t[c];
        wg_barrier();
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += pb[w][lane + 64 * c];
            bsc[c] = cval[c] ? nu[c] * fast_rcp(tot) : 0.f;
        }
    }

    // ---- guard -------------------------------------------------------------------------------
    bool ok = true;
#pragma unroll
    for (int c = 0; c < CPL; ++c) ok = ok && (!cval[c] || scaling_ok(bsc[c]));
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const int sl = RSum::slot_of(i, grp);
        ok = ok && (!(sl < RPW && wave + NW * sl < M) || scaling_ok(ared[i]));
    }
    const bool okw = __all(ok);
    wg_barrier();
    if (lane == 0) ok_s[wave] = okw ? 1 : 0;
    wg_barrier();
    bool all_ok = iters > 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) all_ok = all_ok && ok_s[w] != 0;
    if (threadIdx.x == 0) fail[b] = all_ok ? 0 : 1;
    if (!all_ok) return;            // sinkhorn_wg_kernel will solve this problem

    // ---- duals and epilogue: ((Z + u) + v) - norm ----------------------------------------------
    float ured[N2], vs[CPL];
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const int sl = RSum::slot_of(i, grp);
        ured[i] = logf(ared[i]) - rsave[wave][sl < RPW ? sl : RPW];
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) vs[c] = logf(bsc[c]) - csave[lane + 64 * c];
    const float norm = norm_in ? norm_in[b] : 0.f;
    float* ob = out + (int64_t)b * M * N;
#pragma unroll
    for (int s_ = 0; s_ < RPW; ++s_) {
        const int i = wave + NW * s_;
        const float u_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
            __builtin_bit_cast(int, ured[RSum::slot_reg(s_)]), 16 * RSum::slot_grp(s_)));
        if (i < M) {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (cval[c]) {
                    const int j = lane + 64 * c;
                    float z = (src_at(src, sb, i, j) + u_s) + vs[c];
                    if (norm_in) z = z - norm;
                    ob[(int64_t)i * N + j] = z;
                }
        }
    }
#undef KREF
#undef RS
}

// ------------------------------------------------------------------------------------------
// sinkhorn_cu2_kernel (round 6): the same solve, the same bits, ~25 % fewer instructions a sweep.
// At ONE pair a step - the reference's execution mode - this kernel is a third of a pair's GPU time
// (462 us: one CU per problem, 615 instructions a sweep and wave, VALU-bound).  Two changes:
//   * row slots s and s + N1 - the two partials the transposed row reduction consumes side by
//     side - share a register PAIR, so the row pass is one v_pk_fma_f32 chain per pair: 9 x 5
//     packed + 5 scalar FMAs instead of 95 (same operands, same order per row: same bits).  Pairs
//     (7,17), (8,18) and the single slot 9 are the five slots that live in LDS (as 8-byte pairs).
//   * the cross-wave column sums are formed ONCE (thread j < 320 adds the 16 partials of column j
//     in the old order and publishes b_j) instead of by every wave for itself: 16 + 5 LDS reads a
//     lane of five waves instead of 80 a lane of all sixteen, at the price of a second barrier.
// sinkhorn_cu_kernel stays as the A/B partner (diagnostic library: PATS_CU_V1=1); the GPU test
// holds the two to torch.equal.
// ------------------------------------------------------------------------------------------
template <int RPW, int CPL>
__global__ void __launch_bounds__(1024)
sinkhorn_cu2_kernel(SrcView src, int M, int N, const float* __restrict__ log_mu,
                    const float* __restrict__ log_nu, const float* __restrict__ norm_in, int iters,
                    float* __restrict__ out, int* __restrict__ fail) {
    constexpr int NW = 16, CW = CPL * 64;
    using RSum = RowReduce<RPW, OpSum>;
    using RMax = RowReduce<RPW, OpMax>;
    constexpr int N1 = RSum::N1, N2 = RSum::N2;
    constexpr int PREG = 7;                       // pairs (i, i + N1), i < PREG, in registers
    constexpr int PLDS = N1 - 1 - PREG;           // pairs PREG .. N1 - 2 in LDS; slot N1 - 1 is single (RPW odd)
    static_assert(RPW == 2 * N1 - 1 && PLDS == 2, "19 row slots: seven register pairs, two LDS pairs, one single slot");
    __shared__ float part[NW][CW];                // cross-wave column partials
    __shared__ f2p klp[PLDS][NW][CW];             // slots (7, 17) and (8, 18)
    __shared__ float kls[NW][CW];                 // slot 9
    __shared__ float bl[CW];                      // the column scalings of the sweep
    __shared__ float rsave[NW][RPW + 1];
    __shared__ float csave[CW];
    __shared__ int ok_s[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4;
    const int b = blockIdx.x;
    const float* sb = src.base + (int64_t)b * src.stride;
    const float* lmu = log_mu + (int64_t)b * M;
    const float* lnu = log_nu + (int64_t)b * N;

    f2p K2[PREG][CPL];
    // element (slot s_, column slice c) of this lane's block
#define KPI(s_) ((s_) < N1 ? (s_) : (s_) - N1)
#define KPH(s_) ((s_) < N1 ? 0 : 1)
#define KGET(s_, c) (KPI(s_) < PREG ? K2[KPI(s_) < PREG ? KPI(s_) : 0][c][KPH(s_)] \
                     : KPI(s_) < N1 - 1 ? klp[KPI(s_) < N1 - 1 && KPI(s_) >= PREG ? KPI(s_) - PREG : 0][wave][lane + 64 * (c)][KPH(s_)] \
                     : kls[wave][lane + 64 * (c)])
#define KSET(s_, c, v_) do { const float kv_ = (v_); \
        if (KPI(s_) < PREG) K2[KPI(s_) < PREG ? KPI(s_) : 0][c][KPH(s_)] = kv_; \
        else if (KPI(s_) < N1 - 1) klp[KPI(s_) < N1 - 1 && KPI(s_) >= PREG ? KPI(s_) - PREG : 0][wave][lane + 64 * (c)][KPH(s_)] = kv_; \
        else kls[wave][lane + 64 * (c)] = kv_; } while (0)
    bool cval[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) cval[c] = lane + 64 * c < N;

    // ---- load Z block (rows wave + 16 s, columns lane + 64 c); -inf outside the matrix ----------
#pragma unroll
    for (int s_ = 0; s_ < RPW; ++s_) {
        const int i = wave + NW * s_;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            KSET(s_, c, (i < M && cval[c]) ? src_at(src, sb, i, lane + 64 * c) : -INFINITY);
    }
    // ---- r_i = max_j Z_ij (transposed wave reduction) ------------------------------------------
    {
        float rred[N2];
        RMax::run([&](int s_) {
            float m = KGET(s_, 0);
#pragma unroll
            for (int c = 1; c < CPL; ++c) m = fmaxf(m, KGET(s_, c));
            return m; }, rred, OpMax(), -INFINITY);
#pragma unroll
        for (int i = 0; i < N2; ++i)
            if ((lane & 15) == 0 && RSum::slot_of(i, grp) < RPW) rsave[wave][RSum::slot_of(i, grp)] = rred[i];
    }
    wave_lds_sync();
#define RS(s_) (rsave[wave][s_])
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        float m = -INFINITY;
#pragma unroll
        for (int s_ = 0; s_ < RPW; ++s_)
            if (s_ * NW < M) m = fmaxf(m, (wave + NW * s_ < M) ? KGET(s_, c) - RS(s_) : -INFINITY);
        part[wave][lane + 64 * c] = m;
    }
    wg_barrier();
    float bsc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        float m = part[0][lane + 64 * c];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, part[w][lane + 64 * c]);
        m = cval[c] ? m : 0.f;
        if (wave == 0) csave[lane + 64 * c] = m;
        bsc[c] = m;                                     // temporarily the stabiliser c_j
    }
    // ---- K = exp(Z - r - c), zero outside the matrix -------------------------------------------
#pragma unroll
    for (int s_ = 0; s_ < RPW; ++s_) {
        const float r_s = RS(s_);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const bool v = (wave + NW * s_ < M) && cval[c];
            KSET(s_, c, v ? fast_exp2(((KGET(s_, c) - r_s) - bsc[c]) * LOG2E) : 0.f);
        }
    }
    // marginals: nu of the column this THREAD reduces (j = threadIdx.x < CW); mu in the reduced layout
    const int jb = threadIdx.x;
    const float nu_b = (jb < CW && jb < N) ? expf(lnu[jb < N ? jb : 0]) : 0.f;
    float mured[N2], ared[N2];
#pragma unroll
    for (int c = 0; c < CPL; ++c) bsc[c] = cval[c] ? expf(bsc[c]) : 0.f;          // b starts at exp(c_j)
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const int row = wave + NW * RSum::slot_of(i, grp);
        mured[i] = (RSum::slot_of(i, grp) < RPW && row < M) ? expf(lmu[row]) : 0.f;
        ared[i] = 0.f;
    }
    wg_barrier();        // part is free again

    for (int it = 0; it < iters; ++it) {
        // ---- a_i = mu_i / sum_j K_ij b_j: slots i and i + N1 as one packed chain ---------------------
        f2p bb[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) bb[c] = f2p{bsc[c], bsc[c]};
        float sred[N2];
        RSum::run_pairs([&](int i) {
            if (i < PREG) {
                f2p acc = K2[i < PREG ? i : 0][0] * bb[0];
#pragma unroll
                for (int c = 1; c < CPL; ++c) acc = __builtin_elementwise_fma(K2[i < PREG ? i : 0][c], bb[c], acc);
                return acc;
            }
            if (i < N1 - 1) {
                const int q = i >= PREG && i < N1 - 1 ? i - PREG : 0;
                f2p acc = klp[q][wave][lane] * bb[0];
#pragma unroll
                for (int c = 1; c < CPL; ++c) acc = __builtin_elementwise_fma(klp[q][wave][lane + 64 * c], bb[c], acc);
                return acc;
            }
            float acc = kls[wave][lane] * bsc[0];
#pragma unroll
            for (int c = 1; c < CPL; ++c) acc = fmaf(kls[wave][lane + 64 * c], bsc[c], acc);
            return f2p{acc, 0.f}; }, sred, OpSum(), 0.f);
#pragma unroll
        for (int i = 0; i < N2; ++i) ared[i] = mured[i] * fast_rcp(sred[i]);   // padding slots: never read
        // ---- b_j = nu_j / sum_i K_ij a_i: per-wave partials in the old row order ... ------------------
        float t[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) t[c] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < RPW; ++s_) {
            const float a_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                __builtin_bit_cast(int, ared[RSum::slot_reg(s_)]), 16 * RSum::slot_grp(s_)));
            const float a_u = (wave + NW * s_ < M) ? a_s : 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) t[c] = fmaf(KGET(s_, c), a_u, t[c]);
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) part[wave][lane + 64 * c] = t[c];
        wg_barrier();
        // ---- ... summed over the waves ONCE, by the thread that owns the column (same order w = 0 .. 15) ---
        if (jb < CW) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += part[w][jb];
            bl[jb] = jb < N ? nu_b * fast_rcp(tot) : 0.f;
        }
        wg_barrier();
#pragma unroll
        for (int c = 0; c < CPL; ++c) bsc[c] = bl[lane + 64 * c];
    }

    // ---- guard -------------------------------------------------------------------------------
    bool ok = true;
#pragma unroll
    for (int c = 0; c < CPL; ++c) ok = ok && (!cval[c] || scaling_ok(bsc[c]));
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const int sl = RSum::slot_of(i, grp);
        ok = ok && (!(sl < RPW && wave + NW * sl < M) || scaling_ok(ared[i]));
    }
    const bool okw = __all(ok);
    wg_barrier();
    if (lane == 0) ok_s[wave] = okw ? 1 : 0;
    wg_barrier();
    bool all_ok = iters > 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) all_ok = all_ok && ok_s[w] != 0;
    if (threadIdx.x == 0) fail[b] = all_ok ? 0 : 1;
    if (!all_ok) return;            // sinkhorn_wg_kernel will solve this problem

    // ---- duals and epilogue: ((Z + u) + v) - norm ----------------------------------------------
    float ured[N2], vs[CPL];
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const int sl = RSum::slot_of(i, grp);
        ured[i] = logf(ared[i]) - rsave[wave][sl < RPW ? sl : RPW];
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) vs[c] = logf(bsc[c]) - csave[lane + 64 * c];
    const float norm = norm_in ? norm_in[b] : 0.f;
    float* ob = out + (int64_t)b * M * N;
#pragma unroll
    for (int s_ = 0; s_ < RPW; ++s_) {
        const int i = wave + NW * s_;
        const float u_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
            __builtin_bit_cast(int, ured[RSum::slot_reg(s_)]), 16 * RSum::slot_grp(s_)));
        if (i < M) {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (cval[c]) {
                    const int j = lane + 64 * c;
                    float z = (src_at(src, sb, i, j) + u_s) + vs[c];
                    if (norm_in) z = z - norm;
                    ob[(int64_t)i * N + j] = z;
                }
        }
    }
#undef KPI
#undef KPH
#undef KGET
#undef KSET
#undef RS
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t stream_workspace_bytes(int64_t batch, int M, int N);
static inline size_t al256_fwd(int64_t batch, int M, int N) { return align256(stream_workspace_bytes(batch, M, N)); }

struct OtWorkspace {
    float *log_mu, *log_nu, *norm, *Zw, *Zt;
    int* fail;
    void* stream;
    size_t bytes;
};
static OtWorkspace carve(void* ws, int64_t batch, int M, int N, bool marginals, bool zw) {
    OtWorkspace w{};
    size_t off = 0;
    char* base = (char*)ws;
    auto take = [&](size_t n) { float* p = (float*)(base + off); off += align256(n * sizeof(float)); return p; };
    if (marginals) {
        w.log_mu = take((size_t)batch * M);
        w.log_nu = take((size_t)batch * N);
        w.norm = take((size_t)batch);
    }
    w.fail = (int*)take((size_t)batch);
    if (zw) w.Zw = take((size_t)batch * M * N);
    w.Zt = take((size_t)batch * M * N);
    if ((int64_t)M * N > 304 * 320 && N <= 512 * 9) {       // streaming solver's K / partials / vectors
        w.stream = base + off;
        off += al256_fwd(batch, M, N);
    }
    w.bytes = off;
    return w;
}

static int launch_wg(const SrcView& src, int64_t batch, int M, int N, const float* log_mu,
                     const float* log_nu, const float* norm, int iters, float bias_k, float* out,
                     float* Zw, float* Zt, hipStream_t st, const int* only_if = nullptr) {
    const int threads = ((int64_t)M * N >= 128 * 128) ? 1024 : 256;
    const size_t lds = (size_t)(M + N) * sizeof(float);
    PATS_REQUIRE(lds <= 64 * 1024, "sinkhorn: M+N=%d too large for the one-workgroup kernel", M + N);
    hipLaunchKernelGGL(sinkhorn_wg_kernel, dim3((unsigned)batch), dim3(threads), lds, st, src, M, N,
                       log_mu, log_nu, norm, iters, bias_k, out, Zw, Zt, only_if, only_if ? fallback_counter() : nullptr);
    return check_launch("sinkhorn_wg_kernel");
}

size_t stream_workspace_bytes(int64_t batch, int M, int N);
int launch_stream(const float* base, int64_t stride, int ld, int rows, int cols, const float* alpha,
                  int64_t batch, int M, int N, const float* log_mu, const float* log_nu, const float* norm,
                  int iters, float* out, void* ws, int* fail, hipStream_t st);
static inline bool stream_shape(int M, int N) { return (int64_t)M * N > 304 * 320 && N <= 512 * 9; }

constexpr int CU_RPW = 19, CU_CPL = 5, CU_LSLOTS = 5;            // 16 * 19 = 304 rows, 64 * 5 = 320 columns
static inline bool cu_shape(int M, int N) { return M <= 16 * CU_RPW && N <= 64 * CU_CPL && M * N >= 96 * 96; }

// one-CU linear-domain kernel, then the log-domain kernel on the problems it flagged
static int launch_cu_then_fallback(const SrcView& src, int64_t batch, int M, int N, const float* log_mu,
                                   const float* log_nu, const float* norm, int iters, float* out,
                                   const OtWorkspace& w, hipStream_t st) {
    // sinkhorn_cu2_kernel: the same bits in ~25 % fewer instructions (packed row pass, column sums formed once); the first
    // version stays selectable in the diagnostic library (PATS_CU_V1=1) as its A/B partner
    static const bool v1 = [] { const char* e = diag_env("PATS_CU_V1"); return e && atoi(e) != 0; }();
    if (v1) hipLaunchKernelGGL((sinkhorn_cu_kernel<CU_RPW, CU_CPL, CU_LSLOTS>), dim3((unsigned)batch), dim3(1024), 0, st, src, M,
                               N, log_mu, log_nu, norm, iters, out, w.fail);
    else hipLaunchKernelGGL((sinkhorn_cu2_kernel<CU_RPW, CU_CPL>), dim3((unsigned)batch), dim3(1024), 0, st, src, M,
                            N, log_mu, log_nu, norm, iters, out, w.fail);
    int rc = check_launch("sinkhorn_cu_kernel");
    if (rc) return rc;
    return launch_wg(src, batch, M, N, log_mu, log_nu, norm, iters, 0.f, out, w.Zw, w.Zt, st, w.fail);
}

}  // namespace pats

using namespace pats;

constexpr int NF = 145;   // fine level (12 x 12 + dustbin)
static inline bool resident_shape(int M, int N) { return (M == NT && N == NT) || (M == NF && N == NF); }
static inline int use_linear() { return sinkhorn_mode() != PATS_SINKHORN_LOG; }
// how sinkhorn_rc_kernel re-solves the problems the block kernel flagged: 2 = stabilised linear sweeps (absorption; a problem
// that still leaves the guard ends in the log-sum-exp sweeps inside the same launch), 0 = log-sum-exp sweeps at once (round 3;
// PATS_FINE_REDO_LOG=1)
static inline int redo_mode() {
    static const int m = diag_env("PATS_FINE_REDO_LOG") ? 0 : 2;
    return m;
}

namespace pats {
int launch_blk145(int mode, const float* Z, int64_t batch, const float* log_mu, const float* log_nu,
                  const float* ns, const float* one, int iters, float bias_k, float* out, int* fail,
                  uint8_t* col_nomatch, hipStream_t st, const int64_t* live = nullptr);     // sinkhorn_blk.hip
int launch_col_flags(const float* Z, int64_t batch, int M, int N, uint8_t* col_nomatch, const int* only_if,
                     hipStream_t st);                       // post.hip
bool fine_w2_enabled();                                     // sinkhorn_blk2w.hip: the two-wave form of the same kernel
int launch_blk145_w2(int mode, const float* Z, int64_t batch, const float* log_mu, const float* log_nu, const float* ns,
                     const float* one, int iters, float bias_k, float* out, int* fail, uint8_t* col_nomatch, hipStream_t st,
                     const int64_t* live);
}

// 145 x 145: the register-block kernel solves in the linear domain and flags the problems whose
// scalings left the guard band; sinkhorn_rc_kernel then re-solves exactly those with log-sum-exp
// sweeps.  Without flag storage (no workspace), for iters == 0 or in forced-log mode the rc kernel
// does everything, as before.
static int launch_fine145(int mode, const float* Z, int64_t batch, const float* log_mu, const float* log_nu,
                          const float* ns, const float* one, int iters, float bias_k, float* out, int* fail,
                          hipStream_t st, uint8_t* col_nomatch = nullptr, const int64_t* live = nullptr) {
    static const bool v1_only = diag_env("PATS_FINE_V1") != nullptr;      // A/B switch for benchmarking
    const bool blk = use_linear() && iters > 0 && fail && !v1_only;
    if (blk) {
        int rc = fine_w2_enabled() ? launch_blk145_w2(mode, Z, batch, log_mu, log_nu, ns, one, iters, bias_k, out, fail, col_nomatch, st, live)
                                   : launch_blk145(mode, Z, batch, log_mu, log_nu, ns, one, iters, bias_k, out, fail, col_nomatch, st, live);
        if (rc) return rc;
    }
    if (mode == 0)
        hipLaunchKernelGGL((sinkhorn_rc_kernel<NF, 0>), dim3((unsigned)batch), dim3(384), 0, st, Z, batch, log_mu,
                           log_nu, nullptr, nullptr, iters, 0.f, blk ? redo_mode() : (int)use_linear(), out,
                           fallback_counter(), blk ? fail : nullptr, live);
    else
        hipLaunchKernelGGL((sinkhorn_rc_kernel<NF, 2>), dim3((unsigned)batch), dim3(384), 0, st, Z, batch, nullptr,
                           nullptr, ns, one, iters, bias_k, blk ? redo_mode() : (int)use_linear(), out, fallback_counter(),
                           blk ? fail : nullptr, live);
    int rc = check_launch("sinkhorn_rc_kernel<145>");
    // column flags of the problems the log-domain kernel (re-)solved: all of them without the block kernel
    if (!rc && col_nomatch) rc = launch_col_flags(out, batch, NF, NF, col_nomatch, blk ? fail : nullptr, st);
    return rc;
}

// The fused fine-level step (sinkhorn_blk.hip, FUSED): descriptors -> log-plan in one kernel; problems that leave the guard
// band come back as their raw scores in `out` and the log-domain kernel redoes them IN PLACE (it stages the whole matrix in
// LDS before it writes).  *applied = false: the caller takes the two-kernel path (the default; also forced log domain, iters == 0).
namespace pats {
bool fine_fused();       // host.cpp
int launch_blk145_fused(const float* d0, const float* d1, int D, int64_t batch, const float* ns, const float* one, int iters,
                        float bias_k, float* out, int* fail, uint8_t* col_nomatch, hipStream_t st, const int64_t* live);
int launch_fine145_fused(const float* d0, const float* d1, int D, int64_t batch, const float* ns, const float* one, int iters,
                         float bias_k, float* out, int* fail, uint8_t* col_nomatch, hipStream_t st, bool* applied,
                         const int64_t* live) {
    static const bool v1_only = diag_env("PATS_FINE_V1") != nullptr;
    *applied = fine_fused() && use_linear() && iters > 0 && fail && !v1_only && D > 0;       // pats_set_fine_fused / PATS_FINE_FUSED
    if (!*applied) return PATS_OK;
    int rc = launch_blk145_fused(d0, d1, D, batch, ns, one, iters, bias_k, out, fail, col_nomatch, st, live);
    if (rc) return rc;
    hipLaunchKernelGGL((sinkhorn_rc_kernel<NF, 2>), dim3((unsigned)batch), dim3(384), 0, st, (const float*)out, batch,
                       (const float*)nullptr, (const float*)nullptr, ns, one, iters, bias_k, redo_mode(), out, fallback_counter(),
                       (const int*)fail, live);
    rc = check_launch("sinkhorn_rc_kernel<145>(redo)");
    if (!rc && col_nomatch) rc = launch_col_flags(out, batch, NF, NF, col_nomatch, fail, st);
    return rc;
}
}  // namespace pats

extern "C" size_t pats_sinkhorn_workspace_bytes(int64_t batch, int M, int N) {
    if (M == NF && N == NF) return (size_t)((batch + 63) & ~63ll) * sizeof(int);     // guard flags (optional: see launch_fine145)
    if (resident_shape(M, N)) return 0;
    return carve(nullptr, batch, M, N, false, false).bytes;
}

extern "C" size_t pats_ot_workspace_bytes(int64_t batch, int M, int N) {
    return carve(nullptr, batch, M, N, true, true).bytes;
}

// log_optimal_transport2: the resident shapes compute their marginals in-kernel and keep the plan on
// chip, so they need no marginal / transposed-copy storage: 65 x 65 nothing, 145 x 145 one guard flag
// per problem (at the START of the workspace, so a flag-sized workspace is enough).
extern "C" size_t pats_ot2_workspace_bytes(int64_t batch, int m, int n) {
    if (m == NT && n == NT) return 0;
    if (m == NF && n == NF) return pats_sinkhorn_workspace_bytes(batch, m, n);
    return pats_ot_workspace_bytes(batch, m, n);
}

extern "C" int pats_sinkhorn_f32(const float* Z, int64_t batch, int M, int N, const float* log_mu,
                                 const float* log_nu, int iters, float* out, void* workspace,
                                 size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && M > 0 && N > 0 && iters >= 0, "sinkhorn: bad shape b=%lld M=%d N=%d",
                 (long long)batch, M, N);
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(Z && log_mu && log_nu && out, "sinkhorn: null pointer");
    hipStream_t st = as_stream(stream);
    if (M == NT && N == NT) {
        Ot65Args g{};
        g.fallbacks = fallback_counter();
        g.Zin = Z; g.P = batch; g.log_mu = log_mu; g.log_nu = log_nu; g.iters = iters;
        g.linear = use_linear(); g.out = out;
        hipLaunchKernelGGL((sinkhorn65_kernel<0, 0, 0>), dim3((unsigned)batch), dim3(64), 0, st, g);
        return check_launch("sinkhorn65_kernel<0,0,0>");
    }
    if (M == NF && N == NF) {
        int* fail = (workspace && workspace_bytes >= pats_sinkhorn_workspace_bytes(batch, M, N)) ? (int*)workspace : nullptr;
        return launch_fine145(0, Z, batch, log_mu, log_nu, nullptr, nullptr, iters, 0.f, out, fail, st);
    }
    PATS_REQUIRE(workspace && workspace_bytes >= pats_sinkhorn_workspace_bytes(batch, M, N),
                 "sinkhorn: workspace too small");
    OtWorkspace w = carve(workspace, batch, M, N, false, false);
    SrcView src{Z, (int64_t)M * N, N, M, N, nullptr};   // (w.fail / w.stream are carved for every shape)
    if (use_linear() && cu_shape(M, N))
        return launch_cu_then_fallback(src, batch, M, N, log_mu, log_nu, nullptr, iters, out, w, st);
    if (use_linear() && stream_shape(M, N) && batch <= 65535) {
        int rc = launch_stream(Z, (int64_t)M * N, N, M, N, nullptr, batch, M, N, log_mu, log_nu, nullptr, iters,
                               out, w.stream, w.fail, st);
        if (rc) return rc;
        return launch_wg(src, batch, M, N, log_mu, log_nu, nullptr, iters, 0.f, out, nullptr, w.Zt, st, w.fail);
    }
    return launch_wg(src, batch, M, N, log_mu, log_nu, nullptr, iters, 0.f, out, nullptr, w.Zt, st);
}

extern "C" int pats_log_optimal_transport_f32(const float* scores, int64_t batch, int m, int n,
                                              const float* alpha, const float* ns, int iters,
                                              float* Z, void* workspace, size_t workspace_bytes,
                                              pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && m > 0 && n > 0 && iters >= 0, "log_optimal_transport: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(scores && alpha && ns && Z, "log_optimal_transport: null pointer");
    const int M = m + 1, N = n + 1;
    PATS_REQUIRE(workspace && workspace_bytes >= pats_ot_workspace_bytes(batch, M, N),
                 "log_optimal_transport: workspace too small");
    hipStream_t st = as_stream(stream);
    OtWorkspace w = carve(workspace, batch, M, N, true, true);
    hipLaunchKernelGGL(ot_prep_kernel, dim3((unsigned)batch), dim3(256), 0, st, ns, n, M, N,
                       (float)m, nullptr, w.log_mu, w.log_nu, w.norm);
    int rc = check_launch("ot_prep_kernel");
    if (rc) return rc;
    SrcView src{scores, (int64_t)m * n, n, m, n, alpha};
    if (use_linear() && cu_shape(M, N))
        return launch_cu_then_fallback(src, batch, M, N, w.log_mu, w.log_nu, w.norm, iters, Z, w, st);
    if (use_linear() && stream_shape(M, N) && batch <= 65535) {
        rc = launch_stream(scores, (int64_t)m * n, n, m, n, alpha, batch, M, N, w.log_mu, w.log_nu, w.norm, iters,
                           Z, w.stream, w.fail, st);
        if (rc) return rc;
        return launch_wg(src, batch, M, N, w.log_mu, w.log_nu, w.norm, iters, 0.f, Z, w.Zw, w.Zt, st, w.fail);
    }
    return launch_wg(src, batch, M, N, w.log_mu, w.log_nu, w.norm, iters, 0.f, Z, w.Zw, w.Zt, st);
}

static int ot2_impl(const float* scores, int64_t batch, int m, int n, const float* one, const float* ns, int iters,
                    float bias_k, float* Z, void* workspace, size_t workspace_bytes, pats_stream_t stream,
                    uint8_t* col_nomatch, const int64_t* live = nullptr);

// log_optimal_transport2 + column flags over a capacity of `batch` 145 x 145 problems, *live of them in use (fused.hip)
namespace pats {
int ot2_flags_live(const float* scores, int64_t batch, int m, int n, const float* one, const float* ns, int iters, float bias_k,
                   float* Z, uint8_t* col_nomatch, void* workspace, size_t workspace_bytes, pats_stream_t stream,
                   const int64_t* live) {
    return ot2_impl(scores, batch, m, n, one, ns, iters, bias_k, Z, workspace, workspace_bytes, stream, col_nomatch, live);
}
}  // namespace pats

extern "C" int pats_log_optimal_transport2_f32(const float* scores, int64_t batch, int m, int n,
                                               const float* one, const float* ns, int iters,
                                               float bias_k, float* Z, void* workspace,
                                               size_t workspace_bytes, pats_stream_t stream) {
    return ot2_impl(scores, batch, m, n, one, ns, iters, bias_k, Z, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int pats_log_optimal_transport2_flags_f32(const float* scores, int64_t batch, int m, int n,
                                                     const float* one, const float* ns, int iters,
                                                     float bias_k, float* Z, uint8_t* col_nomatch, void* workspace,
                                                     size_t workspace_bytes, pats_stream_t stream) {
    return ot2_impl(scores, batch, m, n, one, ns, iters, bias_k, Z, workspace, workspace_bytes, stream, col_nomatch);
}

static int ot2_impl(const float* scores, int64_t batch, int m, int n, const float* one, const float* ns, int iters,
                    float bias_k, float* Z, void* workspace, size_t workspace_bytes, pats_stream_t stream,
                    uint8_t* col_nomatch, const int64_t* live) {
    PATS_REQUIRE(!live || (m == NF && n == NF), "log_optimal_transport2: a device-side count is taken for 145 x 145 problems only");
    PATS_REQUIRE(batch >= 0 && m > 1 && n > 1 && iters >= 0, "log_optimal_transport2: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(scores && ns && Z, "log_optimal_transport2: null pointer");
    hipStream_t st = as_stream(stream);
    if (m == NT && n == NT) {
        Ot65Args g{};
        g.fallbacks = fallback_counter();
        g.Zin = scores; g.P = batch; g.ns = ns; g.one = one; g.iters = iters; g.bias_k = bias_k;
        g.linear = use_linear(); g.out = Z;
        hipLaunchKernelGGL((sinkhorn65_kernel<2, 0, 0>), dim3((unsigned)batch), dim3(64), 0, st, g);
        int rc = check_launch("sinkhorn65_kernel<2,0,0>");
        if (!rc && col_nomatch) rc = launch_col_flags(Z, batch, m, n, col_nomatch, nullptr, st);
        return rc;
    }
    if (m == NF && n == NF) {
        int* fail = (workspace && workspace_bytes >= pats_ot2_workspace_bytes(batch, m, n)) ? (int*)workspace : nullptr;
        return launch_fine145(2, scores, batch, nullptr, nullptr, ns, one, iters, bias_k, Z, fail, st, col_nomatch, live);
    }
    PATS_REQUIRE(workspace && workspace_bytes >= pats_ot2_workspace_bytes(batch, m, n),
                 "log_optimal_transport2: workspace too small");   // shapes without a resident kernel
    OtWorkspace w = carve(workspace, batch, m, n, true, true);
    hipLaunchKernelGGL(ot_prep_kernel, dim3((unsigned)batch), dim3(256), 0, st, ns, n - 1, m, n,
                       (float)(m - 1), one, w.log_mu, w.log_nu, w.norm);
    int rc = check_launch("ot_prep_kernel");
    if (rc) return rc;
    SrcView src{scores, (int64_t)m * n, n, m, n, nullptr};
    rc = launch_wg(src, batch, m, n, w.log_mu, w.log_nu, w.norm, iters, bias_k, Z, nullptr, w.Zt, st);
    if (!rc && col_nomatch) rc = launch_col_flags(Z, batch, m, n, col_nomatch, nullptr, st);
    return rc;
}

// descriptors -> log-plan for 65x65 problems in one launch (cost build + OT2), used by pats_cost_ot_f32
namespace pats {
int launch_cost_ot65(const float* d0, const float* d1, int64_t batch, int D, const float* one,
                     const float* ns, int iters, float bias_k, float* Z, pats_stream_t stream) {
    Ot65Args g{};
    g.fallbacks = fallback_counter();
    g.d0 = d0; g.d1 = d1; g.D = D; g.P = batch; g.ns = ns; g.one = one; g.iters = iters;
    g.bias_k = bias_k; g.linear = use_linear(); g.out = Z;
    hipLaunchKernelGGL((sinkhorn65_kernel<2, 1, 0>), dim3((unsigned)batch), dim3(64), 0, as_stream(stream), g);
    return check_launch("sinkhorn65_kernel<2,1,0>");
}
}  // namespace pats


// third-level step of a batch whose problem count lives on the device (throughput mode: no host read between the merge
// that decides P and this launch): the grid covers the capacity P_cap, waves past *P_dev leave at once
extern "C" int pats_third_level_counted_f32(const float* feat0, const float* feat1, int64_t P_cap, const int64_t* P_dev, int D,
                                            const float* scale, const float* scale_x, const float* scale_y,
                                            const int64_t* p_s, const int64_t* p_t, int iters, int outdoor,
                                            float* mkpts0_f, float* mkpts1_f, float* label, uint8_t* if_matching1,
                                            pats_stream_t stream) {
    PATS_REQUIRE(P_cap >= 0 && D > 0 && (D % 32) == 0 && D <= 512 && iters >= 0,
                 "third_level_counted: bad shape (D must be a multiple of 32, at most 512)");
    if (P_cap == 0) return PATS_OK;
    PATS_REQUIRE(P_dev && feat0 && feat1 && scale && p_s && p_t && mkpts0_f && mkpts1_f && label && if_matching1 &&
                     ((scale_x == nullptr) == (scale_y == nullptr)), "third_level_counted: null pointer");
    Fused65Args f{feat0, feat1, D, P_cap, scale, nullptr, iters, 1, scale_x, scale_y, p_s, p_t, outdoor,
                  ComputeResultOut{mkpts0_f, mkpts1_f, nullptr, label, if_matching1, nullptr}, 0, nullptr, 0, P_dev};
    return launch_third_fused(f, as_stream(stream));
}

extern "C" int pats_third_level_f32(const float* feat0, const float* feat1, int64_t P, int D,
                                    const float* scale, const float* scale_x, const float* scale_y,
                                    const int64_t* p_s, const int64_t* p_t, int iters, int outdoor,
                                    float* mkpts0_f, float* mkpts1_f, float* label,
                                    uint8_t* if_matching1, float* Z_out, pats_stream_t stream) {
    PATS_REQUIRE(P >= 0 && D > 0 && (D % 32) == 0 && D <= 512 && iters >= 0,
                 "third_level: bad shape (D must be a multiple of 32, at most 512: the in-wave cost build runs whole prefetch rings)");
    if (P == 0) return PATS_OK;
    PATS_REQUIRE(feat0 && feat1 && scale && scale_x && scale_y && p_s && p_t && mkpts0_f && mkpts1_f &&
                     label && if_matching1, "third_level: null pointer");
    static const bool v1_only = diag_env("PATS_THIRD_V1") != nullptr;     // A/B switch for benchmarking
    if (!Z_out && !v1_only) {      // no plan requested: the 8x8 register-block kernel (third_fused.hip)
        Fused65Args f{feat0, feat1, D, P, scale, nullptr, iters, 1, scale_x, scale_y, p_s, p_t, outdoor,
                      ComputeResultOut{mkpts0_f, mkpts1_f, nullptr, label, if_matching1, nullptr}, 0, nullptr};
        return launch_third_fused(f, as_stream(stream));
    }
    Ot65Args g{};
    g.fallbacks = fallback_counter();
    g.d0 = feat0; g.d1 = feat1; g.D = D; g.P = P; g.ns = scale; g.iters = iters;
    g.linear = use_linear(); g.out = Z_out;
    g.scale_x = scale_x; g.scale_y = scale_y; g.p_s = p_s; g.p_t = p_t; g.outdoor = outdoor;
    g.cr = ComputeResultOut{mkpts0_f, mkpts1_f, nullptr, label, if_matching1, nullptr};
    // one problem ~ (30 + 0.6 * iters) us per wave slot; spread 8 start phases over that period
    // when the launch is at least ~4 full rounds of the 2048 wave slots (s_sleep(127) ~ 3.4 us)
    // (measured on MI355X: unit 1-3 all give ~-11 %, larger units lose it again)
    if (P >= 8192) g.stagger = (int)((30.0f + 0.6f * (float)iters) / 16.0f / 3.4f);
    if (const char* e = diag_env("PATS_STAGGER")) g.stagger = atoi(e);
    hipLaunchKernelGGL((sinkhorn65_kernel<2, 1, 1>), dim3((unsigned)P), dim3(64), 0, as_stream(stream), g);
    return check_launch("sinkhorn65_kernel<2,1,1>");
}
