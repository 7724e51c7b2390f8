// Streaming Sinkhorn for problems too large for one CU (coarse level on big images: 769^2 at
// 1024x768, 1901^2 at 1600 px; BASELINE config 5: 4097^2, 200 sweeps).
//
// Same linear-domain iteration as sinkhorn.hip (a = mu / K b ; b = nu / K^T a on
// K = exp(Z - r - c), models/modules.py:137-143 in kernel-matrix form), but K lives in HBM
// (67 MB at 4097^2: resident in the 256 MB Infinity Cache) and every sweep is TWO launches:
//
//   stream_sweep_kernel   one 512-thread workgroup per block of RB rows.  Thread t owns columns
//                         t, t+512, ... (CPT of them): it loads its RB x CPT piece of K ONCE
//                         (fully coalesced 2 KB row segments), forms the RB partial row dots with
//                         the lane-local b, reduces them (transposed wave all-reduce + 8-wave LDS
//                         combine) to a_i = mu_i / sum, and - K still in registers - accumulates
//                         the column partials sum_i K_ij a_i of its block, stored as one
//                         coalesced row of `partial[block][N]`.  K is read once per sweep, not
//                         twice: HBM/L3 traffic per sweep = 4 M N (+ partials), half the
//                         "two passes" model (8 M N).
//   stream_colreduce_kernel  b_j = nu_j / sum_blocks partial[block][j]   (16 waves split the blocks)
//
// Since round 5 the 4097^2-class shape (one problem, N in 4097..4608, at most one 17-row block per CU) and batches of narrow problems
// (N <= 1024: the 769^2 coarse level of BASELINE config [3], all blocks of all problems resident at once) run ALL their sweeps in ONE
// launch instead: stream_resident_kernel keeps a workgroup's 17 x N piece of K in REGISTERS (153 a lane) for the whole solve, so a
// sweep moves no K at all - 16 KB of column partials out, 17 columns' worth of partials and the 16 KB scaling vector in - and pays two
// grid-wide barriers instead of two kernel boundaries and a 67 MB read.  See the kernel.
//
// Set-up (row max, column max of Z - r, K build) and the epilogue (duals back to log space,
// Z_out = ((Z + u) + v) - norm, guard) are plain streaming kernels.  Deterministic: no atomics.
#include "common.hpp"
#include "lane_reduce.hpp"
#include <algorithm>
#ifdef PATS_DIAG
#include <cstdio>
#endif

namespace pats {

struct SrcViewS {           // same virtual source as sinkhorn.hip: plain [M,N] or scores+alpha border
    const float* base;
    int64_t stride;
    int ld, rows, cols;
    const float* alpha;
};
__device__ __forceinline__ float srcs_at(const SrcViewS& s, const float* b, int i, int j) {
    if (s.alpha && (i >= s.rows || j >= s.cols)) return *s.alpha;
    return b[(int64_t)i * s.ld + j];
}

constexpr int ST = 512;      // threads per workgroup
constexpr int SW = ST / 64;  // waves

// block-wide sum / max of one value per thread, result to all threads (LDS scratch: SW floats)
template <class Op>
__device__ __forceinline__ float block_allreduce(float v, Op op, float* scratch) {
    v = wave_allreduce(v, op);
    wg_barrier();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    wg_barrier();
    float r = scratch[0];
#pragma unroll
    for (int w = 1; w < SW; ++w) r = op(r, scratch[w]);
    return r;
}

// ---- set-up 1: r_i = max_j Z_ij ; one workgroup per row -------------------------------------------
__global__ void __launch_bounds__(ST)
stream_rowmax_kernel(SrcViewS src, int M, int N, float* __restrict__ r) {
    __shared__ float scratch[SW];
    const int i = blockIdx.x, b = blockIdx.y;
    const float* sb = src.base + (int64_t)b * src.stride;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < N; j += ST) m = fmaxf(m, srcs_at(src, sb, i, j));
    m = block_allreduce(m, OpMax(), scratch);
    if (threadIdx.x == 0) r[(int64_t)b * M + i] = m;
}

// ---- set-up 2: partial column max of (Z - r) over a block of RB rows -------------------------------
template <int RB>
__global__ void __launch_bounds__(ST)
stream_colmax_partial_kernel(SrcViewS src, int M, int N, const float* __restrict__ r,
                             float* __restrict__ partial, int nblk) {
    const int blk = blockIdx.x, b = blockIdx.y;
    const float* sb = src.base + (int64_t)b * src.stride;
    const float* rb = r + (int64_t)b * M;
    float* pb = partial + ((int64_t)b * nblk + blk) * N;
    for (int j = threadIdx.x; j < N; j += ST) {
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int i = blk * RB + k;
            if (i < M) m = fmaxf(m, srcs_at(src, sb, i, j) - rb[i]);
        }
        pb[j] = m;
    }
}

// ---- column reduce: MODE 0: c_j = max over blocks ; MODE 1: b_j = nu_j / sum over blocks -----------
// MODE 0 also writes b_j = exp(c_j) (the scaling b starts there) to `aux`.
template <int MODE>
__global__ void __launch_bounds__(1024)
stream_colreduce_kernel(const float* __restrict__ partial, int nblk, int N,
                        const float* __restrict__ log_nu, float* __restrict__ outv,
                        float* __restrict__ aux) {
    __shared__ float sm[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const int j = blockIdx.x * 64 + lane;
    const float* pb = partial + (int64_t)b * nblk * N;
    float acc = MODE == 0 ? -INFINITY : 0.f;
    if (j < N) {
        // up to 18 partials per wave (nblk <= 288) are all in flight before the first add - the plain loop waited for
        // each load in turn, 17 memory latencies at 4097 rows; the order of the additions is the same
        constexpr int Q = 18;
        float v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int k = wave + 16 * q;
            v[q] = pb[(int64_t)(k < nblk ? k : 0) * N + j];      // clamped rows are loaded and not added
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (wave + 16 * q < nblk) acc = MODE == 0 ? fmaxf(acc, v[q]) : acc + v[q];
        for (int k = wave + 16 * Q; k < nblk; k += 16) {
            const float x = pb[(int64_t)k * N + j];
            acc = MODE == 0 ? fmaxf(acc, x) : acc + x;
        }
    }
    sm[wave][lane] = acc;
    wg_barrier();
    if (wave == 0 && j < N) {
        float t = sm[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) t = MODE == 0 ? fmaxf(t, sm[w][lane]) : t + sm[w][lane];
        if (MODE == 0) {
            outv[(int64_t)b * N + j] = t;
            aux[(int64_t)b * N + j] = expf(t);
        } else {
            outv[(int64_t)b * N + j] = expf(log_nu[(int64_t)b * N + j]) * __builtin_amdgcn_rcpf(t);
        }
    }
}

// ---- set-up 3: K = exp((Z - r_i) - c_j) ------------------------------------------------------------
__global__ void __launch_bounds__(ST)
stream_kbuild_kernel(SrcViewS src, int M, int N, const float* __restrict__ r,
                     const float* __restrict__ c, float* __restrict__ K) {
    const int i = blockIdx.x, b = blockIdx.y;
    const float* sb = src.base + (int64_t)b * src.stride;
    const float ri = r[(int64_t)b * M + i];
    const float* cb = c + (int64_t)b * N;
    float* Kr = K + ((int64_t)b * M + i) * N;
    for (int j = threadIdx.x; j < N; j += ST)
        Kr[j] = fast_exp2(((srcs_at(src, sb, i, j) - ri) - cb[j]) * LOG2E);
}

// ---- the sweep ------------------------------------------------------------------------------------
template <int RB, int CPT>
__global__ void __launch_bounds__(ST)
stream_sweep_kernel(const float* __restrict__ K, int M, int N, const float* __restrict__ bvec,
                    const float* __restrict__ log_mu, float* __restrict__ avec,
                    float* __restrict__ partial, int nblk) {
    __shared__ float red[SW][RB];
    __shared__ float a_s[RB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int blk = blockIdx.x, b = blockIdx.y;
    const float* Kb = K + (int64_t)b * M * N;
    const float* bb = bvec + (int64_t)b * N;
    float kv[RB][CPT], bq[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int j = t + ST * q;
        bq[q] = j < N ? bb[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        const int i = blk * RB + k;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int j = t + ST * q;
            kv[k][q] = (i < M && j < N) ? Kb[(int64_t)i * N + j] : 0.f;
        }
    }
    // row dots: per-thread partials, wave all-reduce, then across the 8 waves through LDS
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        float p = 0.f;
#pragma unroll
        for (int q = 0; q < CPT; ++q) p = fmaf(kv[k][q], bq[q], p);
        p = wave_sum(p);
        if (lane == 0) red[wave][k] = p;
    }
    wg_barrier();
    if (t < RB) {
        const int i = blk * RB + t;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < SW; ++w) s += red[w][t];
        const float a = i < M ? expf(log_mu[(int64_t)b * M + i]) * __builtin_amdgcn_rcpf(s) : 0.f;
        a_s[t] = a;
        if (i < M) avec[(int64_t)b * M + i] = a;
    }
    wg_barrier();
    // column partials of this row block, K still in registers
    float* pb = partial + ((int64_t)b * nblk + blk) * N;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < RB; ++k) acc = fmaf(kv[k][q], a_s[k], acc);
        const int j = t + ST * q;
        if (j < N) pb[j] = acc;
    }
}

// ---- all sweeps of one problem in one launch: K register-resident ------------------------------------------------------------
// Grid = nblk workgroups (blocks of RB = 17 rows), ALL co-resident (the host launches it only when nblk <= CUs; a workgroup's 153 +
// registers a lane admit one per CU).  Thread t owns columns 8 t .. 8 t + 7 and, in a ninth slot, column 4096 + t (N <= 4608).
// A sweep:   b -> registers (from the set-up's vector in sweep 0, then from GRANULES, below); row dots -> a (local);
//            column partials of the block -> partial[blk][NP] (rows padded to 16 bytes: two 16-byte stores a thread);
//            GRID BARRIER;
//            workgroup g < ceil(N / 32) reduces columns 32 g .. 32 g + 31 over all blocks in a fixed order (16-byte loads: eight
//            threads a block row, 64 slices of the blocks) -> b_j, published as an 8-byte granule {b_j, sweep + 1}.
// There is no second barrier: a thread of the next sweep polls ITS nine granules until their tags say "this sweep" (the micro-arch
// guide's cheapest transport: one naturally aligned 8-byte {data, tag} written by ONE store).  That is safe: a reducer publishes a
// group only after it has read every partial of it, so a workgroup that holds all granules may overwrite its partial row; and a
// reducer overwrites a granule only behind the next barrier, which every reader of the old one has passed.
// What crosses workgroups is stored and loaded `sc1`: written through to memory and read past the caches that are not coherent
// across XCDs (guide: "16 B sc1 stores AND sc1 loads") - the 16-byte ones as inline assembly (an atomic gives 8 bytes at most; the
// loads' `s_waitcnt vmcnt(0)` is explicit, the compiler does not see them), granules and counters as agent-scope relaxed atomics.
// So the barrier needs NO release / acquire fence (buffer_wbl2 / buffer_inv: 1.7-6.5 us each), only `s_waitcnt vmcnt(0)` before the
// arrival.  The barrier: eight monotonic arrival counters (shard = blk & 7, 64 bytes apart: one word takes ~88 arrivals a
// microsecond), thread 0 adds 1 to its shard's and polls all eight until each has reached generation x shard size.  EVERY spin is
// bounded: a workgroup that never sees the others (fewer CUs available than the launch assumed) gives up after ~0.2 s, raises *err,
// and the problem is re-solved by the log-domain kernel like any guard failure - a wrong residency assumption costs time, not a hang.
// Deterministic: fixed reduction orders.
constexpr int RES_CPT = 9;
constexpr unsigned RES_SPIN_LIMIT = 1u << 15;      // ~6 us a poll under load: ~0.2 s a wait (a normal one takes one to three polls)
typedef float f4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ld_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st2_sc1(float* p, float x, float y) {               // p 8-byte aligned
    const unsigned long long v = (unsigned long long)__builtin_bit_cast(unsigned, x) | ((unsigned long long)__builtin_bit_cast(unsigned, y) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st4_sc1(float* p, const f4s v) {                  // p 16-byte aligned
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
// granule {value, tag}
__device__ __forceinline__ void st_granule(unsigned long long* g, float v, unsigned tag) {
    __hip_atomic_store(g, (unsigned long long)__builtin_bit_cast(unsigned, v) | ((unsigned long long)tag << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool ld_granule(const unsigned long long* g, unsigned tag, float& v) {
    const unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = __builtin_bit_cast(float, (unsigned)(x & 0xffffffffu));
    return (unsigned)(x >> 32) == tag;
}

// grid-wide barrier, generation gen = 1, 2, ...; returns false when it gave up (then every workgroup gives up: the counters stop)
// (wait = false: arrive only - a workgroup that reduces nothing needs nobody's partials; what it waits for is the granules)
__device__ __forceinline__ bool res_grid_sync(unsigned* cnt, unsigned gen, int nblk, int* lds_ok, bool wait) {
    __builtin_amdgcn_s_waitcnt(0);                     // this thread's sc1 stores are written through
    wg_barrier();
    if (threadIdx.x == 0) {
        const int blk = blockIdx.x;
        __hip_atomic_fetch_add(&cnt[(blk & 7) * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = wait ? 0 : 1;
        for (unsigned spin = 0; wait && spin < RES_SPIN_LIMIT; ++spin) {
            bool all = true;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const unsigned members = (unsigned)((nblk - x + 7) >> 3);          // workgroups with blk & 7 == x
                const unsigned v = __hip_atomic_load(&cnt[x * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all = all && v >= gen * members;
            }
            if (all) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        *lds_ok = ok;
    }
    wg_barrier();
    return *lds_ok != 0;
}

// Eight values a lane, each summed over the 64 lanes: lane 8 I + J returns the total of index J.  The transposed butterfly of
// lane_reduce.hpp over a lane's row-octet (4 + 2 + 1 combines) and three more levels across the octets - ten steps for eight sums
// where eight wave_sum() take forty-eight (the row dots of a 64-row block were 4.6 us of a 10 us sweep).
__device__ __forceinline__ float wave_sum8(const float (&p)[8], int lane) {
    float v = reduce8_consecutive(p, OpSum(), lane);
    v = v + dpp_f<DPP_ROW_ROR8>(v);                                   // lane ^ 8
    {
        unsigned x = __builtin_bit_cast(unsigned, v), y = x;
        lane_swap16(x, y);
        v = __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
    }
    {
        unsigned x = __builtin_bit_cast(unsigned, v), y = x;
        lane_swap32(x, y);
        v = __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
    }
    return v;
}

// Two shapes of the kernel (template):
//   <17, 9, 32>   the 4097^2 class: thread t owns columns 8 t .. 8 t + 7 and, in a ninth slot, column 4096 + t; 32-column reduce groups
//   <RB, 2, 64>   N <= 1024 (the 769^2 coarse level of 1024 x 768 images, batched: BASELINE config [3]): thread t owns columns 2 t,
//                 2 t + 1, RB = 32 or 64 rows a workgroup so that every problem of the batch has all its blocks resident at once
//                 (blockIdx.y = problem; counters, partial rows and granules per problem), 64-column reduce groups
template <int RB, int CPT, int GW>
__global__ void __launch_bounds__(ST)
stream_resident_kernel(const float* __restrict__ K, int M, int N, float* bvec, const float* __restrict__ log_mu,
                       const float* __restrict__ log_nu, float* __restrict__ avec, float* partial, int64_t partial_stride, int nblk,
                       int iters, unsigned* cnt, int* err, long long* tl) {
    static_assert((CPT == 9 && GW == 32) || (CPT == 2 && GW == 64), "the two column layouts");
    constexpr int LG = GW / 4, NSL = ST / LG, PARTS = ST / GW;     // threads per block row of a group, slices of the blocks, second-stage parts
    static_assert(NSL == 4 * PARTS, "four slices a part");
#ifdef PATS_DIAG
    long long tsum[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memrealtime();
#define RT(k) { __builtin_amdgcn_sched_barrier(0); const long long now_ = __builtin_amdgcn_s_memrealtime(); tsum[k] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define RT(k) do { } while (0)
#endif
    __shared__ float red[SW][RB];
    __shared__ float a_s[RB];
    __shared__ float red2[NSL][GW + 1];
    __shared__ int sync_ok;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, blk = blockIdx.x;
    {                                                              // this problem's slices of every array
        const int64_t pb_ = blockIdx.y;
        K += pb_ * M * N; bvec += pb_ * N; log_mu += pb_ * M; log_nu += pb_ * N; avec += pb_ * M;
        partial += pb_ * partial_stride; cnt += pb_ * 128;
    }
    const int NP = (N + 3) & ~3;                                   // row pitch of `partial`: 16-byte rows
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(partial + (size_t)nblk * NP);      // N granules behind the rows
    auto col = [&](int q) { return CPT == 9 ? (q < 8 ? 8 * t + q : 4096 + t) : 2 * t + q; };
    // ---- this thread's piece of K: rows blk RB .. -----------------------------------------------------------------------------------
    float kv[RB][CPT];
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        const int i = blk * RB + k;
        const float* Kr = K + (int64_t)i * N;
#pragma unroll
        for (int q = 0; q < CPT; ++q) kv[k][q] = (i < M && col(q) < N) ? Kr[col(q)] : 0.f;
    }
    float mu = 0.f;
    if (t < RB && blk * RB + t < M) mu = expf(log_mu[blk * RB + t]);
    // the columns this workgroup reduces: group blk of GW, four columns a thread (c4), NSL slices of the blocks (rs)
    const int ngroups = (N + GW - 1) / GW;
    const int c4 = t % LG, rs = t / LG;
    const int rj0 = blk * GW + 4 * c4;
    const bool reducer = blk < ngroups && rj0 < NP;
    float nu = 0.f;
    if (t < GW && blk < ngroups && blk * GW + t < N) nu = expf(log_nu[blk * GW + t]);
    // the granule polled first: the last valid one of the thread's consecutive columns (one reduce group holds them all)
    const int jlast = CPT == 9 ? 8 * t + 7 : 2 * t + 1;
    const int jp = jlast < N ? jlast : N - 1;
    bool alive = true;
    for (int it = 0; it < iters && alive; ++it) {
        // ---- b -> registers ---------------------------------------------------------------------------------------------------------
        float bq[CPT];
        if (it == 0) {                                             // the set-up's vector (a kernel boundary behind us)
#pragma unroll
            for (int q = 0; q < CPT; ++q) bq[q] = col(q) < N ? bvec[col(q)] : 0.f;
        } else {
            bool ok = false;
            const unsigned want = (unsigned)it;
            // One granule is polled, then all are read and checked: the thread's consecutive columns sit in ONE reduce group, whose
            // granules leave their reducer in one store instruction - a fraction of the polling traffic of every thread re-reading
            // everything (the polls share the memory side with the reducers' loads they are waiting for).
            for (unsigned spin = 0; spin < RES_SPIN_LIMIT && !ok; ++spin) {
                float probe;
                if (!ld_granule(gran + jp, want, probe)) continue;
                ok = true;
                // (8-byte agent-scope atomic loads, one granule each.  Four 16-byte `sc1` loads by inline assembly were tried here and
                //  hipcc (ROCm 7.2) MISCOMPILES the component reads behind them: `g.y == want && g.w == want` on an asm-defined
                //  float4 became ONE compare of g.x - every poll "failed" and the kernel gave up; tools/asm_vec_component_repro.hip.
                //  Whole-vector arithmetic on such a value - the reducers' sums - is translated correctly.)
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    if (col(q) < N) ok = ld_granule(gran + col(q), want, bq[q]) && ok;
                    else bq[q] = 0.f;
                }
            }
            if (!ok) alive = false;                                // (every thread still walks to the barrier: it fails there too)
        }
        RT(0);
        // ---- row dots: per-thread partials, summed over the wave eight rows at a time, then across the 8 waves through LDS ------------
#pragma unroll
        for (int g8 = 0; g8 < RB / 8; ++g8) {
            float pk[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float p = 0.f;
#pragma unroll
                for (int q = 0; q < CPT; ++q) p = fmaf(kv[8 * g8 + e][q], bq[q], p);
                pk[e] = p;
            }
            const float tot = wave_sum8(pk, lane);
            if (lane < 8) red[wave][8 * g8 + lane] = tot;
        }
#pragma unroll
        for (int k = 8 * (RB / 8); k < RB; ++k) {
            float p = 0.f;
#pragma unroll
            for (int q = 0; q < CPT; ++q) p = fmaf(kv[k][q], bq[q], p);
            p = wave_sum(p);
            if (lane == 0) red[wave][k] = p;
        }
        wg_barrier();
        if (t < RB) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < SW; ++w) s += red[w][t];
            const float a = blk * RB + t < M ? mu * __builtin_amdgcn_rcpf(s) : 0.f;
            a_s[t] = a;
            if (it + 1 == iters && blk * RB + t < M) avec[blk * RB + t] = a;
        }
        wg_barrier();
        RT(1);
        // ---- column partials of this row block ---------------------------------------------------------------------------------
        {
            float* pb = partial + (size_t)blk * NP;
            float acc[CPT];
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                acc[q] = 0.f;
#pragma unroll
                for (int k = 0; k < RB; ++k) acc[q] = fmaf(kv[k][q], a_s[k], acc[q]);
            }
            if (CPT == 9) {
                if (8 * t < NP) st4_sc1(pb + 8 * t, f4s{acc[0], acc[1], acc[2], acc[3]});
                if (8 * t + 4 < NP) st4_sc1(pb + 8 * t + 4, f4s{acc[4], acc[5], acc[6], acc[7]});
                if (4096 + t < N) st_sc1(pb + 4096 + t, acc[CPT - 1]);
            } else {
                if (2 * t < NP) st2_sc1(pb + 2 * t, acc[0], acc[1]);          // (NP is a multiple of 4: the pair is inside the row)
            }
        }
        RT(2);
        if (!res_grid_sync(cnt, (unsigned)(it + 1), nblk, &sync_ok, blk < ngroups)) alive = false;
        RT(3);
        if (!alive) break;
        // ---- b_j = nu_j / sum over the blocks, columns GW blk .. -------------------------------------------------------------------
        if (blk < ngroups) {                                       // (workgroup-uniform)
            f4s v = {0.f, 0.f, 0.f, 0.f};
            if (reducer) {
                // four block rows a thread (nblk <= 4 NSL), in ONE assembly statement with its wait: the compiler does not know that
                // these registers are written asynchronously - between a separate "issue" and "wait" it may copy them
                const float* src = partial + rj0;
                const float* p0 = src + (size_t)(rs < nblk ? rs : 0) * NP;
                const float* p1 = src + (size_t)(rs + NSL < nblk ? rs + NSL : 0) * NP;
                const float* p2 = src + (size_t)(rs + 2 * NSL < nblk ? rs + 2 * NSL : 0) * NP;
                const float* p3 = src + (size_t)(rs + 3 * NSL < nblk ? rs + 3 * NSL : 0) * NP;
                f4s x0, x1, x2, x3;
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                             "global_load_dwordx4 %1, %5, off sc1\n\t"
                             "global_load_dwordx4 %2, %6, off sc1\n\t"
                             "global_load_dwordx4 %3, %7, off sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
                const f4s z4 = {0.f, 0.f, 0.f, 0.f};
                v = (((rs < nblk ? x0 : z4) + (rs + NSL < nblk ? x1 : z4)) + (rs + 2 * NSL < nblk ? x2 : z4)) + (rs + 3 * NSL < nblk ? x3 : z4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) red2[rs][4 * c4 + e] = v[e];
            wg_barrier();
            {                                                      // NSL slices -> PARTS (four a thread) -> 1: the publishers' chain is 4 + PARTS reads
                const int cc = t % GW, part = t / GW;
                const float s4 = ((red2[4 * part][cc] + red2[4 * part + 1][cc]) + red2[4 * part + 2][cc]) + red2[4 * part + 3][cc];
                wg_barrier();
                red2[part][cc] = s4;
                wg_barrier();
            }
            if (t < GW) {
                const int j = blk * GW + t;
                float s = red2[0][t];
#pragma unroll
                for (int w = 1; w < PARTS; ++w) s += red2[w][t];
                if (j < N) {
                    const float bj = nu * __builtin_amdgcn_rcpf(s);
                    st_granule(gran + j, bj, (unsigned)(it + 1));
                    if (it + 1 == iters) st_sc1(bvec + j, bj);
                }
            }
        }
        RT(4);
    }
    if (!alive && t == 0) atomicOr(err, 1);
#ifdef PATS_DIAG
    if (tl && t == 0 && blockIdx.y == 0) for (int k = 0; k < 5; ++k) tl[blk * 6 + k] = tsum[k];
#endif
}

// ---- epilogue: guard + Z_out = ((Z + u) + v) - norm -------------------------------------------------
__global__ void __launch_bounds__(ST)
stream_guard_kernel(const float* __restrict__ a, int M, const float* __restrict__ bvec, int N,
                    int* __restrict__ fail, const int* __restrict__ err) {
    __shared__ float scratch[SW];
    const int b = blockIdx.x;
    float bad = 0.f;
    for (int i = threadIdx.x; i < M; i += ST) {
        const float x = a[(int64_t)b * M + i];
        if (!(x <= 1073741824.0f && x > 0.f)) bad = 1.f;
    }
    for (int j = threadIdx.x; j < N; j += ST) {
        const float x = bvec[(int64_t)b * N + j];
        if (!(x <= 1073741824.0f && x > 0.f)) bad = 1.f;
    }
    bad = block_allreduce(bad, OpMax(), scratch);
    if (threadIdx.x == 0) fail[b] = (bad > 0.f || (err && *err != 0)) ? 1 : 0;      // err: the resident kernel gave up at a barrier
}

__global__ void __launch_bounds__(ST)
stream_finish_kernel(SrcViewS src, int M, int N, const float* __restrict__ a,
                     const float* __restrict__ r, const float* __restrict__ bvec,
                     const float* __restrict__ c, const float* __restrict__ norm_in,
                     const int* __restrict__ fail, float* __restrict__ out) {
    const int i = blockIdx.x, b = blockIdx.y;
    if (fail[b]) return;                    // the log-domain kernel re-solves this problem
    const float* sb = src.base + (int64_t)b * src.stride;
    const float u = logf(a[(int64_t)b * M + i]) - r[(int64_t)b * M + i];
    const float norm = norm_in ? norm_in[b] : 0.f;
    float* ob = out + ((int64_t)b * M + i) * N;
    for (int j = threadIdx.x; j < N; j += ST) {
        const float v = logf(bvec[(int64_t)b * N + j]) - c[(int64_t)b * N + j];
        float z = (srcs_at(src, sb, i, j) + u) + v;
        if (norm_in) z = z - norm;
        ob[j] = z;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// A problem's share of the `partial` area in floats: ceil(M / 16) rows of N, rounded up to a multiple of FOUR floats.  The resident
// kernel puts every problem's rows (pitch NP = N rounded to 4) and its 8-byte {b_j, tag} granules at partial + problem * stride: an odd
// stride (49 x 769 for the 16 x 769^2 shape) left every odd problem's granule atomics and 16-byte sc1 loads on a 4-byte boundary,
// where a granule can straddle a cache line and tear (round-5 advice).
static inline size_t res_partial_stride(int M, int N) { return (((size_t)((M + 15) / 16) * (size_t)N) + 3) & ~(size_t)3; }

// zeroes the N granules {b_j, tag} of every problem before the resident kernel: sweep 1 accepts a granule whose tag word is 1, and
// the area is otherwise whatever the workspace held (tags of an earlier solve, any int32 array)
__global__ void stream_granule_clear_kernel(float* partial, int64_t stride, size_t gran_off, int N) {
    unsigned long long* g = reinterpret_cast<unsigned long long*>(partial + (int64_t)blockIdx.y * stride + gran_off);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) g[j] = 0ull;
}

size_t stream_workspace_bytes(int64_t batch, int M, int N) {
    size_t f = 0;
    f += al256((size_t)batch * M * N * 4);          // K
    f += al256((size_t)batch * res_partial_stride(M, N) * 4);   // partial (a problem's share rounded to 16 bytes: see res_partial_stride)
    f += 2 * al256((size_t)batch * M * 4);          // r, a
    f += 2 * al256((size_t)batch * N * 4);          // c, b
    f += al256(512 * (size_t)batch + 512);          // the resident kernel's arrival counters (8 x 64 B a problem) and its give-up flag
    return f;
}

template <int RB, int CPT>
static void launch_sweep(const float* K, int M, int N, const float* bvec, const float* log_mu, float* a,
                         float* partial, int nblk, int64_t batch, hipStream_t st) {
    hipLaunchKernelGGL((stream_sweep_kernel<RB, CPT>), dim3(nblk, (unsigned)batch), dim3(ST), 0, st, K, M, N,
                       bvec, log_mu, a, partial, nblk);
}

// Solves `batch` problems; fail[b] != 0 marks problems the caller must re-solve in the log domain.
int launch_stream(const float* base, int64_t stride, int ld, int rows, int cols, const float* alpha,
                  int64_t batch, int M, int N, const float* log_mu, const float* log_nu, const float* norm,
                  int iters, float* out, void* ws, int* fail, hipStream_t st) {
    PATS_REQUIRE(N <= ST * 9, "streaming sinkhorn: N=%d exceeds %d columns", N, ST * 9);
    PATS_REQUIRE(batch <= 65535, "streaming sinkhorn: batch too large");
    SrcViewS src{base, stride, ld, rows, cols, alpha};
    const int cpt = (N + ST - 1) / ST;
    // Rows per workgroup.  With 7+ columns per thread the K piece alone is 112+ VGPRs, so one workgroup fills a CU
    // and the grid runs in rounds of one workgroup per CU: 4097 rows in blocks of 16 are 257 workgroups - a second
    // round for ONE block on a 256-CU part.  Blocks of 17 rows (241 workgroups) finish in one.
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    auto rounds = [&](int rb) { return (batch * ((M + rb - 1) / rb) + n_cu - 1) / n_cu; };
    const bool rb17 = cpt >= 7 && rounds(17) < rounds(16);
    const int RBr = rb17 ? 17 : 16;
    const int nblk = (M + RBr - 1) / RBr;
    char* p = (char*)ws;
    float* K = (float*)p;        p += al256((size_t)batch * M * N * 4);
    float* partial = (float*)p;  p += al256((size_t)batch * res_partial_stride(M, N) * 4);  // sized for blocks of 16, 16-byte shares
    float* r = (float*)p;        p += al256((size_t)batch * M * 4);
    float* a = (float*)p;        p += al256((size_t)batch * M * 4);
    float* c = (float*)p;        p += al256((size_t)batch * N * 4);
    float* bv = (float*)p;       p += al256((size_t)batch * N * 4);
    unsigned* cnt = (unsigned*)p;
    int* err = (int*)(p + 512 * (size_t)batch);
    const dim3 rows_grid(M, (unsigned)batch), blk_grid(nblk, (unsigned)batch), col_grid((N + 63) / 64, (unsigned)batch);
    hipLaunchKernelGGL(stream_rowmax_kernel, rows_grid, dim3(ST), 0, st, src, M, N, r);
    if (rb17) hipLaunchKernelGGL((stream_colmax_partial_kernel<17>), blk_grid, dim3(ST), 0, st, src, M, N, r, partial, nblk);
    else hipLaunchKernelGGL((stream_colmax_partial_kernel<16>), blk_grid, dim3(ST), 0, st, src, M, N, r, partial, nblk);
    hipLaunchKernelGGL((stream_colreduce_kernel<0>), col_grid, dim3(1024), 0, st, partial, nblk, N, log_nu, c, bv);
    hipLaunchKernelGGL(stream_kbuild_kernel, rows_grid, dim3(ST), 0, st, src, M, N, r, c, K);
    // one problem of the 4097^2 class, every 17-row block on its own CU: all sweeps in one launch, K in registers
    // PATS_STREAM_RESIDENT=0 (production switch): never the resident kernel.  Its co-residency check knows the
    // stream's CU mask but not what OTHER streams or processes keep on the CUs: workgroups that cannot be placed make every
    // spin run into its bound (~0.2 s a solve), then the guard flag re-solves the batch in the log domain - bounded, but a latency
    // cliff for a process that shares the GPU (round-5 advice).
    static const bool resident_off = [] { const char* e = env_switch("PATS_STREAM_RESIDENT"); return e && atoi(e) == 0; }();
    const int NPr = (N + 3) & ~3;
    // the CUs THIS stream may use (ops.masked_stream / hipExtStreamCreateWithCUMask): the grid must fit on them at once
    int stream_cus = n_cu;
    {
        uint32_t mask[16] = {0};
        if (hipExtStreamGetCUMask(st, 16, mask) == hipSuccess) {
            int bits = 0;
            for (int i = 0; i < 16; ++i) bits += __builtin_popcount(mask[i]);
            if (bits > 0 && bits < stream_cus) stream_cus = bits;
        } else {
            (void)hipGetLastError();
        }
    }
    // (diagnostic library: PATS_STREAM_RESIDENT=2 launches the resident kernel whatever the stream's CUs - the give-up path on purpose)
    static const bool resident_force = [] { const char* e = diag_env("PATS_STREAM_RESIDENT"); return e && atoi(e) == 2; }();
    // which shape of stream_resident_kernel, if any: 1 = <17, 9, 32> (one problem of the 4097^2 class), 2 / 3 = <32 | 64, 2, 64>
    // (N <= 1024, every problem of the batch with all its blocks on the stream's CUs at once)
    int res_kind = 0, res_nblk = 0;
    if (!resident_off && iters > 0) {
        if (batch == 1 && rb17 && cpt == RES_CPT) { res_kind = 1; res_nblk = nblk; }
        else if (N <= 1024 && (int64_t)M * N >= 500 * 500) {
            const int64_t per = stream_cus / batch;                       // blocks a problem may have
            const int64_t need = per >= 1 ? (M + per - 1) / per : 1 << 30;
            if (need <= 32) { res_kind = 2; res_nblk = (M + 31) / 32; }
            else if (need <= 64) { res_kind = 3; res_nblk = (M + 63) / 64; }
        }
        if (res_kind) {
            const int gw = res_kind == 1 ? 32 : 64, nsl = ST / (gw / 4);
            const bool fits = (int64_t)res_nblk * batch <= stream_cus || resident_force;
            if (!(fits && res_nblk <= 4 * nsl && (N + gw - 1) / gw <= res_nblk &&
                  (size_t)res_nblk * NPr + 2 * (size_t)N <= (size_t)((M + 15) / 16) * N)) res_kind = 0;
        }
    }
    const bool resident = res_kind != 0;
#ifdef PATS_DIAG
    if (diag_env("PATS_STREAM_TRACE")) fprintf(stderr, "launch_stream: batch %lld M %d N %d cpt %d rb17 %d nblk %d stream CUs %d iters %d -> resident shape %d, %d blocks a problem\n", (long long)batch, M, N, cpt, (int)rb17, nblk, stream_cus, iters, res_kind, res_nblk);
#endif
    if (resident) {
        if (int rc = fill_bytes(cnt, 0, 512 * (size_t)batch + 256, st)) return rc;
        long long* tl = nullptr;
#ifdef PATS_DIAG
        if (diag_env("PATS_STREAM_TL")) { (void)hipMalloc((void**)&tl, (size_t)res_nblk * 6 * 8); (void)hipMemset(tl, 0, (size_t)res_nblk * 6 * 8); }
#endif
        const dim3 rgrid((unsigned)res_nblk, (unsigned)batch);
        const int64_t pstride = (int64_t)res_partial_stride(M, N);       // a problem's share of the partial area (floats, multiple of 4)
        hipLaunchKernelGGL(stream_granule_clear_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)batch), dim3(256), 0, st, partial, pstride,
                           (size_t)res_nblk * NPr, N);
        if (res_kind == 1) hipLaunchKernelGGL((stream_resident_kernel<17, 9, 32>), rgrid, dim3(ST), 0, st, K, M, N, bv, log_mu, log_nu, a, partial, pstride, res_nblk, iters, cnt, err, tl);
        else if (res_kind == 2) hipLaunchKernelGGL((stream_resident_kernel<32, 2, 64>), rgrid, dim3(ST), 0, st, K, M, N, bv, log_mu, log_nu, a, partial, pstride, res_nblk, iters, cnt, err, tl);
        else hipLaunchKernelGGL((stream_resident_kernel<64, 2, 64>), rgrid, dim3(ST), 0, st, K, M, N, bv, log_mu, log_nu, a, partial, pstride, res_nblk, iters, cnt, err, tl);
#ifdef PATS_DIAG
        if (tl) {
            (void)hipStreamSynchronize(st);
            static long long h[256 * 6];
            (void)hipMemcpy(h, tl, (size_t)res_nblk * 6 * 8, hipMemcpyDeviceToHost);
            static const char* names[5] = {"b -> registers (granule wait)", "row dots -> a", "column partials + stores issued", "grid barrier", "reduce + publish (reducers)"};
            for (int k = 0; k < 5; ++k) {
                double sum = 0, mx = 0;
                for (int w = 0; w < res_nblk; ++w) { sum += (double)h[w * 6 + k]; mx = std::max(mx, (double)h[w * 6 + k]); }
                fprintf(stderr, "  resident sweep: %-34s mean %.2f us, slowest workgroup %.2f us per sweep\n", names[k], sum / res_nblk / iters / 100.0, mx / iters / 100.0);
            }
            (void)hipFree(tl);
        }
#endif
    }
    for (int it = 0; it < (resident ? 0 : iters); ++it) {
        if (rb17) {
            switch (cpt) {
                case 7: launch_sweep<17, 7>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                case 8: launch_sweep<17, 8>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                default: launch_sweep<17, 9>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
            }
        } else {
            constexpr int RB = 16;
            switch (cpt) {
                case 1: launch_sweep<RB, 1>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                case 2: launch_sweep<RB, 2>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                case 3: launch_sweep<RB, 3>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                case 4: launch_sweep<RB, 4>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                case 5: launch_sweep<RB, 5>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                case 6: launch_sweep<RB, 6>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                case 7: launch_sweep<RB, 7>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                case 8: launch_sweep<RB, 8>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
                default: launch_sweep<RB, 9>(K, M, N, bv, log_mu, a, partial, nblk, batch, st); break;
            }
        }
        hipLaunchKernelGGL((stream_colreduce_kernel<1>), col_grid, dim3(1024), 0, st, partial, nblk, N, log_nu,
                           bv, (float*)nullptr);
    }
    if (iters == 0) {       // u = v = 0: a = exp(r), b = exp(c)  ->  Z + 0 + 0; simplest: mark for the log kernel
        (void)hipMemsetAsync(fail, 0xff, sizeof(int) * (size_t)batch, st);     // failure surfaces in check_launch
        return check_launch("streaming sinkhorn (iters == 0)");
    }
    hipLaunchKernelGGL(stream_guard_kernel, dim3((unsigned)batch), dim3(ST), 0, st, a, M, bv, N, fail, resident ? (const int*)err : (const int*)nullptr);
    hipLaunchKernelGGL(stream_finish_kernel, rows_grid, dim3(ST), 0, st, src, M, N, a, r, bv, c, norm, fail, out);
    return check_launch("streaming sinkhorn");
}

}  // namespace pats
