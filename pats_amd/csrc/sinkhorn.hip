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
    __syncthreads();
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
// 65 x 65, one wave per problem
// ------------------------------------------------------------------------------------------
constexpr int NB = 64;          // real rows / columns
constexpr int NT = NB + 1;      // + dustbin
constexpr int TILE = NT * NT;   // 4225

struct Wave65Lds {
    float tile[TILE + 3];       // staging for the load-time transpose
    float us[NT + 3];           // broadcast buffers for the duals
    float vs[NT + 3];
};

__device__ __forceinline__ float lse_finish(float s, float mI) {
    return (fast_log2(s) + mI) * LN2;
}

// mode 0: log_mu/log_nu given (a6)      mode 2: ns given, log_optimal_transport2 marginals (a5)
template <int MODE>
__global__ void __launch_bounds__(64)
sinkhorn65_kernel(const float* __restrict__ Zin, int64_t P, const float* __restrict__ log_mu_in,
                  const float* __restrict__ log_nu_in, const float* __restrict__ ns,
                  const float* __restrict__ one, int iters, float bias_k,
                  float* __restrict__ out) {
    __shared__ Wave65Lds lds;
    const int lane = threadIdx.x;
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    const float* Zp = Zin + p * TILE;

    // ---- coalesced load -> LDS -> row-per-lane and column-per-lane register images ----------
#pragma unroll 11
    for (int k = 0; k < 66; ++k) lds.tile[k * 64 + lane] = Zp[k * 64 + lane];
    if (lane == 0) lds.tile[TILE - 1] = Zp[TILE - 1];
    __syncthreads();
    float zr[NB], zc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) zr[j] = lds.tile[lane * NT + j];   // stride 65: conflict-free
#pragma unroll
    for (int i = 0; i < NB; ++i) zc[i] = lds.tile[i * NT + lane];
    const float zr64 = lds.tile[lane * NT + NB];    // Z[lane][64]   (dustbin column)
    const float zc64 = lds.tile[NB * NT + lane];    // Z[64][lane]   (dustbin row)
    const float corner = lds.tile[TILE - 1];        // Z[64][64]

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

    float u = 0.f, u64 = 0.f, v = 0.f, v64 = 0.f;
    for (int it = 0; it < iters; ++it) {
        // ---- u = log_mu - lse_j(Z + v) ------------------------------------------------------
        __syncthreads();
        lds.vs[lane] = v;
        if (lane == 0) lds.vs[NB] = v64;
        __syncthreads();
        {
            float m = zr64 + v64;
#pragma unroll
            for (int j = 0; j < NB; ++j) m = fmaxf(m, zr[j] + lds.vs[j]);
            const float mI = ceilf(m * LOG2E);
            float s = fast_exp2(fmaf(zr64 + v64, LOG2E, -mI));
#pragma unroll
            for (int j = 0; j < NB; ++j) s += fast_exp2(fmaf(zr[j] + lds.vs[j], LOG2E, -mI));
            u = lmu - lse_finish(s, mI);
            // dustbin row: elements live one per lane
            const float t = zc64 + v, tc = corner + v64;
            const float mI2 = ceilf(fmaxf(wave_max(t), tc) * LOG2E);
            const float s2 = wave_sum(fast_exp2(fmaf(t, LOG2E, -mI2))) + fast_exp2(fmaf(tc, LOG2E, -mI2));
            u64 = lmu64 - lse_finish(s2, mI2);
        }
        // ---- v = log_nu - lse_i(Z + u) ------------------------------------------------------
        __syncthreads();
        lds.us[lane] = u;
        if (lane == 0) lds.us[NB] = u64;
        __syncthreads();
        {
            float m = zc64 + u64;
#pragma unroll
            for (int i = 0; i < NB; ++i) m = fmaxf(m, zc[i] + lds.us[i]);
            const float mI = ceilf(m * LOG2E);
            float s = fast_exp2(fmaf(zc64 + u64, LOG2E, -mI));
#pragma unroll
            for (int i = 0; i < NB; ++i) s += fast_exp2(fmaf(zc[i] + lds.us[i], LOG2E, -mI));
            v = lnu - lse_finish(s, mI);
            const float t = zr64 + u, tc = corner + u64;
            const float mI2 = ceilf(fmaxf(wave_max(t), tc) * LOG2E);
            const float s2 = wave_sum(fast_exp2(fmaf(t, LOG2E, -mI2))) + fast_exp2(fmaf(tc, LOG2E, -mI2));
            v64 = lnu64 - lse_finish(s2, mI2);
        }
    }

    // ---- Z + u + v - norm (+ the caller's dustbin bias), rows written coalesced ---------------
    __syncthreads();
    lds.us[lane] = u;
    if (lane == 0) lds.us[NB] = u64;
    __syncthreads();
    const float lb = bias_k > 0.f ? logf(bias_k) : 0.f;
    float* Op = out + p * TILE;
#pragma unroll
    for (int i = 0; i < NB; ++i) {   // full unroll: zc[] must stay in registers (static indices)
        Op[i * NT + lane] = ((zc[i] + lds.us[i]) + v) - norm;
    }
    {   // dustbin column entries (i, 64) and dustbin row entries (64, j)
        float c = ((zr64 + u) + v64) - norm;
        float r = ((zc64 + u64) + v) - norm;
        if (bias_k > 0.f) { c += lb; r += lb; }
        Op[lane * NT + NB] = c;
        Op[NB * NT + lane] = r;
        if (lane == 0) {
            float q = ((corner + u64) + v64) - norm;
            if (bias_k > 0.f) { q += lb; q += lb; }     // corner receives both in-place adds
            Op[TILE - 1] = q;
        }
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
                                   float* __restrict__ wsT) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
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
    __threadfence_block();
    __syncthreads();

    const float* lmu = log_mu + (int64_t)b * M;
    const float* lnu = log_nu + (int64_t)b * N;
    for (int it = 0; it < iters; ++it) {
        for (int i = wave; i < M; i += nwave) {
            const float l = wave_lse(Zr + (int64_t)i * N, v, N, lane);
            if (lane == 0) u[i] = lmu[i] - l;
        }
        __syncthreads();
        for (int j = wave; j < N; j += nwave) {
            const float l = wave_lse(Zt + (int64_t)j * M, u, M, lane);
            if (lane == 0) v[j] = lnu[j] - l;
        }
        __syncthreads();
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
// host side
// ------------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct OtWorkspace {
    float *log_mu, *log_nu, *norm, *Zw, *Zt;
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
    if (zw) w.Zw = take((size_t)batch * M * N);
    w.Zt = take((size_t)batch * M * N);
    w.bytes = off;
    return w;
}

static int launch_wg(const SrcView& src, int64_t batch, int M, int N, const float* log_mu,
                     const float* log_nu, const float* norm, int iters, float bias_k, float* out,
                     float* Zw, float* Zt, hipStream_t st) {
    const int threads = ((int64_t)M * N >= 128 * 128) ? 1024 : 256;
    const size_t lds = (size_t)(M + N) * sizeof(float);
    PATS_REQUIRE(lds <= 64 * 1024, "sinkhorn: M+N=%d too large for the one-workgroup kernel", M + N);
    hipLaunchKernelGGL(sinkhorn_wg_kernel, dim3((unsigned)batch), dim3(threads), lds, st, src, M, N,
                       log_mu, log_nu, norm, iters, bias_k, out, Zw, Zt);
    return check_launch("sinkhorn_wg_kernel");
}

}  // namespace pats

using namespace pats;

extern "C" size_t pats_sinkhorn_workspace_bytes(int64_t batch, int M, int N) {
    if (M == NT && N == NT) return 0;
    return carve(nullptr, batch, M, N, false, false).bytes;
}

extern "C" size_t pats_ot_workspace_bytes(int64_t batch, int M, int N) {
    return carve(nullptr, batch, M, N, true, true).bytes;
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
        hipLaunchKernelGGL(sinkhorn65_kernel<0>, dim3((unsigned)batch), dim3(64), 0, st, Z, batch,
                           log_mu, log_nu, nullptr, nullptr, iters, 0.f, out);
        return check_launch("sinkhorn65_kernel<0>");
    }
    PATS_REQUIRE(workspace && workspace_bytes >= pats_sinkhorn_workspace_bytes(batch, M, N),
                 "sinkhorn: workspace too small");
    OtWorkspace w = carve(workspace, batch, M, N, false, false);
    SrcView src{Z, (int64_t)M * N, N, M, N, nullptr};
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
    return launch_wg(src, batch, M, N, w.log_mu, w.log_nu, w.norm, iters, 0.f, Z, w.Zw, w.Zt, st);
}

extern "C" int pats_log_optimal_transport2_f32(const float* scores, int64_t batch, int m, int n,
                                               const float* one, const float* ns, int iters,
                                               float bias_k, float* Z, void* workspace,
                                               size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && m > 1 && n > 1 && iters >= 0, "log_optimal_transport2: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(scores && ns && Z, "log_optimal_transport2: null pointer");
    hipStream_t st = as_stream(stream);
    if (m == NT && n == NT) {
        hipLaunchKernelGGL(sinkhorn65_kernel<2>, dim3((unsigned)batch), dim3(64), 0, st, scores,
                           batch, nullptr, nullptr, ns, one, iters, bias_k, Z);
        return check_launch("sinkhorn65_kernel<2>");
    }
    PATS_REQUIRE(workspace && workspace_bytes >= pats_ot_workspace_bytes(batch, m, n),
                 "log_optimal_transport2: workspace too small");
    OtWorkspace w = carve(workspace, batch, m, n, true, true);
    hipLaunchKernelGGL(ot_prep_kernel, dim3((unsigned)batch), dim3(256), 0, st, ns, n - 1, m, n,
                       (float)(m - 1), one, w.log_mu, w.log_nu, w.norm);
    int rc = check_launch("ot_prep_kernel");
    if (rc) return rc;
    SrcView src{scores, (int64_t)m * n, n, m, n, nullptr};
    return launch_wg(src, batch, m, n, w.log_mu, w.log_nu, w.norm, iters, bias_k, Z, nullptr, w.Zt, st);
}
