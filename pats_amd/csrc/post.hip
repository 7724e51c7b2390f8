// Post-OT reductions and argmax for PATS on gfx950.
//   colmass : scales = sqrt(exp(Z[:, :-1, :-1]).sum(1) + 1e-8)          models/first_layer.py:117-118
//   bias    : Z[:, :, -1] += log(k); Z[:, -1, :] += log(k)              models/second_layer.py:107-112
//   exp     : torch.exp(scores_origin)                                  models/third_layer.py:159
//   argmax  : scores.max(2).indices, scores.max(1).indices (first index on ties)
//                                                   first_layer.py:162  second_layer.py:243
// All HBM-bound single-pass kernels; column-direction work maps lanes to consecutive columns so
// every wave load is one contiguous segment.
#include "common.hpp"

namespace pats {

// out (optional): sqrt(column mass + 1e-8); col_nomatch (optional): scores.max(1).indices == M - 1, i.e. the
// dustbin row strictly above every real row (first index wins ties)   first_layer.py:117-118,163,167
// 64 columns x 4 row slices per workgroup (round 6): one thread per column walked all M rows alone - 27 us for ONE 301 x 301 plan,
// a fixed cost of every pair's coarse level and again of its fine level's flags.  Slice q sums rows q, q + 4, ... (two running sums
// as before), the four partials meet in LDS and are added in a fixed order; the exponentials are skipped when only the flags are
// asked for (the fine level's call).
__global__ void __launch_bounds__(256)
colmass_kernel(const float* __restrict__ Z, int M, int N, float* __restrict__ out, uint8_t* __restrict__ col_nomatch,
               const int* __restrict__ only_if) {
    __shared__ float ps[4][64], pm[4][64];
    const int64_t b = blockIdx.y;
    if (only_if && !only_if[b]) return;                 // (workgroup-uniform)
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cl;
    const bool live = j < N - 1;
    const float* z = Z + b * (int64_t)M * N + (live ? j : 0);
    float s0 = 0.f, s1 = 0.f, mx = -INFINITY;
    if (live) {
        int i = q;
        if (out) {
            for (; i + 4 < M - 1; i += 8) {
                const float x0 = z[(int64_t)i * N], x1 = z[(int64_t)(i + 4) * N];
                s0 += expf(x0);
                s1 += expf(x1);
                mx = fmaxf(mx, fmaxf(x0, x1));
            }
            if (i < M - 1) {
                const float x0 = z[(int64_t)i * N];
                s0 += expf(x0);
                mx = fmaxf(mx, x0);
            }
        } else {
            for (; i + 4 < M - 1; i += 8) mx = fmaxf(mx, fmaxf(z[(int64_t)i * N], z[(int64_t)(i + 4) * N]));
            if (i < M - 1) mx = fmaxf(mx, z[(int64_t)i * N]);
        }
    }
    ps[q][cl] = s0 + s1;
    pm[q][cl] = mx;
    wg_barrier();
    if (q != 0 || !live) return;
    const float tot = (ps[0][cl] + ps[1][cl]) + (ps[2][cl] + ps[3][cl]);
    mx = fmaxf(fmaxf(pm[0][cl], pm[1][cl]), fmaxf(pm[2][cl], pm[3][cl]));
    if (out) out[b * (N - 1) + j] = sqrtf(tot + 1e-8f);
    if (col_nomatch) col_nomatch[b * (N - 1) + j] = z[(int64_t)(M - 1) * N] > mx;
}

__global__ void __launch_bounds__(256)
bias_kernel(float* __restrict__ Z, int M, int N, float lb) {
    const int64_t b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    float* z = Z + b * (int64_t)M * N;
    // in-place adds of second_layer.py:108-109: column first, then row; the corner gets both
    if (t < M - 1) z[(int64_t)t * N + (N - 1)] += lb;
    if (t < N - 1) z[(int64_t)(M - 1) * N + t] += lb;
    if (t == 0) {
        float c = z[(int64_t)(M - 1) * N + (N - 1)];
        c += lb;
        c += lb;
        z[(int64_t)(M - 1) * N + (N - 1)] = c;
    }
}

__global__ void __launch_bounds__(256)
exp_kernel(const float* __restrict__ in, int64_t count, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < count; i += stride) out[i] = expf(in[i]);
}

// one wave per row
__global__ void __launch_bounds__(256)
argmax_rows_kernel(const float* __restrict__ Z, int64_t rows, int N, int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* z = Z + row * (int64_t)N;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < N; j += 64) {
        const float x = z[j];
        if (x > bv || bi == 0x7fffffff) { bv = x; bi = j; }
    }
    wave_argmax(bv, bi);
    if (lane == 0) out[row] = bi;
}

// one thread per column
__global__ void __launch_bounds__(256)
argmax_cols_kernel(const float* __restrict__ Z, int M, int N, int64_t* __restrict__ out) {
    const int64_t b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float* z = Z + b * (int64_t)M * N + j;
    float bv = z[0];
    int bi = 0;
    for (int i = 1; i < M; ++i) {
        const float x = z[(int64_t)i * N];
        if (x > bv) { bv = x; bi = i; }
    }
    out[b * N + j] = bi;
}

}  // namespace pats

using namespace pats;

extern "C" int pats_colmass_sqrt_f32(const float* Z, int64_t batch, int M, int N, float* out,
                                     pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && M > 1 && N > 1, "colmass: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(Z && out, "colmass: null pointer");
    PATS_REQUIRE(batch <= 65535, "colmass: batch too large");
    hipLaunchKernelGGL(colmass_kernel, dim3((unsigned)ceil_div(N - 1, 64), (unsigned)batch),
                       dim3(256), 0, as_stream(stream), Z, M, N, out, (uint8_t*)nullptr, (const int*)nullptr);
    return check_launch("colmass_kernel");
}

extern "C" int pats_colmass_flags_f32(const float* Z, int64_t batch, int M, int N, float* out, uint8_t* col_nomatch,
                                      pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && M > 1 && N > 1, "colmass_flags: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(Z && (out || col_nomatch), "colmass_flags: null pointer");
    PATS_REQUIRE(batch <= 65535, "colmass_flags: batch too large");
    hipLaunchKernelGGL(colmass_kernel, dim3((unsigned)ceil_div(N - 1, 64), (unsigned)batch),
                       dim3(256), 0, as_stream(stream), Z, M, N, out, col_nomatch, (const int*)nullptr);
    return check_launch("colmass_kernel");
}

namespace pats {
// column flags of the problems with only_if[b] != 0 (all when only_if is null): the OT entry points use it for the
// shapes / problems whose Sinkhorn kernel does not emit the flags from its own epilogue
int launch_col_flags(const float* Z, int64_t batch, int M, int N, uint8_t* col_nomatch, const int* only_if, hipStream_t st) {
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
        hipLaunchKernelGGL(colmass_kernel, dim3((unsigned)ceil_div(N - 1, 64), (unsigned)nb), dim3(256), 0, st,
                           Z + b0 * (int64_t)M * N, M, N, (float*)nullptr, col_nomatch + b0 * (N - 1),
                           only_if ? only_if + b0 : nullptr);
    }
    return check_launch("colmass_kernel(flags)");
}
}  // namespace pats

extern "C" int pats_dustbin_bias_inplace_f32(float* Z, int64_t batch, int M, int N, float k,
                                             pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && M > 0 && N > 0 && k > 0.f, "dustbin_bias: bad argument");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(Z, "dustbin_bias: null pointer");
    PATS_REQUIRE(batch <= 65535, "dustbin_bias: batch too large");
    const int len = M > N ? M : N;
    hipLaunchKernelGGL(bias_kernel, dim3((unsigned)ceil_div(len, 256), (unsigned)batch), dim3(256),
                       0, as_stream(stream), Z, M, N, logf(1.0f * k));
    return check_launch("bias_kernel");
}

extern "C" int pats_exp_f32(const float* Z, int64_t count, float* out, pats_stream_t stream) {
    PATS_REQUIRE(count >= 0, "exp: bad count");
    if (count == 0) return PATS_OK;
    PATS_REQUIRE(Z && out, "exp: null pointer");
    const int64_t blocks = ceil_div(count, 256);
    hipLaunchKernelGGL(exp_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       as_stream(stream), Z, count, out);
    return check_launch("exp_kernel");
}

extern "C" int pats_argmax_f32(const float* Z, int64_t batch, int M, int N, int64_t* row_arg,
                               int64_t* col_arg, pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && M > 0 && N > 0, "argmax: bad shape");
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(Z, "argmax: null pointer");
    PATS_REQUIRE(batch <= 65535, "argmax: batch too large");
    if (row_arg) {
        const int64_t rows = batch * M;
        hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0,
                           as_stream(stream), Z, rows, N, row_arg);
        int rc = check_launch("argmax_rows_kernel");
        if (rc) return rc;
    }
    if (col_arg) {
        hipLaunchKernelGGL(argmax_cols_kernel, dim3((unsigned)ceil_div(N, 256), (unsigned)batch),
                           dim3(256), 0, as_stream(stream), Z, M, N, col_arg);
        return check_launch("argmax_cols_kernel");
    }
    return PATS_OK;
}
