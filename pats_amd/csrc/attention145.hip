// attention(query, key, value) at the FINE level's shape (round 4): up to 160 tokens per side, up to 80 channels per head
// (models/modules.py:84-88; second_layer.py:89-91 runs it on [264, 145] descriptors, 4 heads of 66 channels).
//
// The general kernel (attention.hip) multiplies on the fp32 matrix pipe with one operand pair per MFMA straight from HBM / an LDS
// slab: 3.3 ms per 4 096 problems, 27 TF/s.  Here one 640-thread workgroup owns a (problem, head) and both products run as
// fp16-split three-pass v_mfma_f32_16x16x32_f16 (the contraction of the cost build, csrc/mfma_tile.hpp) with the scores in
// REGISTERS from the first product to the second:
//   1. K is staged once into LDS, split (hi + lo fp16 of 2^6 k) in MFMA fragment order with the KEYS as rows (60 KB: 96 channels
//      - 66 and zeros - x 160 keys); wave w's 16 queries come straight from HBM as the B operand;
//   2. S^T = K^T Q: wave w holds the scores of its 16 queries against all 160 keys in 40 accumulator registers - lane
//      (q' = lane >> 4, j = lane & 15) has keys 16 t + 4 q' + r of query 16 w + j: the softmax over the keys is in-lane plus two
//      exchanges (lanes 16 / 32 apart); padded keys are masked to -inf;
//   3. the unnormalised probabilities never move: accumulator registers of key tiles 2 kk and 2 kk + 1 ARE (split again) the B
//      operand of k-step kk of out^T = V P^T - V is staged (over K's LDS, behind a barrier) with its keys in exactly that order
//      (slot (q', e): key 16 (2 kk + (e >> 2)) + 4 q' + (e & 3));
//   4. out^T / sum leaves through LDS as whole rows of queries.
// 60 KB of LDS and 96 registers: two workgroups (20 waves) per CU - one stages or stores while the other multiplies.
// Range: |q|, |k|, |v| < 1023; a non-finite output raises *flag and the general kernel, queued behind and gated on it, redoes the launch.
#include "common.hpp"

namespace pats {

namespace {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr float PRE = 64.0f, UNS = 1.0f / 4096.0f;
constexpr int A_T = 10;                    // 16-token tiles per side (160 tokens)
constexpr int A_KS = 3;                    // 32-channel k-steps of the first product (96 channels)
constexpr int A_DT = 5;                    // 16-channel row tiles of the second product (80 channels)
constexpr int A_KK = A_T / 2;              // its k-steps: pairs of key tiles
constexpr int A_FRAG = 1024;               // one fragment: 64 lanes x 8 halves
constexpr int A_OSTR = 164;                // floats per channel row of the output tile
constexpr int A_LDS = A_KS * 2 * A_T * A_FRAG;            // K: 61 440 B; V (51 200) and the output tile (52 480) overlay it
static_assert(A_LDS >= 2 * A_DT * A_KK * A_FRAG && A_LDS >= 16 * A_DT * A_OSTR * 4, "overlays fit");

struct A145Args {
    const float* q; const float* k; const float* v; float* out;
    int dim, heads, n, m;
    int64_t items, per_xcd;                // (problem, head) pairs, and how many each of the 8 XCDs takes
    float c;                               // 2^-12 / sqrt(dim) * log2(e): accumulator -> exponent of 2
    int* flag;                             // raised if an output is not finite
    const int* gate;                       // optional: no-op unless *gate != 0
};

__device__ __forceinline__ void split4(const f4v v, h4v& hi, h4v& lo) {
    const f4v s = v * PRE;
    hi = __builtin_convertvector(s, h4v);
    lo = __builtin_convertvector(s - __builtin_convertvector(hi, f4v), h4v);
}
__device__ __forceinline__ void split8(const f4v a, const f4v b, h8v& hi, h8v& lo) {
    h4v ah, al, bh, bl;
    split4(a, ah, al);
    split4(b, bh, bl);
    hi = h8v{ah.x, ah.y, ah.z, ah.w, bh.x, bh.y, bh.z, bh.w};
    lo = h8v{al.x, al.y, al.z, al.w, bl.x, bl.y, bl.z, bl.w};
}
__device__ __forceinline__ f4v mfma3(const h8v ah, const h8v al, const h8v bh, const h8v bl, f4v c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);       // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
}

}  // namespace

__global__ void __launch_bounds__(640, 5)
attention145_kernel(A145Args g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (g.gate && *g.gate == 0) return;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int qp = lane >> 4, j = lane & 15;
    // workgroup i runs on XCD i % 8: every XCD takes one contiguous range of (problem, head) - the heads of a problem interleave
    // their 145-float rows, i.e. share 128-byte lines
    const int64_t lid = (int64_t)(blockIdx.x & 7) * g.per_xcd + (blockIdx.x >> 3);
    if (lid >= g.items) return;
    const int64_t bi = lid / g.heads;
    const int h = (int)(lid - bi * g.heads);
    const int n = g.n, m = g.m, dim = g.dim;
    const int rsn = g.heads * n, rsm = g.heads * m;                       // channel strides
    const float* Q = g.q + (bi * dim * g.heads + h) * (int64_t)n;
    const float* K = g.k + (bi * dim * g.heads + h) * (int64_t)m;
    const float* V = g.v + (bi * dim * g.heads + h) * (int64_t)m;
    float* O = g.out + (bi * dim * g.heads + h) * (int64_t)n;

    // ---- 1. K -> LDS: wave-items (8-channel group: 12, chunk of 64 keys: 3), lane = key --------------------------------------------
    {
        f4v ka[4], kb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = wave + 10 * i, cg = item / 3, key = (item - 3 * cg) * 64 + lane;
            const bool ok = item < 36 && key < m;
            const float* p = K + (int64_t)(cg * 8) * rsm + key;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (ok && cg * 8 + e < dim) ? p[e * rsm] : 0.f;
            ka[i] = f4v{x[0], x[1], x[2], x[3]};
            kb[i] = f4v{x[4], x[5], x[6], x[7]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = wave + 10 * i, cg = item / 3, key = (item - 3 * cg) * 64 + lane;
            if (item < 36 && key < 16 * A_T) {
                h8v hi, lo;
                split8(ka[i], kb[i], hi, lo);
                char* d = lds + (((cg >> 2) * 2) * A_T + (key >> 4)) * A_FRAG + ((cg & 3) * 16 + (key & 15)) * 16;
                *reinterpret_cast<h8v*>(d) = hi;
                *reinterpret_cast<h8v*>(d + A_T * A_FRAG) = lo;
            }
        }
    }
    // this wave's queries as the B operand: lane (q', j) holds channels 32 ks + 8 q' + e of query 16 wave + j
    h8v qh[A_KS], ql[A_KS];
    {
        const int query = 16 * wave + j;
        const bool qok = query < n;
        const float* p = Q + (qok ? query : 0);
#pragma unroll
        for (int ks = 0; ks < A_KS; ++ks) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = 32 * ks + 8 * qp + e;
                x[e] = (qok && ch < dim) ? p[(int64_t)ch * rsn] : 0.f;
            }
            split8(f4v{x[0], x[1], x[2], x[3]}, f4v{x[4], x[5], x[6], x[7]}, qh[ks], ql[ks]);
        }
    }
    wg_barrier();

    // ---- 2. S^T = K^T Q, softmax over the keys in registers ---------------------------------------------------------------------------
    f4v s[A_T];
#pragma unroll
    for (int kt = 0; kt < A_T; ++kt) {
        f4v c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < A_KS; ++ks) {
            const char* f = lds + ((ks * 2) * A_T + kt) * A_FRAG + lane * 16;
            const h8v kh = *reinterpret_cast<const h8v*>(f), kl = *reinterpret_cast<const h8v*>(f + A_T * A_FRAG);
            c = mfma3(kh, kl, qh[ks], ql[ks], c);
        }
        s[kt] = c;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < A_T; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = 16 * kt + 4 * qp + r < m ? s[kt][r] * g.c : -INFINITY;
            s[kt][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float den = 0.f;
#pragma unroll
    for (int kt = 0; kt < A_T; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = fast_exp2(s[kt][r] - mx);                      // exp2(-inf) = 0 for the padded keys
            s[kt][r] = e;
            den += e;
        }
    den += __shfl_xor(den, 16);
    den += __shfl_xor(den, 32);
    const float inv = UNS / den;
    // the probabilities as the B operand of the second product: k-step kk = key tiles 2 kk, 2 kk + 1
    h8v ph[A_KK], pl[A_KK];
#pragma unroll
    for (int kk = 0; kk < A_KK; ++kk) split8(s[2 * kk], s[2 * kk + 1], ph[kk], pl[kk]);
    wg_barrier();                                                           // every wave has read K

    // ---- 3. V -> LDS over K: wave-items (row tile: 5, k-step: 5), lane (q', channel) loads its two key quads -----------------------
    {
        f4v va[3], vb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int item = wave + 10 * i, dt = item / A_KK, kk = item - dt * A_KK;
            const int d = 16 * dt + j, k0 = 32 * kk + 4 * qp, k1 = k0 + 16;
            const bool ok = item < A_DT * A_KK && d < dim;
            const float* p = V + (int64_t)d * rsm;
            if (ok && k0 + 3 < m) va[i] = *reinterpret_cast<const f4u*>(p + k0);
            else va[i] = f4v{ok && k0 < m ? p[k0] : 0.f, ok && k0 + 1 < m ? p[k0 + 1] : 0.f, ok && k0 + 2 < m ? p[k0 + 2] : 0.f, 0.f};
            if (ok && k1 + 3 < m) vb[i] = *reinterpret_cast<const f4u*>(p + k1);
            else vb[i] = f4v{ok && k1 < m ? p[k1] : 0.f, ok && k1 + 1 < m ? p[k1 + 1] : 0.f, ok && k1 + 2 < m ? p[k1 + 2] : 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int item = wave + 10 * i, dt = item / A_KK, kk = item - dt * A_KK;
            if (item < A_DT * A_KK) {
                h8v hi, lo;
                split8(va[i], vb[i], hi, lo);
                char* d = lds + (dt * A_KK + kk) * A_FRAG + lane * 16;
                *reinterpret_cast<h8v*>(d) = hi;
                *reinterpret_cast<h8v*>(d + A_DT * A_KK * A_FRAG) = lo;
            }
        }
    }
    wg_barrier();

    // ---- 4. out^T = V P^T ---------------------------------------------------------------------------------------------------------------
    f4v o[A_DT];
#pragma unroll
    for (int dt = 0; dt < A_DT; ++dt) o[dt] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < A_KK; ++kk)
#pragma unroll
        for (int dt = 0; dt < A_DT; ++dt) {
            const char* f = lds + (dt * A_KK + kk) * A_FRAG + lane * 16;
            const h8v vh = *reinterpret_cast<const h8v*>(f), vl = *reinterpret_cast<const h8v*>(f + A_DT * A_KK * A_FRAG);
            o[dt] = mfma3(vh, vl, ph[kk], pl[kk], o[dt]);
        }
    wg_barrier();                                                           // every wave has read V
    float* ot = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int dt = 0; dt < A_DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[(16 * dt + 4 * qp + r) * A_OSTR + 16 * wave + j] = o[dt][r] * inv;
    wg_barrier();
    bool bad = false;
    for (int d = wave; d < dim; d += 10) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int query = 64 * c + lane;
            if (query < n) {
                const float v = ot[d * A_OSTR + query];
                bad |= !(fabsf(v) <= 3.0e38f);
                O[(int64_t)d * rsn + query] = v;
            }
        }
    }
    if (__any(bad) && lane == 0 && g.flag) atomicOr(g.flag, 1);
}

// 1 = launched; PATS_ERR_UNSUPPORTED = not this kernel's shape (or the device refused the LDS): the caller takes the general kernel.
// flag: one int, zero on entry; the caller queues the general kernel behind, gated on it.
int launch_attention145(const float* query, const float* key, const float* value, int64_t batch, int dim, int heads, int n, int m,
                        float* out, int* flag, const int* gate, hipStream_t st) {
    static const bool off = [] { const char* e = diag_env("PATS_ATTN145"); return e && atoi(e) == 0; }();     // A/B switch
    if (off || n > 16 * A_T || m > 16 * A_T || n <= 96 || m <= 96 || dim > 16 * A_DT || dim <= 32 || !flag) return PATS_ERR_UNSUPPORTED;
    if (batch * heads + 8 >= (1ll << 31)) return PATS_ERR_UNSUPPORTED;
    static int state[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
    if (state[dev] == 0) {
        const bool ok = hipFuncSetAttribute((const void*)attention145_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        state[dev] = ok ? 1 : -1;
    }
    if (state[dev] != 1) return PATS_ERR_UNSUPPORTED;
    const int64_t items = batch * heads, per_xcd = (items + 7) / 8;
    A145Args g{query, key, value, out, dim, heads, n, m, items, per_xcd, (float)((double)UNS / sqrt((double)dim) * 1.4426950408889634), flag, gate};
    hipLaunchKernelGGL(attention145_kernel, dim3((unsigned)(8 * per_xcd)), dim3(640), A_LDS, st, g);
    return check_launch("attention145_kernel");
}

}  // namespace pats
