// Conv1d(kernel_size = 1) of the GNN layers with PACKED weights (round 4): y[b] = W . x[b] (+ bias, + residual) over the flattened
// (batch, token) axis, W handed over pre-split (fp16 hi + lo of 2^6 w) in MFMA fragment order (pats_propagation_pack_f32, once per
// layer).
//
// Why beside conv_lean_kernel (gnn.hip): at the fine level's shape (264 / 528 channels, 145 tokens) that kernel spends two thirds of
// its loads re-fetching the fp32 weights for every 64-column tile, splits them again each time and keeps one 16-channel chunk of
// operands in flight per workgroup (matrix pipe ~10 % busy, a third of the HBM rate).  Here
//   * a 512-thread workgroup owns 64 columns x up to 272 output rows (wave w: row tiles w, w + 8 of 16 rows each, all four
//     16-column tiles; a 17th row tile - 264 rows are 16 tiles and a half - has its four column tiles on waves 0..3);
//   * the activations of its 64 columns are staged WHOLE, up to 288 channels at a time (one "pass": 72 KB of LDS, split into fp16
//     hi + lo in MFMA fragment order on the way, the BatchNorm affine + ReLU of the producing layer applied there, (x | message)
//     read from two tensors without a cat): every load of a pass is in flight at once, and the k loop behind it has no barrier;
//   * the weights are the A operand straight from L2 into registers - one 16-byte load per lane and fragment, a ring of two
//     32-channel k-steps, no LDS staging, no VALU split (the fused layer's scheme, csrc/gnn_fused.hip);
//   * two workgroups share a CU (128 registers, 2 x 72 KB): one loads or stores while the other multiplies;
//   * the result leaves through LDS: rows of 64 consecutive columns per wave store instead of the MFMA layout's 16;
//   * v_mfma_f32_16x16x32_f16, three exact-product passes, fp32 accumulation - the contraction of the cost build.
// Measured and not kept (round 4): 16-byte lane loads of four consecutive columns turned in LDS or by 4 x 4 DPP transposes (rows of
// 145 floats: three quads in four are misaligned and pass the address unit no faster than four dwords - 510 / 680 us against 400
// per [264 -> 264] product); a persistent grid that fetches the next tile under the epilogue (vmcnt is in order: the stash's wait for
// those loads waits for the epilogue's stores too - 470 us); LOADER waves (one persistent 768-thread workgroup per CU, waves 8..11 fill
// the other of two 72 KB stages with the next tile - 72 dword loads a lane in flight - while waves 0..7 multiply and store from
// registers, one barrier per tile: 590-600 us, the same as ONE conv_pk workgroup per CU - four loader waves do not keep the vector
// memory pipe as full as the eight of a second workgroup do, and the register stores of 16 columns a row are slower than the
// LDS-turned rows of 64).  A second attempt at the end of the round (eight multiplier + four loader waves with separate code paths,
// no spills, weights requested ahead of the unit's barrier; profiles/r04_conv_ps_ab.txt) ran 506 us and its knock-outs say why: without
// the loaders' global loads 352, without the stores 368, without either 256, and without the MFMAs no faster at all - the 256 us that
// remain are the WEIGHT STREAM: 360 KB of fragments per 64-column tile through the CU's vector L1 at ~24 bytes a clock (7.1 us a tile),
// which is also conv_pk_kernel's own skeleton (251 us).  Overlapping activations with MFMAs cannot go below it; halving it takes
// weights that stay on the CU (a weights-stationary tile: 128 output rows x 264 channels = 135 KB of LDS) or 128-column tiles.
// (Non-temporal weight loads - past the L1 - cost 45 %: 395 -> 570 us; the two workgroups of a CU share half of that stream in it.)  Timeline of a workgroup (s_memrealtime, mean of 9 280): 5.5 us issuing its
// 40 dword loads per lane, 1.8 stash, 6.3 k loop (two workgroups share the matrix pipe), 3.3 at barriers, 2.5 epilogue: the vector
// memory pipe (activations in and out at 4 bytes a lane, 360 KB of weights per tile) is what bounds it, not HBM and not the MFMAs.
// Range: |activation| < 1023; a non-finite output raises *redo - the layer's one flag: the round-2 composition queued behind the
// layer, gated on it, recomputes the whole layer (gnn.hip, propagation_impl).
#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#ifdef PATS_DIAG
#include <cstdio>
#include <vector>
#endif

namespace pats {

namespace {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef const __attribute__((address_space(1))) h8v* gptr_h8;

constexpr float PRE = 64.0f, UNS = 1.0f / 4096.0f;
constexpr int FR = 2 * 64;                 // h8v per fragment (hi | lo)
constexpr int PK_NT = 4, PK_NC = 16 * PK_NT;      // 16-column tiles per workgroup: 64 columns
constexpr int PK_KS_BYTES = 2 * PK_NT * 1024;     // one k-step of activations in LDS: (hi | lo) x 4 tiles x 64 lanes x 16 B
constexpr int PK_KP = 288, PK_KSPP = PK_KP / 32;  // channels / k-steps of a pass at most
constexpr int PK_ROWS = 17 * 16;
constexpr int PK_OSTRIDE = PK_ROWS + 4;           // floats per COLUMN of the epilogue's LDS tile
constexpr int PK_LDS = PK_KSPP * PK_KS_BYTES;     // 73 728 B; the output tile (64 x 276 x 4 = 70 656) overlays the activation stage
static_assert(PK_LDS >= PK_NC * PK_OSTRIDE * 4, "the output tile overlays the activation stage");

struct PkGeom { int npass, Kp, kspp; };
// K channels in passes of at most 288, every pass a whole number of 8-channel groups and of (zero-padded) 32-channel k-steps
inline PkGeom pk_geom(int K) {
    PkGeom q;
    q.npass = (K + PK_KP - 1) / PK_KP;
    q.Kp = (((K + q.npass - 1) / q.npass) + 7) & ~7;
    q.kspp = (q.Kp + 31) / 32;
    return q;
}

struct PkArgs {
    const h8v* pw;             // [row tiles][passes][k-steps of a pass][hi | lo][64]
    const float* x0;           // [batch, K0, n]
    const float* x1;           // [batch, K1, n] or null
    int layout;                // bit 0 / 1: x0 / x1 channel-BLOCKED, bit 2: y channel-blocked - [batch, C / 8, n, 8]: a token's eight
                               // channels are 32 contiguous bytes (the layer's own intermediate tensors; 16-byte accesses both ways)
    int K0, K1, M, n, npass, Kp, kspp, mtiles, tpg, groups;   // tpg: row tiles per row group
    int64_t cols;              // batch * n
    int64_t wgs, per_xcd;      // logical workgroups, and how many of them each of the 8 XCDs takes
    const float* in_scale;     // [K0 + K1] or null: x <- max(0, x * scale + shift) while staging
    const float* in_shift;
    const float* bias;
    const float* residual;
    float* y;
    int* redo;
    const int* gate;
#ifdef PATS_DIAG
    int* tl;                   // diagnostic library: nine ints per workgroup - the phase durations of its thread 0 (10 ns units)
#endif
};

// Diagnostic library only (python -m pats_amd.build --diag, PATS_AMD_DIAG_LIB=1, PATS_PK_TL=1): s_memrealtime stamps at the phase
// boundaries of a workgroup; launch_conv_pk prints their means (tools/conv_pk_timeline.sh -> profiles/r04_gnn_fine_timeline.txt)
#ifdef PATS_DIAG
#define PK_TS(k) { __builtin_amdgcn_sched_barrier(0); T##k = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define PK_TS(k) do { } while (0)
#endif

__device__ __forceinline__ gptr_h8 uniform_ptr(const h8v* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr_h8)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void split4(const f4v v, h4v& hi, h4v& lo) {
    const f4v s = v * PRE;
    hi = __builtin_convertvector(s, h4v);
    lo = __builtin_convertvector(s - __builtin_convertvector(hi, f4v), h4v);
}

}  // namespace

// weights [K][M] (transposed, as the C-ABI takes them) -> fragments; rows >= M and channels past a pass / past K are zero
__global__ void __launch_bounds__(256)
conv_pack_kernel(const float* __restrict__ wt, int K, int M, int npass, int Kp, int kspp, int mtiles, h8v* __restrict__ pw) {
    const int gid = blockIdx.x * 256 + threadIdx.x, lane = gid & 63, f = gid >> 6;
    if (f >= mtiles * npass * kspp) return;
    const int mt = f / (npass * kspp), r = f - mt * (npass * kspp), ps = r / kspp, ks = r - ps * kspp;
    const int row = mt * 16 + (lane & 15);
    h8v hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kin = ks * 32 + 8 * (lane >> 4) + e, k = ps * Kp + kin;
        const float s = (row < M && kin < Kp && k < K) ? wt[(int64_t)k * M + row] * PRE : 0.f;
        const _Float16 h = (_Float16)s;
        hi[e] = h;
        lo[e] = (_Float16)(s - (float)h);
    }
    pw[f * FR + lane] = hi;
    pw[f * FR + 64 + lane] = lo;
}

// MT: rounds of eight row tiles (wave w: local tiles w, w + 8); SH: one more row tile (local index 8 MT) whose four column tiles
// go to waves 0..3 - a third round for wave 0 alone would idle the other seven
template <int MT, bool SH>
__global__ void __launch_bounds__(512, 4)
conv_pk_kernel(PkArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (g.gate && *g.gate == 0) return;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef PATS_DIAG
    long long T0 = 0, T1 = 0, T2 = 0, T3 = 0, T4 = 0, T5 = 0, T6 = 0, T7 = 0, T8 = 0;
#endif
    PK_TS(0);
    const int qp = lane >> 4, j = lane & 15;
    // Workgroup i runs on XCD i % 8, each with its own L2: the logical tiles are dealt out so that every XCD walks ONE contiguous
    // range of columns (neighbouring tiles share the 128-byte lines their 145-float rows straddle; the row groups of one column
    // tile - neighbours in that order - read the same activations)
    // (32-bit throughout: launch_conv_pk requires 8 * per_xcd and the column count below 2^31)
    const unsigned lid = (blockIdx.x & 7u) * (unsigned)g.per_xcd + (blockIdx.x >> 3);
    if (lid >= (unsigned)g.wgs) return;
    const int grp = (int)(lid % (unsigned)g.groups);
    const unsigned j0 = (lid / (unsigned)g.groups) * PK_NC;
    const int mt0 = grp * g.tpg, mt_end = min(g.mtiles, mt0 + g.tpg);
    const int n = g.n, KSPP = g.kspp, Kt = g.K0 + g.K1;
    int mts[MT];
    bool has[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { mts[m] = mt0 + wave + 8 * m; has[m] = mts[m] < mt_end; mts[m] = min(mts[m], mt_end - 1); }
    // (a row tile this wave does not have is computed on a clamped index and not stored: no branch inside the k loop)
    const int mt_sh = min(mt0 + 8 * MT, mt_end - 1), nt_sh = min(wave, PK_NT - 1);
    const bool has_sh = SH && mt0 + 8 * MT < mt_end && wave < PK_NT;

    // this lane's column while staging and storing: flattened (problem, token); one past the end repeats the last and is not stored
    const unsigned cglob = min(j0 + lane, (unsigned)g.cols - 1u);
    const unsigned cb = cglob / (unsigned)n, ctk = cglob - cb * (unsigned)n;
    const float* base0 = g.x0 + (int64_t)cb * g.K0 * n + ctk;
    const float* base1 = g.K1 > 0 ? g.x1 + (int64_t)cb * g.K1 * n + ctk : base0;
    const int soff = (lane >> 4) * 1024 + (lane & 15) * 16;    // + k-step * 8192 + (group & 3) * 256 (+ 4096 for the lo plane)

    // The accumulators start at 2^12 bias (exact): the bias costs no load, no exchange and no latency in the epilogue.  Lane (q', j)
    // holds rows 16 tile + 4 q' .. + 3 (M is a multiple of 8: all four exist or none)
    const int rows = min(16 * (mt_end - mt0), g.M - 16 * mt0);
    auto bias4 = [&](int mt) {
        const int row = 16 * mt + 4 * qp;
        if (!g.bias || row >= g.M) return f4v{0.f, 0.f, 0.f, 0.f};
        const f4u b = *reinterpret_cast<const f4u*>(g.bias + row);
        return f4v{b.x, b.y, b.z, b.w} * (1.0f / UNS);
    };
    f4v acc[MT][PK_NT], accs = bias4(mt_sh);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const f4v b = bias4(mts[m]);
#pragma unroll
        for (int nt = 0; nt < PK_NT; ++nt) acc[m][nt] = b;
    }

    for (int ps = 0; ps < g.npass; ++ps) {
        // ---- stage the pass: wave w takes the 8-channel groups w, w + 8, .. (a group = one 256-byte row of 64 columns per channel) --
        // ---- the k loop of the pass: no barrier; weight fragments in a ring of two k-steps, straight from L2 ------------------------
        h8v wa[2][MT][2], was[2][2];
        was[0][0] = was[0][1] = was[1][0] = was[1][1] = h8v{0, 0, 0, 0, 0, 0, 0, 0};
#define PK_WLOAD(ks_, slot_)                                                                                     \
    {                                                                                                            \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                         \
            gptr_h8 Wf = uniform_ptr(g.pw + (((int64_t)mts[m] * g.npass + ps) * KSPP + (ks_)) * FR);             \
            wa[slot_][m][0] = Wf[lane];                                                                          \
            wa[slot_][m][1] = Wf[64 + lane];                                                                     \
        }                                                                                                        \
        if (SH && wave < PK_NT) {                                                                                \
            gptr_h8 Wf = uniform_ptr(g.pw + (((int64_t)mt_sh * g.npass + ps) * KSPP + (ks_)) * FR);              \
            was[slot_][0] = Wf[lane];                                                                            \
            was[slot_][1] = Wf[64 + lane];                                                                       \
        }                                                                                                        \
    }
        if (ps > 0) wg_barrier();                               // the previous pass has been read
        const int cbase = ps * g.Kp, cend = min(Kt, cbase + g.Kp), ngroups = KSPP * 4;
        {
            constexpr int NR = (PK_KSPP * 4 + 7) / 8;           // rounds of eight groups: every load of the pass is in flight at once
            f4v ra[NR], rb[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int cg = wave + 8 * i, ch = cbase + cg * 8;       // wave-uniform
                ra[i] = rb[i] = f4v{0.f, 0.f, 0.f, 0.f};
                if (cg < ngroups && ch < cend) {
                    const bool second = ch >= g.K0;             // (K0, K1 multiples of 8: a group never straddles the sources)
                    if (g.layout & (second ? 2 : 1)) {          // blocked source: the group is 32 contiguous bytes of this lane's token
                        const int64_t Ks8 = (second ? g.K1 : g.K0) >> 3, c8 = (second ? ch - g.K0 : ch) >> 3;
                        const float* p = (second ? g.x1 : g.x0) + (((int64_t)cb * Ks8 + c8) * n + ctk) * 8;
                        ra[i] = *reinterpret_cast<const f4v*>(p);
                        rb[i] = *reinterpret_cast<const f4v*>(p + 4);
                    } else {
                        const float* p = second ? base1 + (int64_t)(ch - g.K0) * n : base0 + (int64_t)ch * n;
                        ra[i] = f4v{p[0], p[n], p[2 * n], p[3 * n]};
                        rb[i] = f4v{p[4 * n], p[5 * n], p[6 * n], p[7 * n]};
                    }
                }
            }
            if (ps == 0) PK_TS(1);
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int cg = wave + 8 * i, ch = cbase + cg * 8;
                if (cg < ngroups) {
                    f4v a = ra[i], b = rb[i];
                    if (g.in_scale && ch < cend) {
                        const float* sc = g.in_scale + ch;
                        const float* sh = g.in_shift + ch;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[e] = fmaxf(fmaf(a[e], sc[e], sh[e]), 0.f);
                            b[e] = fmaxf(fmaf(b[e], sc[4 + e], sh[4 + e]), 0.f);
                        }
                    }
                    h4v ah, al, bh, bl;
                    split4(a, ah, al);
                    split4(b, bh, bl);
                    char* d = lds + (cg >> 2) * PK_KS_BYTES + (cg & 3) * 256 + soff;
                    *reinterpret_cast<h8v*>(d) = h8v{ah.x, ah.y, ah.z, ah.w, bh.x, bh.y, bh.z, bh.w};
                    *reinterpret_cast<h8v*>(d + PK_NT * 1024) = h8v{al.x, al.y, al.z, al.w, bl.x, bl.y, bl.z, bl.w};
                }
            }
        }
        if (ps == 0) PK_TS(2);
        wg_barrier();
        if (ps == 0) PK_TS(3);

        PK_WLOAD(0, 0);
        if (KSPP > 1) PK_WLOAD(1, 1);
        for (int ks0 = 0; ks0 < KSPP; ks0 += 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ks = ks0 + s;
                if (ks >= KSPP) break;
                const char* blk = lds + ks * PK_KS_BYTES;
#pragma unroll
                for (int nt = 0; nt < PK_NT; ++nt) {
                    const h8v bh = *reinterpret_cast<const h8v*>(blk + nt * 1024 + lane * 16);
                    const h8v bl = *reinterpret_cast<const h8v*>(blk + PK_NT * 1024 + nt * 1024 + lane * 16);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        f4v c = acc[m][nt];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[s][m][1], bh, c, 0, 0, 0);       // small terms first
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[s][m][0], bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[s][m][0], bh, c, 0, 0, 0);
                        acc[m][nt] = c;
                    }
                }
                if (SH && wave < PK_NT) {                       // the shared row tile: this wave's one column tile of it
                    const h8v bh = *reinterpret_cast<const h8v*>(blk + nt_sh * 1024 + lane * 16);
                    const h8v bl = *reinterpret_cast<const h8v*>(blk + PK_NT * 1024 + nt_sh * 1024 + lane * 16);
                    accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(was[s][1], bh, accs, 0, 0, 0);
                    accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(was[s][0], bl, accs, 0, 0, 0);
                    accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(was[s][0], bh, accs, 0, 0, 0);
                }
                if (ks + 2 < KSPP) PK_WLOAD(ks + 2, s);         // the ring slot just used
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef PK_WLOAD
    }
    PK_TS(4);

    // ---- epilogue: the tile through LDS, COLUMN-major ([column][row], 276 floats apart: an accumulator's four rows are one 16-byte
    // write, a lane's four rows of its column one 16-byte read; 276 = 20 mod 64 spreads 16 lanes over all banks), then whole rows of
    // 64 columns per wave store ------------------------------------------------------------------------------------------------
    wg_barrier();
    PK_TS(5);
    float* ot = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (!has[m]) continue;
        const int rl = 16 * (wave + 8 * m) + 4 * qp;
#pragma unroll
        for (int nt = 0; nt < PK_NT; ++nt) *reinterpret_cast<f4v*>(ot + (16 * nt + j) * PK_OSTRIDE + rl) = acc[m][nt] * UNS;
    }
    if (SH && has_sh) *reinterpret_cast<f4v*>(ot + (16 * wave + j) * PK_OSTRIDE + 16 * 8 * MT + 4 * qp) = accs * UNS;
    wg_barrier();
    PK_TS(6);
    // wave w stores the row quads w, w + 8, ..: all its LDS reads (and residual loads) first, then the stores - no round trip per row.
    // Addresses: a wave-uniform row base + one 32-bit lane offset (the lane's problem relative to the tile's first, its token).
    const bool colok = j0 + lane < (unsigned)g.cols;
    if (g.layout & 4) {
        // blocked output (no residual in this mode): wave w stores the 8-row blocks w, w + 8, .. - two 16-byte reads of its column,
        // two 16-byte stores (32 contiguous bytes per token, 2 KB per wave)
        constexpr int NB = (PK_ROWS / 8 + 7) / 8;
        f4v va[NB], vb[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int rc = min(8 * (wave + 8 * i), rows - 8);
            va[i] = *reinterpret_cast<const f4v*>(ot + lane * PK_OSTRIDE + rc);
            vb[i] = *reinterpret_cast<const f4v*>(ot + lane * PK_OSTRIDE + rc + 4);
        }
        bool badb = false;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u) badb |= !(fabsf(va[i][u]) <= 3.0e38f) || !(fabsf(vb[i][u]) <= 3.0e38f);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r0 = 8 * (wave + 8 * i);
            if (r0 < rows && colok) {
                float* yb = g.y + (((int64_t)cb * (g.M >> 3) + ((16 * mt0 + r0) >> 3)) * n + ctk) * 8;
                *reinterpret_cast<f4v*>(yb) = va[i];
                *reinterpret_cast<f4v*>(yb + 4) = vb[i];
            }
        }
        if (__any(badb && colok) && lane == 0 && g.redo) atomicOr(g.redo, 1);
        return;
    }
    const unsigned b0 = j0 / (unsigned)n;                                       // wave-uniform
    const unsigned ovoff = ((cb - b0) * (unsigned)g.M * (unsigned)n + ctk) * 4u;
    const int64_t obase = ((int64_t)b0 * g.M + 16 * mt0) * n;                   // wave-uniform
    constexpr int NQ = (PK_ROWS / 4 + 7) / 8;
    f4v v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int rc = min(4 * (wave + 8 * i), rows - 4);
        v[i] = *reinterpret_cast<const f4v*>(ot + lane * PK_OSTRIDE + rc);
    }
    PK_TS(7);
    bool bad = false;
    if (g.residual) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int rc = min(4 * (wave + 8 * i), rows - 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bad |= !(fabsf(v[i][u]) <= 3.0e38f);
                const float r = colok ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.residual + obase + (int64_t)(rc + u) * n) + ovoff) : 0.f;
                v[i][u] = r + v[i][u];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u) bad |= !(fabsf(v[i][u]) <= 3.0e38f);
    }
    bad &= colok;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int r0 = 4 * (wave + 8 * i);
        if (r0 < rows && colok) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                *reinterpret_cast<float*>(reinterpret_cast<char*>(g.y + obase + (int64_t)(r0 + u) * n) + ovoff) = v[i][u];
        }
    }
#ifdef PATS_DIAG
    PK_TS(8);
    if (g.tl && t == 0) {
        int* o = g.tl + (size_t)lid * 9;
        o[0] = (int)(T1 - T0); o[1] = (int)(T2 - T1); o[2] = (int)(T3 - T2); o[3] = (int)(T4 - T3); o[4] = (int)(T5 - T4);
        o[5] = (int)(T6 - T5); o[6] = (int)(T7 - T6); o[7] = (int)(T8 - T7); o[8] = (int)(T0 & 0x7fffffff);
    }
#endif
    if (__any(bad) && lane == 0 && g.redo) atomicOr(g.redo, 1);
}

// ---- host side ------------------------------------------------------------------------------------------------------------
size_t conv_packed_bytes(int K, int M) {
    const PkGeom q = pk_geom(K);
    return (size_t)((M + 15) / 16) * q.npass * q.kspp * FR * sizeof(h8v);
}

int launch_conv_pack(const float* wt, int K, int M, void* packed, hipStream_t st) {
    const PkGeom q = pk_geom(K);
    const int mtiles = (M + 15) / 16;
    const int threads = mtiles * q.npass * q.kspp * 64;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, wt, K, M, q.npass, q.Kp, q.kspp, mtiles,
                       (h8v*)packed);
    return check_launch("conv_pack_kernel");
}

// 1 if this device grants the kernel its 72 KB of dynamic LDS (asked once per device)
bool conv_pk_ready() {
    static int state[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
    if (state[dev] == 0) {
        const bool ok = hipFuncSetAttribute((const void*)conv_pk_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PK_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)conv_pk_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PK_LDS) == hipSuccess &&
                        hipFuncSetAttribute((const void*)conv_pk_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PK_LDS) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        state[dev] = ok ? 1 : -1;
    }
    return state[dev] == 1;
}

// y = W . (x0 | x1) with packed W; K0, K1 multiples of 8.  redo: as conv_lean_kernel (the caller queues conv1x1_kernel behind).
// PATS_ERR_UNSUPPORTED: the device refused the LDS - the caller takes the unpacked kernels.
int launch_conv_pk(const void* packed, const float* x0, const float* x1, int K0, int K1, int M, int n, int64_t cols,
                   const float* in_scale, const float* in_shift, const float* bias, const float* residual, float* y, int* redo,
                   const int* gate, hipStream_t st, int layout) {
    if (!conv_pk_ready()) return PATS_ERR_UNSUPPORTED;
    PATS_REQUIRE(!((layout & 4) && residual), "conv_pk: no residual with a blocked output");
    const PkGeom q = pk_geom(K0 + K1);
    const int mtiles = (M + 15) / 16;
    // row groups of at most 17 tiles, evenly: 264 rows -> one group of 16 + the shared 17th; 528 -> 17 + 16; 128 -> 8
    const int groups = (mtiles + 16) / 17, tpg = (mtiles + groups - 1) / groups;
    const int64_t wgs = ((cols + PK_NC - 1) / PK_NC) * groups, per_xcd = (wgs + 7) / 8;
    PATS_REQUIRE(8 * per_xcd < (1ll << 31), "conv_pk: grid too large (split the batch)");
    PkArgs g{(const h8v*)packed, x0, x1, layout, K0, K1, M, n, q.npass, q.Kp, q.kspp, mtiles, tpg, groups, cols, wgs, per_xcd, in_scale, in_shift, bias, residual, y, redo, gate};
#ifdef PATS_DIAG
    g.tl = nullptr;
    if (diag_env("PATS_PK_TL") && !(layout & 4)) { (void)hipMalloc((void**)&g.tl, (size_t)wgs * 9 * 4); (void)hipMemset(g.tl, 0, (size_t)wgs * 9 * 4); }
#endif
    const dim3 grid((unsigned)(8 * per_xcd)), block(512);
    // (10..16 tiles take the 17-tile instantiation too: without the shared tile the same loop spills 29 registers at the 128 cap)
    if (tpg <= 8) hipLaunchKernelGGL((conv_pk_kernel<1, false>), grid, block, PK_LDS, st, g);
    else if (tpg == 9) hipLaunchKernelGGL((conv_pk_kernel<1, true>), grid, block, PK_LDS, st, g);
    else hipLaunchKernelGGL((conv_pk_kernel<2, true>), grid, block, PK_LDS, st, g);
#ifdef PATS_DIAG
    if (g.tl) {
        (void)hipStreamSynchronize(st);
        std::vector<int> h((size_t)wgs * 9);
        (void)hipMemcpy(h.data(), g.tl, h.size() * 4, hipMemcpyDeviceToHost);
        double sum[8] = {0};
        long long tmin = 1ll << 62, tmax = 0;
        for (int64_t w = 0; w < wgs; ++w) {
            for (int q8 = 0; q8 < 8; ++q8) sum[q8] += h[w * 9 + q8];
            tmin = std::min<long long>(tmin, h[w * 9 + 8]);
            tmax = std::max<long long>(tmax, h[w * 9 + 8]);
        }
        fprintf(stderr, "conv_pk timeline (K=%d M=%d, %lld workgroups, mean per workgroup, us): start->loads issued %.2f, ->stashed %.2f, ->barrier %.2f, "
                        "->k loops %.2f, ->barrier %.2f, ->tile in LDS + barrier %.2f, ->read back %.2f, ->stores issued %.2f; first to last start %.1f us\n",
                K0 + K1, M, (long long)wgs, sum[0] / wgs / 100, sum[1] / wgs / 100, sum[2] / wgs / 100, sum[3] / wgs / 100, sum[4] / wgs / 100,
                sum[5] / wgs / 100, sum[6] / wgs / 100, sum[7] / wgs / 100, (tmax - tmin) / 100.0);
        (void)hipFree(g.tl);
    }
#endif
    return check_launch("conv_pk_kernel");
}

}  // namespace pats
