// The steps either side of the OT path on gfx950: index arithmetic on small
// grids, one thread per element, no host round-trip inside a call.
//   merge_patches_new / _old   models/second_layer.py:137-238
//   third-level inputs          models/pats.py:53-58
//   result scatter              models/pats.py:59-67
//   get_result                  utils/utils.py:189-213
// The reference expresses all of these as boolean-mask indexing (each one a device->host sync for
// the output size) plus argsort / scatter on [1,4h,4w,9] tensors.  Here the order-preserving
// compactions use one exclusive scan of the keep-flags (scan_flags) and every consumer computes its
// own output slot; the merge is a gather ("new": each output entry finds its unique writer) or a
// last-writer-wins scatter made deterministic with atomicMax on (source order, value) ("old").
#include "common.hpp"

namespace pats {

// ---- exclusive scan of keep-flags (keep = flag byte == 0) ------------------------------------------
// offs[i] = number of kept elements before i inside i's 2048-element tile; tile_base[t] = kept elements
// before tile t (after scan_tiles_kernel); total = kept elements overall.
constexpr int SCAN_TILE = 2048;

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

__global__ void __launch_bounds__(256)
scan_flags_kernel(const uint8_t* __restrict__ flags, int64_t n, int32_t* __restrict__ offs,
                  int32_t* __restrict__ tile_base) {
    __shared__ int wsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_TILE + t * 8;
    int keep[8], c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        keep[k] = (i0 + k < n) ? (flags[i0 + k] == 0) : 0;
        c += keep[k];
    }
    const int incl = wave_incl_scan(c, lane);
    if (lane == 63) wsum[wave] = incl;
    wg_barrier();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int run = base + incl - c;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (i0 + k < n) offs[i0 + k] = run;
        run += keep[k];
    }
    if (t == 255) tile_base[blockIdx.x] = run;
}

__global__ void __launch_bounds__(1024)
scan_tiles_kernel(int32_t* __restrict__ tile_base, int tiles, int64_t* __restrict__ total) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    wg_barrier();
    for (int b0 = 0; b0 < tiles; b0 += 1024) {
        const int v = (b0 + t < tiles) ? tile_base[b0 + t] : 0;
        const int incl = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wave] = incl;
        wg_barrier();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        if (b0 + t < tiles) tile_base[b0 + t] = base + incl - v;
        wg_barrier();
        if (t == 1023) carry = base + incl;
        wg_barrier();
    }
    if (t == 0 && total) *total = carry;
}

// n <= 16 384 flags (a chunk's 144 B cells, a pair's coarse cells): the whole scan in ONE workgroup and one launch - 16 flags a
// thread; offs holds the global exclusive offsets, the tile bases are zero.  (Walking PATS.forward chunk by chunk runs four scans
// per chunk: two launches each were 64 of a pair's 210.)
constexpr int SCAN_SMALL = 16384;
__global__ void __launch_bounds__(1024)
scan_small_kernel(const uint8_t* __restrict__ flags, int n, int32_t* __restrict__ offs, int32_t* __restrict__ tile_base, int tiles,
                  int64_t* __restrict__ total) {
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i0 = t * 16;
    int keep[16], c = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        keep[k] = (i0 + k < n) ? (flags[i0 + k] == 0) : 0;
        c += keep[k];
    }
    const int incl = wave_incl_scan(c, lane);
    if (lane == 63) wsum[wave] = incl;
    wg_barrier();
    int base = 0, all = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) base += wsum[w]; all += wsum[w]; }
    int run = base + incl - c;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (i0 + k < n) offs[i0 + k] = run;
        run += keep[k];
    }
    if (t < tiles) tile_base[t] = 0;
    if (t == 0 && total) *total = all;
}

struct Scan {
    int32_t* offs;
    int32_t* tile_base;
    __device__ __forceinline__ int64_t at(int64_t i) const { return (int64_t)offs[i] + tile_base[i / SCAN_TILE]; }
};

static size_t scan_bytes(int64_t n) {
    const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (size_t)(((n + 3) & ~3ll) + ((tiles + 3) & ~3ll)) * sizeof(int32_t);
}

// carve a Scan out of `ws` and run it; *ws is advanced
static int run_scan(const uint8_t* flags, int64_t n, int64_t* total_dev, char** ws, Scan* sc, hipStream_t st) {
    const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    sc->offs = reinterpret_cast<int32_t*>(*ws);
    sc->tile_base = sc->offs + ((n + 3) & ~3ll);
    *ws += scan_bytes(n);
    if (n == 0) {
        if (total_dev && hipMemsetAsync(total_dev, 0, sizeof(int64_t), st) != hipSuccess) return check_launch("scan memset");
        return PATS_OK;
    }
    if (n <= SCAN_SMALL) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, flags, (int)n, sc->offs, sc->tile_base, (int)tiles, total_dev);
        return check_launch("scan_small");
    }
    hipLaunchKernelGGL(scan_flags_kernel, dim3((unsigned)tiles), dim3(256), 0, st, flags, n, sc->offs, sc->tile_base);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, st, sc->tile_base, (int)tiles, total_dev);
    return check_launch("scan_flags");
}

// ---- merge_patches ----------------------------------------------------------------------------------
// The rows one merge call works on.  Single-chunk API: rows [0, count) of the caller's tensors.  Batch mode
// (pats_merge_patches_batch): block [blk[0], blk[1]) of the row table of batch.hip, read from DEVICE memory - the launch
// covers the block's capacity and threads past its end leave.
struct RowBlock {
    const int64_t* blk;
    int64_t count;
    __device__ __forceinline__ void get(int64_t& base, int64_t& n) const {
        base = blk ? blk[0] : 0;
        n = blk ? blk[1] - blk[0] : count;
    }
};

// slot[q] = row of trust_score that coarse patch q owns in this chunk, or -1; patch_of[b] = its inverse
__global__ void __launch_bounds__(256)
merge_slots_kernel(const uint8_t* __restrict__ ifn_L1, Scan sc, int64_t NP, int64_t B, int32_t* __restrict__ slot,
                   int32_t* __restrict__ patch_of) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= NP) return;
    int64_t s = -1;
    if (!ifn_L1[q]) {
        s = sc.at(q);
        if (s >= B) s = -1;                 // more unmasked patches than rows: ignored (host wrapper validates)
        else patch_of[s] = (int32_t)q;
    }
    slot[q] = (int32_t)s;
}

// border weighting, flag update, score hand-over into scores_back (second_layer.py:140-149,161-163 / :192-201,210-211)
// for element e = row b, window cell `cell` of the trust / flag tensors
__device__ __forceinline__ void merge_prepare_el(int merge_new, int64_t e, float* __restrict__ trust, uint8_t* __restrict__ ifn_L2,
                                                 const int32_t* __restrict__ patch_of, double* __restrict__ scores_back) {
    const int64_t b = e / 144;
    const int cell = (int)(e - b * 144), x = cell % 12, y = cell / 12;
    float t = trust[e];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (x < 3 - i || x > 7 + i || y < 3 - i || y > 7 + i) t *= 2.0f;
    uint8_t f = ifn_L2[e];
    if (t > 2.0f) f = 1;
    if (x < 1 || x > 10 || y < 1 || y > 10) f = 1;
    if (merge_new && !f) t -= 10000.0f;
    trust[e] = t;
    ifn_L2[e] = f;
    const int a = y / 4, r = y % 4, c = x / 4, s = x % 4;
    const int64_t q = patch_of[b];
    if (q >= 0) scores_back[(q * 16 + r * 4 + s) * 9 + a * 3 + c] = (double)t;
}

__global__ void __launch_bounds__(256)
merge_prepare_kernel(int merge_new, RowBlock rb, float* __restrict__ trust, uint8_t* __restrict__ ifn_L2,
                     const int32_t* __restrict__ patch_of, double* __restrict__ scores_back) {
    int64_t base, B;
    rb.get(base, B);
    const int64_t el = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (el >= B * 144) return;
    merge_prepare_el(merge_new, el + base * 144, trust, ifn_L2, patch_of, scores_back);
}

struct MergeGeom {
    int h, w, h4, w4;
    int64_t per;                 // 4h * 4w * 9
};

// "new" (second_layer.py:214-240) as a gather.  Output entry (patch q, window cell (a,r,c,s)) lives at
// owner-layout position (Y', X', k') = (4hh + r, 4ww + s, 3a + c).  The only fine cell whose scatter
// index (:234-236) can hit it is (Y, X) = (Y' + 4(a-1), X' + 4(c-1)) choosing sb = 8 - k'; that cell's
// choice is the first minimum of ITS OWNER's nine scores (+100000 for windows leaving the grid) -
// argsort runs on scores_back_use, not on the re-gathered copy (:232).  The value scattered is
// if_matching2[Y, X, sb] = if_matching at the entry itself, i.e. the (updated) L2 flag.
__device__ __forceinline__ void merge_select_new_el(const MergeGeom& g, int64_t e, const int32_t* __restrict__ patch_of,
                                                    const uint8_t* __restrict__ ifn_L2, const double* __restrict__ scores_back,
                                                    const uint8_t* __restrict__ row_forced, uint8_t* __restrict__ out) {
    const int64_t b = e / 144;
    const int cell = (int)(e - b * 144), x = cell % 12, y = cell / 12;
    const int a = y / 4, r = y % 4, c = x / 4, s = x % 4;
    const int64_t q = patch_of[b];
    uint8_t res = 1;
    if (q >= 0) {
        const int hw = g.h * g.w;
        const int64_t bt = q / hw;
        const int p = (int)(q - bt * hw), hh = p / g.w, ww = p % g.w;
        const int Y = 4 * hh + r + 4 * (a - 1), X = 4 * ww + s + 4 * (c - 1);
        if (Y >= 0 && Y < g.h4 && X >= 0 && X < g.w4) {
            const int64_t qs = bt * hw + (Y / 4) * g.w + X / 4;
            const double* u = scores_back + (qs * 16 + (Y % 4) * 4 + X % 4) * 9;
            int sb = 0;
            double best = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int by = Y + 4 * (k / 3 - 1), bx = X + 4 * (k % 3 - 1);
                double v = u[k];
                if (by < 0 || by >= g.h4 || bx < 0 || bx >= g.w4) v += 100000.0;
                if (k == 0 || v < best) { best = v; sb = k; }
            }
            if (sb == 8 - (a * 3 + c)) res = ifn_L2[e];
        }
    }
    if (row_forced && row_forced[b]) res = 1;          // pats.py:38-39 on the returned flags (batch mode)
    out[e] = res;
}

__global__ void __launch_bounds__(256)
merge_select_new_kernel(MergeGeom g, RowBlock rb, const int32_t* __restrict__ patch_of,
                        const uint8_t* __restrict__ ifn_L2, const double* __restrict__ scores_back,
                        const uint8_t* __restrict__ row_forced, uint8_t* __restrict__ out) {
    int64_t base, B;
    rb.get(base, B);
    const int64_t el = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (el >= B * 144) return;
    merge_select_new_el(g, el + base * 144, patch_of, ifn_L2, scores_back, row_forced, out);
}

// "old" (second_layer.py:165-189): one thread per fine cell re-gathers its nine candidates by geometry
// (channel k from the owner 4(a-1), 4(c-1) cells away where that slice assignment reaches, its own
// otherwise), applies -10000 to matching ones, takes the first minimum and scatters the flag to the
// (clamped) source entry.  ATen's CPU scatter is sequential, so the last source wins:
// atomicMax on ((source index + 1) << 1 | value).
__device__ __forceinline__ void merge_scatter_old_cell(const MergeGeom& g, int64_t bt, int64_t n, const int32_t* __restrict__ slot,
                                                       const uint8_t* __restrict__ ifn_L2, const double* __restrict__ scores_back,
                                                       unsigned* __restrict__ winner) {
    const int Y = (int)(n / g.w4), X = (int)(n % g.w4), hw = g.h * g.w;
    int sb = 0;
    double best = 0.0;
    bool m_best = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dx = k / 3 - 1, dy = k % 3 - 1;
        const bool reach = Y >= 4 * max(dx, 0) && Y < g.h4 + 4 * min(dx, 0) && X >= 4 * max(dy, 0) && X < g.w4 + 4 * min(dy, 0);
        const int sy = reach ? Y - 4 * dx : Y, sx = reach ? X - 4 * dy : X;
        const int64_t qs = bt * hw + (sy / 4) * g.w + sx / 4;
        double v = scores_back[(qs * 16 + (sy % 4) * 4 + sx % 4) * 9 + k];
        const int sl = slot[qs];
        const bool m = sl >= 0 && !ifn_L2[(int64_t)sl * 144 + ((k / 3) * 4 + sy % 4) * 12 + (k % 3) * 4 + sx % 4];
        if (m) v -= 10000.0;
        if (k == 0 || v < best) { best = v; sb = k; m_best = m; }
    }
    int64_t s2 = sb + n * 9 - (int64_t)(sb % 3 - 1) * 36 - (int64_t)(sb / 3 - 1) * 4 * g.w4 * 9;
    const int64_t hy = n / g.w / 4 - (sb / 3 - 1) * 4, wx = n % g.w4 - (sb % 3 - 1) * 4;
    if (hy < 0 || hy >= g.h4 || wx < 0 || wx >= g.w4) m_best = false;
    s2 = min(max(s2, (int64_t)0), g.per - 1);
    atomicMax(&winner[bt * g.per + s2], (unsigned)(((n + 1) << 1) | (m_best ? 0 : 1)));
}

__global__ void __launch_bounds__(256)
merge_scatter_old_kernel(MergeGeom g, int batch_num, const int32_t* __restrict__ slot,
                         const uint8_t* __restrict__ ifn_L2, const double* __restrict__ scores_back,
                         unsigned* __restrict__ winner) {
    const int64_t cells = (int64_t)g.h4 * g.w4;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= cells * batch_num) return;
    const int64_t bt = e / cells;
    merge_scatter_old_cell(g, bt, e - bt * cells, slot, ifn_L2, scores_back, winner);
}

__device__ __forceinline__ void merge_finish_old_el(const MergeGeom& g, int64_t e, const int32_t* __restrict__ patch_of,
                                                    const unsigned* __restrict__ winner, const uint8_t* __restrict__ row_forced,
                                                    uint8_t* __restrict__ out) {
    const int64_t b = e / 144;
    const int cell = (int)(e - b * 144), x = cell % 12, y = cell / 12;
    const int a = y / 4, r = y % 4, c = x / 4, s = x % 4;
    const int64_t q = patch_of[b];
    uint8_t res = 1;
    if (q >= 0) {
        const int hw = g.h * g.w;
        const int64_t bt = q / hw;
        const int p = (int)(q - bt * hw), hh = p / g.w, ww = p % g.w;
        const unsigned v = winner[bt * g.per + ((int64_t)(4 * hh + r) * g.w4 + 4 * ww + s) * 9 + a * 3 + c];
        if (v) res = (uint8_t)(v & 1u);
    }
    if (row_forced && row_forced[b]) res = 1;
    out[e] = res;
}

__global__ void __launch_bounds__(256)
merge_finish_old_kernel(MergeGeom g, RowBlock rb, const int32_t* __restrict__ patch_of,
                        const unsigned* __restrict__ winner, const uint8_t* __restrict__ row_forced,
                        uint8_t* __restrict__ out) {
    int64_t base, B;
    rb.get(base, B);
    const int64_t el = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (el >= B * 144) return;
    merge_finish_old_el(g, el + base * 144, patch_of, winner, row_forced, out);
}

// ---- merge_patches_new for SEVERAL chunks of a row table in two fully parallel launches ---------------------------------------
// The reference walks the chunks in order because they couple through scores_back (pats.py:32,37): the select of chunk c reads,
// for every candidate window, the score its owner patch left there - written by the prepare of the LATEST chunk <= c that holds the
// owner (a patch of the grid row two chunks share is in both), or what scores_back held before.  That value is a function of the
// owner row's RAW trust score and flag alone (merge_prepared below), so nothing has to be written before it is read:
//   select   one thread per (row, window cell): finds each candidate owner's row through the table (row_slot of chunk c, c - 1, ...),
//            recomputes the nine prepared scores from the raw tensors, falls back to the scores_back it was handed (zeros when the
//            call starts a pair) and writes `out`.  Reads trust / flags, writes only `out`.
//   prepare  afterwards, in place: border weighting + flag update of every row (second_layer.py:192-201), the scores of a patch's
//            LAST row in the range into scores_back (what the sequential walk leaves there), zeros for patches without a row when
//            the call starts a pair, "no match" for rows outside the walked blocks.
// Two launches for all chunks of all pairs (rounds 3-5: two per chunk, 16 of a pair's 60 launches at one pair a step); the same
// two per chunk when PATS.forward's loop is walked chunk by chunk (pats_merge_patches_chunks).  A one-workgroup-per-pair kernel
// with barriers between the phases was tried first: 18 us a chunk, VALU-bound on its single CU.
struct MergeTable {
    int Cmax, c_lo, c_hi;
    int64_t pairs, row_origin, rows_local;
    const int64_t* chunk_base; const int32_t* row_cell; const int32_t* row_slot; const uint8_t* row_forced;
};

// second_layer.py:194-201 for one cell: weighted trust score and updated flag from the raw ones
__device__ __forceinline__ float merge_prepared(float t, uint8_t f_in, int x, int y, uint8_t& f_out) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (x < 3 - i || x > 7 + i || y < 3 - i || y > 7 + i) t *= 2.0f;
    uint8_t f = f_in;
    if (t > 2.0f) f = 1;
    if (x < 1 || x > 10 || y < 1 || y > 10) f = 1;
    if (!f) t -= 10000.0f;
    f_out = f;
    return t;
}

__global__ void __launch_bounds__(256)
merge_select_new_par_kernel(MergeGeom g, MergeTable tb, const float* __restrict__ trust, const uint8_t* __restrict__ ifn_L2,
                            const double* __restrict__ scores_back, int fresh, uint8_t* __restrict__ out) {
    const int64_t first = tb.chunk_base[tb.c_lo], last = tb.chunk_base[tb.c_hi];
    const int64_t el = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (el >= (last - first) * 144) return;
    const int64_t row = first + el / 144;
    const int cell = (int)(el % 144), x = cell % 12, y = cell / 12;
    int c = tb.c_lo;
    while (c + 1 < tb.c_hi && row >= tb.chunk_base[c + 1]) ++c;       // the row's chunk
    const int N = g.h * g.w;
    const int64_t NP = tb.pairs * N;
    const int32_t q = tb.row_cell[row];
    trust -= tb.row_origin * 144; ifn_L2 -= tb.row_origin * 144; out -= tb.row_origin * 144;
    const int64_t e = row * 144 + cell;
    uint8_t res = 1;
    if (q >= 0) {
        uint8_t f_own;
        (void)merge_prepared(trust[e], ifn_L2[e], x, y, f_own);
        const int bt = q / N, pl = q - bt * N, hh = pl / g.w, ww = pl - hh * g.w;
        const int a = y >> 2, r = y & 3, cc = x >> 2, s_ = x & 3;
        const int Y = 4 * hh + r + 4 * (a - 1), X = 4 * ww + s_ + 4 * (cc - 1);
        if (Y >= 0 && Y < g.h4 && X >= 0 && X < g.w4) {
            const int64_t qs = (int64_t)bt * N + (Y >> 2) * g.w + (X >> 2);      // the owner of fine cell (Y, X)
            int64_t rs = -1;                                                  // its row in the latest chunk <= c of the walk
            for (int c2 = c; c2 >= tb.c_lo && rs < 0; --c2) rs = tb.row_slot[(int64_t)c2 * NP + qs];
            const double* u = scores_back + (qs * 16 + (Y & 3) * 4 + (X & 3)) * 9;
            int sb = 0;
            double best = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                double v;
                if (rs >= 0) {
                    const int yy = (k / 3) * 4 + (Y & 3), xx = (k % 3) * 4 + (X & 3);
                    uint8_t fk;
                    v = (double)merge_prepared(trust[rs * 144 + yy * 12 + xx], ifn_L2[rs * 144 + yy * 12 + xx], xx, yy, fk);
                } else {
                    v = fresh ? 0.0 : u[k];
                }
                const int by = Y + 4 * (k / 3 - 1), bx = X + 4 * (k % 3 - 1);
                if (by < 0 || by >= g.h4 || bx < 0 || bx >= g.w4) v += 100000.0;
                if (k == 0 || v < best) { best = v; sb = k; }
            }
            if (sb == 8 - (a * 3 + cc)) res = f_own;
        }
    }
    if (tb.row_forced[row]) res = 1;                    // pats.py:38-39 on the returned flags
    out[e] = res;
}

__global__ void __launch_bounds__(256)
merge_prepare_new_par_kernel(MergeGeom g, MergeTable tb, float* __restrict__ trust, uint8_t* __restrict__ ifn_L2,
                             double* __restrict__ scores_back, int fresh, uint8_t* __restrict__ out) {
    const int64_t first = tb.chunk_base[tb.c_lo], last = tb.chunk_base[tb.c_hi];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int N = g.h * g.w;
    const int64_t NP = tb.pairs * N;
    if (fresh && i < NP * 144) {                        // pats.py:32: a pair starts from zeros - patches without a row keep them
        const int64_t qz = i / 144;
        bool has = false;
        for (int c2 = tb.c_lo; c2 < tb.c_hi && !has; ++c2) has = tb.row_slot[(int64_t)c2 * NP + qz] >= 0;
        if (!has) scores_back[i] = 0.0;
    }
    if (i >= tb.rows_local * 144) return;
    const int64_t row = tb.row_origin + i / 144;
    if (row < first || row >= last) { out[i] = 1; return; }            // rows outside the walked blocks: "no match"
    const int cell = (int)(i % 144), x = cell % 12, y = cell / 12;
    uint8_t f;
    const float t = merge_prepared(trust[i], ifn_L2[i], x, y, f);
    trust[i] = t;
    ifn_L2[i] = f;
    const int32_t q = tb.row_cell[row];
    if (q < 0) return;
    int c = tb.c_lo;
    while (c + 1 < tb.c_hi && row >= tb.chunk_base[c + 1]) ++c;
    for (int c2 = c + 1; c2 < tb.c_hi; ++c2)
        if (tb.row_slot[(int64_t)c2 * NP + q] >= 0) return;           // a later chunk of the walk holds the patch again: its row writes
    scores_back[((int64_t)q * 16 + (y & 3) * 4 + (x & 3)) * 9 + (y >> 2) * 3 + (x >> 2)] = (double)t;
}

// ---- third-level inputs, pats.py:53-58 ---------------------------------------------------------------
__global__ void __launch_bounds__(256)
third_inputs_kernel(const uint8_t* __restrict__ ifn2, const float* __restrict__ pts, int64_t n, Scan sc,
                    int64_t capacity, float* __restrict__ mk0, float* __restrict__ mk1, int64_t* __restrict__ b_ids) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n || ifn2[e]) return;
    const int64_t P = sc.at(e);
    if (P >= capacity) return;
    const int64_t b = e / 144;
    const int cell = (int)(e - b * 144);
    mk0[P * 2 + 0] = (float)(cell % 12 * 4 + 2) * 2.0f;
    mk0[P * 2 + 1] = (float)(cell / 12 * 4 + 2) * 2.0f;
    mk1[P * 2 + 0] = rintf(pts[e * 2 + 1] * 4.0f) * 2.0f;      // torch.round: half to even
    mk1[P * 2 + 1] = rintf(pts[e * 2 + 0] * 4.0f) * 2.0f;
    b_ids[P] = b;
}

// ---- result scatter, pats.py:59-67 -------------------------------------------------------------------
__global__ void __launch_bounds__(256)
refine_scatter_kernel(const uint8_t* __restrict__ ifn2, const float* __restrict__ pts, const float* __restrict__ mkpts1,
                      const float* __restrict__ label, int label_stride, int64_t P_rows, int64_t n16, Scan sc,
                      uint8_t* __restrict__ ifn16, float* __restrict__ pts16) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;       // output index, [B,12,4,12,4] order
    if (o >= n16) return;
    const int64_t b = o / 2304;
    const int q = (int)(o - b * 2304), yy = q / 48, xx = q % 48;
    const int cell = (yy / 4) * 12 + xx / 4, sub = (yy % 4) * 4 + xx % 4;
    const int64_t e = b * 144 + cell;
    float py = pts[e * 2], px = pts[e * 2 + 1];
    uint8_t f = 1;
    if (!ifn2[e]) {
        const int64_t P = sc.at(e);
        if (P < P_rows) {
            py = mkpts1[(P * 16 + sub) * 2];
            px = mkpts1[(P * 16 + sub) * 2 + 1];
            f = label[(P * 16 + sub) * (int64_t)label_stride] < -9.9f;
        }
    }
    pts16[o * 2] = py;
    pts16[o * 2 + 1] = px;
    ifn16[o] = f;
}

// ---- get_result (layer_num = 2), utils.py:189-213 -----------------------------------------------------
__global__ void __launch_bounds__(256)
result_rows_kernel(const uint8_t* __restrict__ ifn0, int64_t n, Scan sc, int64_t rows1, int32_t* __restrict__ row_cell) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n || ifn0[e]) return;
    const int64_t k = sc.at(e);
    if (k < rows1) row_cell[k] = (int32_t)e;
}

struct ResultArgs {
    const uint8_t* ifn1;
    const float* ap0; const float* ap1; const float* sc0; const float* sc1;
    int64_t sc1_cell_stride;
    int s0, h0, w0, s1, h1, w1;
    const uint8_t* ch0; const uint8_t* ch1;
    const int64_t* rows0_dev;    // number of surviving level-0 cells
    int64_t rows1, capacity;
    float* ml; float* mr;
    // chunk-batch mode (pats_get_result_chunks_f32): the level-0 tensors cover `period0` cells (pairs * N) and every
    // chunk of the level-0 batch reads the same ones (e % period0); points arrive un-flipped and un-divided
    // (pats.py:71: `.flip(dims=[2]) / 32.0`, `/ 2.0` - done on load); the level-1 scale of row k is the level-0 scale of
    // its cell (pats.py:70); match_row (optional) receives the level-1 row of every match
    int64_t period0;
    float ap0_div, ap1_div;
    int32_t* match_row;
};

__global__ void __launch_bounds__(256)
get_result_kernel(ResultArgs g, const int32_t* __restrict__ row_cell, Scan sc1) {
    // a workgroup stays inside one level-1 row (blocks per row = ceil(n1 / 256)): the row index and everything that
    // hangs on it is workgroup-uniform, and no thread does a 64-bit division
    const unsigned n0 = (unsigned)(g.h0 * g.w0), n1 = (unsigned)(g.h1 * g.w1), bpr = (n1 + 255u) / 256u;
    const unsigned krow = blockIdx.x / bpr, j = (blockIdx.x - krow * bpr) * 256u + threadIdx.x;
    const int64_t k = krow, f = k * n1 + j;
    if (j >= n1 || g.ifn1[f]) return;
    if (k >= *g.rows0_dev) return;                  // more rows than surviving cells: ignored
    const int64_t M = sc1.at(f);
    if (M >= g.capacity) return;
    const int64_t ecell = row_cell[k];
    const unsigned bt = (unsigned)ecell / n0, i = (unsigned)ecell - bt * n0;
    const int64_t e = g.period0 ? ecell % g.period0 : ecell;      // where the level-0 point / scale of that cell live
    const bool c0 = g.ch0[bt] != 0, c1 = g.ch1[k] != 0;
    const float z0 = (float)g.s0, z1 = (float)g.s1;
    const float sc1l = g.sc1 ? g.sc1[g.sc1_cell_stride ? f * g.sc1_cell_stride + 1 : k * 2 + 1] : g.sc0[e * 2 + 1];
    const float sc1r = g.sc1 ? g.sc1[g.sc1_cell_stride ? f * g.sc1_cell_stride : k * 2] : g.sc0[e * 2];
    if (g.match_row) g.match_row[M] = (int32_t)k;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float pos0 = (float)((d == 0 ? i / (unsigned)g.w0 : i % (unsigned)g.w0) * (unsigned)g.s0);
        float dl0 = pos0 + 0.5f * z0;
        dl0 = dl0 - (1.5f * g.sc0[e * 2 + 1]) * z0;
        const float a0 = g.ap0_div != 0.0f ? g.ap0[e * 2 + (1 - d)] / g.ap0_div : g.ap0[e * 2 + d];
        const float a1 = g.ap1_div != 0.0f ? g.ap1[f * 2 + (1 - d)] / g.ap1_div : g.ap1[f * 2 + d];
        const float dr0 = (a0 - 1.5f * g.sc0[e * 2]) * z0;
        const float l0 = 0.0f + (c0 ? dl0 : dr0), r0 = 0.0f + (c0 ? dr0 : dl0);
        const float pos1 = (float)((d == 0 ? j / (unsigned)g.w1 : j % (unsigned)g.w1) * (unsigned)g.s1);
        float dl1 = pos1 + 0.5f * z1;
        dl1 = dl1 * sc1l;
        const float dr1 = (a1 * z1) * sc1r;
        g.ml[M * 2 + d] = l0 + (c1 ? dl1 : dr1);
        g.mr[M * 2 + d] = r0 + (c1 ? dr1 : dl1);
    }
}

}  // namespace pats

using namespace pats;

static inline unsigned blocks256(int64_t n) { return (unsigned)((n + 255) / 256); }

extern "C" size_t pats_merge_workspace_bytes(int64_t B, int H, int W, int batch_num) {
    if (B < 0 || H < 32 || W < 32 || batch_num < 1) return 0;
    const int64_t NP = (int64_t)batch_num * (H / 32) * (W / 32);
    const int64_t per = (int64_t)(H / 32) * 4 * (W / 32) * 4 * 9;
    return scan_bytes(NP) + (size_t)(((NP + 3) & ~3ll) + ((B + 4) & ~3ll) + batch_num * per) * sizeof(int32_t) + 64;
}

extern "C" int pats_merge_patches(int merge_new, int64_t B, float* trust_score, int H, int W, int batch_num,
                                  const uint8_t* if_nomatching1_L1, uint8_t* if_nomatching1_L2,
                                  double* scores_back, uint8_t* out, void* workspace, size_t workspace_bytes,
                                  pats_stream_t stream) {
    PATS_REQUIRE(B >= 0 && H >= 32 && W >= 32 && batch_num >= 1, "merge_patches: bad shape");
    if (B == 0) return PATS_OK;
    PATS_REQUIRE(trust_score && if_nomatching1_L1 && if_nomatching1_L2 && scores_back && out, "merge_patches: null pointer");
    PATS_REQUIRE(workspace && workspace_bytes >= pats_merge_workspace_bytes(B, H, W, batch_num),
                 "merge_patches: workspace too small");
    hipStream_t st = as_stream(stream);
    MergeGeom g{H / 32, W / 32, 4 * (H / 32), 4 * (W / 32), (int64_t)(H / 32) * 4 * (W / 32) * 4 * 9};
    const int64_t NP = (int64_t)batch_num * g.h * g.w;
    char* ws = static_cast<char*>(workspace);
    Scan sc;
    int rc = run_scan(if_nomatching1_L1, NP, nullptr, &ws, &sc, st);
    if (rc != PATS_OK) return rc;
    int32_t* slot = reinterpret_cast<int32_t*>(ws);
    int32_t* patch_of = slot + ((NP + 3) & ~3ll);
    unsigned* winner = reinterpret_cast<unsigned*>(patch_of + ((B + 4) & ~3ll));
    if (hipMemsetAsync(patch_of, 0xff, sizeof(int32_t) * (size_t)B, st) != hipSuccess) return check_launch("merge memset");
    hipLaunchKernelGGL(merge_slots_kernel, dim3(blocks256(NP)), dim3(256), 0, st, if_nomatching1_L1, sc, NP, B, slot, patch_of);
    const RowBlock rb{nullptr, B};
    hipLaunchKernelGGL(merge_prepare_kernel, dim3(blocks256(B * 144)), dim3(256), 0, st, merge_new, rb, trust_score,
                       if_nomatching1_L2, patch_of, scores_back);
    if (merge_new) {
        hipLaunchKernelGGL(merge_select_new_kernel, dim3(blocks256(B * 144)), dim3(256), 0, st, g, rb, patch_of,
                           if_nomatching1_L2, scores_back, (const uint8_t*)nullptr, out);
    } else {
        if (hipMemsetAsync(winner, 0, sizeof(unsigned) * (size_t)(batch_num * g.per), st) != hipSuccess)
            return check_launch("merge memset");
        hipLaunchKernelGGL(merge_scatter_old_kernel, dim3(blocks256((int64_t)g.h4 * g.w4 * batch_num)), dim3(256), 0, st, g,
                           batch_num, slot, if_nomatching1_L2, scores_back, winner);
        hipLaunchKernelGGL(merge_finish_old_kernel, dim3(blocks256(B * 144)), dim3(256), 0, st, g, rb, patch_of, winner,
                           (const uint8_t*)nullptr, out);
    }
    return check_launch("merge_patches");
}

// ---- the merges of a batch of pairs: chunk blocks in order, every block over all pairs (row table of batch.hip) ----
extern "C" size_t pats_merge_batch_workspace_bytes(int64_t pairs, int H, int W) {
    if (pairs < 1 || H < 32 || W < 32) return 0;
    return (size_t)(pairs * (int64_t)(H / 32) * 4 * (W / 32) * 4 * 9) * sizeof(unsigned) + 64;
}

extern "C" int pats_merge_patches_batch(int merge_new, int Cmax, int64_t pairs, int H, int W, int64_t rows_cap,
                                        const int64_t* chunk_base, const int32_t* row_cell, const int32_t* row_slot,
                                        const uint8_t* row_forced, float* trust_score, uint8_t* if_nomatching1_L2,
                                        double* scores_back, int zero_scores_back, uint8_t* out, void* workspace,
                                        size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(Cmax >= 1 && pairs >= 0 && H >= 32 && W >= 32 && rows_cap >= 0, "merge_patches_batch: bad shape");
    if (pairs == 0 || rows_cap == 0) return PATS_OK;
    PATS_REQUIRE(chunk_base && row_cell && row_slot && row_forced && trust_score && if_nomatching1_L2 && scores_back && out,
                 "merge_patches_batch: null pointer");
    PATS_REQUIRE(merge_new || (workspace && workspace_bytes >= pats_merge_batch_workspace_bytes(pairs, H, W)),
                 "merge_patches_batch: workspace too small");
    hipStream_t st = as_stream(stream);
    MergeGeom g{H / 32, W / 32, 4 * (H / 32), 4 * (W / 32), (int64_t)(H / 32) * 4 * (W / 32) * 4 * 9};
    const int64_t NP = pairs * g.h * g.w;
    const int64_t block_cap = NP < rows_cap ? NP : rows_cap;       // a chunk block holds at most one row per coarse cell
    unsigned* winner = reinterpret_cast<unsigned*>(workspace);
    // merge_new: two parallel launches for all chunks (merge_select_new_par_kernel); PATS_MERGE_PER_CHUNK=1 (diagnostic library): the
    // launch chain per chunk of rounds 3-5.  merge_old (indoor: one chunk, cap 512) keeps its chain.
    static const bool per_chunk = [] { const char* e = diag_env("PATS_MERGE_PER_CHUNK"); return e && atoi(e) != 0; }();
    if (merge_new && !per_chunk) {
        const MergeTable tb{Cmax, 0, Cmax, pairs, 0, rows_cap, chunk_base, row_cell, row_slot, row_forced};
        hipLaunchKernelGGL(merge_select_new_par_kernel, dim3(blocks256(rows_cap * 144)), dim3(256), 0, st, g, tb, trust_score, if_nomatching1_L2,
                           scores_back, zero_scores_back, out);
        const int64_t n2 = rows_cap > NP ? rows_cap : NP;
        hipLaunchKernelGGL(merge_prepare_new_par_kernel, dim3(blocks256(n2 * 144)), dim3(256), 0, st, g, tb, trust_score, if_nomatching1_L2,
                           scores_back, zero_scores_back, out);
        return check_launch("merge_patches_batch");
    }
    // rows outside every block (padding past the total) are never visited: "no match"
    if (fill_bytes(out, 1, (size_t)rows_cap * 144, st)) return PATS_ERR_LAUNCH;
    // pats.py:32: every pair starts from a zeroed scores_back
    if (zero_scores_back && fill_bytes(scores_back, 0, sizeof(double) * (size_t)(NP * 144), st)) return PATS_ERR_LAUNCH;
    for (int c = 0; c < Cmax; ++c) {
        const RowBlock rb{chunk_base + c, 0};
        hipLaunchKernelGGL(merge_prepare_kernel, dim3(blocks256(block_cap * 144)), dim3(256), 0, st, merge_new, rb, trust_score,
                           if_nomatching1_L2, row_cell, scores_back);
        if (merge_new) {
            hipLaunchKernelGGL(merge_select_new_kernel, dim3(blocks256(block_cap * 144)), dim3(256), 0, st, g, rb, row_cell,
                               if_nomatching1_L2, scores_back, row_forced, out);
        } else {
            if (fill_bytes(winner, 0, sizeof(unsigned) * (size_t)(pairs * g.per), st)) return PATS_ERR_LAUNCH;
            hipLaunchKernelGGL(merge_scatter_old_kernel, dim3(blocks256((int64_t)g.h4 * g.w4 * pairs)), dim3(256), 0, st, g,
                               (int)pairs, row_slot + (int64_t)c * NP, if_nomatching1_L2, scores_back, winner);
            hipLaunchKernelGGL(merge_finish_old_kernel, dim3(blocks256(block_cap * 144)), dim3(256), 0, st, g, rb, row_cell, winner,
                               row_forced, out);
            // merge_patches_old hands back a zeroed scores_back (second_layer.py:191): the next chunk starts from zeros
            if (fill_bytes(scores_back, 0, sizeof(double) * (size_t)(NP * 144), st)) return PATS_ERR_LAUNCH;
        }
    }
    return check_launch("merge_patches_batch");
}

// Chunks [c_lo, c_hi) of the table only, on tensors that hold table rows row_origin .. row_origin + rows_local (trust_score,
// if_nomatching1_L2, out [rows_local,144]): PATS.forward's chunk loop (pats.py:33-37) walked chunk by chunk with the device-side
// table - one launch per chunk, scores_back handed from call to call (zero_scores_back on the first).  row_forced applies
// pats.py:38-39.  No reference counterpart beyond merge_patches_new / _old themselves (second_layer.py:137-238).
extern "C" int pats_merge_patches_chunks(int merge_new, int Cmax, int c_lo, int c_hi, int64_t pairs, int H, int W, int64_t row_origin,
                                         int64_t rows_local, const int64_t* chunk_base, const int32_t* row_cell,
                                         const int32_t* row_slot, const uint8_t* row_forced, float* trust_score,
                                         uint8_t* if_nomatching1_L2, double* scores_back, int zero_scores_back, uint8_t* out,
                                         void* workspace, size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(Cmax >= 1 && 0 <= c_lo && c_lo <= c_hi && c_hi <= Cmax && pairs >= 0 && pairs <= 65535 && H >= 32 && W >= 32 &&
                     row_origin >= 0 && rows_local >= 0, "merge_patches_chunks: bad shape");
    if (pairs == 0 || rows_local == 0 || c_lo == c_hi) return PATS_OK;
    PATS_REQUIRE(chunk_base && row_cell && row_slot && row_forced && trust_score && if_nomatching1_L2 && scores_back && out,
                 "merge_patches_chunks: null pointer");
    PATS_REQUIRE(merge_new || (workspace && workspace_bytes >= pats_merge_batch_workspace_bytes(pairs, H, W)),
                 "merge_patches_chunks: workspace too small");
    MergeGeom g{H / 32, W / 32, 4 * (H / 32), 4 * (W / 32), (int64_t)(H / 32) * 4 * (W / 32) * 4 * 9};
    hipStream_t st = as_stream(stream);
    const int64_t NP = pairs * g.h * g.w;
    if (merge_new) {
        const MergeTable tb{Cmax, c_lo, c_hi, pairs, row_origin, rows_local, chunk_base, row_cell, row_slot, row_forced};
        hipLaunchKernelGGL(merge_select_new_par_kernel, dim3(blocks256(rows_local * 144)), dim3(256), 0, st, g, tb, trust_score,
                           if_nomatching1_L2, scores_back, zero_scores_back, out);
        const int64_t n2 = zero_scores_back && NP > rows_local ? NP : rows_local;
        hipLaunchKernelGGL(merge_prepare_new_par_kernel, dim3(blocks256(n2 * 144)), dim3(256), 0, st, g, tb, trust_score, if_nomatching1_L2,
                           scores_back, zero_scores_back, out);
        return check_launch("merge_patches_chunks");
    }
    // merge_old: the per-chunk chain on pointers shifted to table rows
    unsigned* winner = reinterpret_cast<unsigned*>(workspace);
    float* tr = trust_score - row_origin * 144;
    uint8_t* fl = if_nomatching1_L2 - row_origin * 144;
    uint8_t* ou = out - row_origin * 144;
    if (fill_bytes(out, 1, (size_t)rows_local * 144, st)) return PATS_ERR_LAUNCH;
    if (zero_scores_back && fill_bytes(scores_back, 0, sizeof(double) * (size_t)(NP * 144), st)) return PATS_ERR_LAUNCH;
    const int64_t block_cap = NP < rows_local ? NP : rows_local;
    for (int c = c_lo; c < c_hi; ++c) {
        const RowBlock rb{chunk_base + c, 0};
        hipLaunchKernelGGL(merge_prepare_kernel, dim3(blocks256(block_cap * 144)), dim3(256), 0, st, 0, rb, tr, fl, row_cell, scores_back);
        if (fill_bytes(winner, 0, sizeof(unsigned) * (size_t)(pairs * g.per), st)) return PATS_ERR_LAUNCH;
        hipLaunchKernelGGL(merge_scatter_old_kernel, dim3(blocks256((int64_t)g.h4 * g.w4 * pairs)), dim3(256), 0, st, g, (int)pairs,
                           row_slot + (int64_t)c * NP, fl, scores_back, winner);
        hipLaunchKernelGGL(merge_finish_old_kernel, dim3(blocks256(block_cap * 144)), dim3(256), 0, st, g, rb, row_cell, winner, row_forced, ou);
        if (fill_bytes(scores_back, 0, sizeof(double) * (size_t)(NP * 144), st)) return PATS_ERR_LAUNCH;
    }
    return check_launch("merge_patches_chunks");
}

// ---- matches of a batch, grouped by pair (throughput mode's hand-over: what batch.split_by_pair did with argsort + bincount) ----
// pats_get_result_chunks_f32 emits the matches in (chunk, pair, patch, sub-cell) order: the matches of one (chunk, pair) are a
// contiguous run, a pair's list in the reference's order is the concatenation of its runs over the chunks.  Three small launches:
// run boundaries (one thread per match looks at its neighbour), destination offsets (one workgroup, pairs x Cmax entries), copy.
namespace pats {
struct ByPairArgs {
    const float* ml; const float* mr; const int32_t* match_row; const int64_t* M; const int32_t* row_cell; const int64_t* chunk_base;
    int Cmax; int64_t pairs; int N;
    float* out_l; float* out_r; int64_t* pair_off; int64_t* seg_lo; int64_t* seg_hi; int64_t* seg_dst;
    const int64_t* P_dev; const int32_t* status;      // optional: the step's summary behind the offsets (pats_matches_by_pair_summary_f32)
};
__device__ __forceinline__ int64_t bypair_key(const ByPairArgs& g, int64_t i) {
    const int64_t row = g.match_row[i];
    int c = 0;
    for (int k = 1; k < g.Cmax; ++k) c += (row >= g.chunk_base[k]) ? 1 : 0;            // chunk blocks are ascending
    return (int64_t)c * g.pairs + g.row_cell[row] / g.N;
}
__global__ void __launch_bounds__(256) bypair_runs_kernel(ByPairArgs g) {
    const int64_t M = *g.M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const int64_t k = bypair_key(g, i);
        if (i == 0 || bypair_key(g, i - 1) != k) g.seg_lo[k] = i;
        if (i == M - 1 || bypair_key(g, i + 1) != k) g.seg_hi[k] = i + 1;
    }
}
__global__ void __launch_bounds__(256) bypair_offsets_kernel(ByPairArgs g) {              // one workgroup; thread p walks pair p's chunks
    __shared__ int64_t total[1024];          // in: a pair's match count; after thread 0's scan: its offset (the hand-over between the
    __shared__ int64_t carry;                // waves stays in LDS - pair_off in global memory is an OUTPUT here, never read back)
    const int64_t P = g.pairs;
    if (threadIdx.x == 0) carry = 0;
    for (int64_t p0 = 0; p0 < P; p0 += 1024) {                                            // (pairs per batch: tens)
        for (int64_t p = p0 + threadIdx.x; p < P && p < p0 + 1024; p += 256) {
            int64_t t = 0;
            for (int c = 0; c < g.Cmax; ++c) t += g.seg_hi[c * P + p] - g.seg_lo[c * P + p];
            total[p - p0] = t;
        }
        wg_barrier();
        if (threadIdx.x == 0) {
            int64_t run = carry;
            for (int64_t p = p0; p < P && p < p0 + 1024; ++p) { const int64_t t = total[p - p0]; total[p - p0] = run; g.pair_off[p] = run; run += t; }
            g.pair_off[P < p0 + 1024 ? P : p0 + 1024] = run;
            carry = run;
            if (g.status && p0 + 1024 >= P) {             // the hand-over's counters in the same buffer: ONE device-to-host copy per step
                g.pair_off[P + 1] = *g.M;
                g.pair_off[P + 2] = g.P_dev ? *g.P_dev : 0;
                g.pair_off[P + 3] = (int64_t)*g.status;
            }
        }
        wg_barrier();
        for (int64_t p = p0 + threadIdx.x; p < P && p < p0 + 1024; p += 256) {
            int64_t run = total[p - p0];
            for (int c = 0; c < g.Cmax; ++c) { g.seg_dst[c * P + p] = run; run += g.seg_hi[c * P + p] - g.seg_lo[c * P + p]; }
        }
        wg_barrier();
    }
}
__global__ void __launch_bounds__(256) bypair_copy_kernel(ByPairArgs g) {
    const int64_t M = *g.M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const int64_t k = bypair_key(g, i), d = g.seg_dst[k] + (i - g.seg_lo[k]);
        reinterpret_cast<float2*>(g.out_l)[d] = reinterpret_cast<const float2*>(g.ml)[i];
        reinterpret_cast<float2*>(g.out_r)[d] = reinterpret_cast<const float2*>(g.mr)[i];
    }
}
}  // namespace pats

extern "C" size_t pats_matches_by_pair_workspace_bytes(int Cmax, int64_t pairs) {
    return Cmax > 0 && pairs > 0 ? (size_t)3 * Cmax * pairs * sizeof(int64_t) : 0;
}

static int matches_by_pair_impl(const float* matches_l, const float* matches_r, const int32_t* match_row, const int64_t* M_dev,
                                const int32_t* row_cell, const int64_t* chunk_base, int Cmax, int64_t pairs, int N,
                                float* out_l, float* out_r, int64_t* pair_off, const int64_t* P_dev, const int32_t* status,
                                void* workspace, size_t workspace_bytes, pats_stream_t stream);
extern "C" int pats_matches_by_pair_f32(const float* matches_l, const float* matches_r, const int32_t* match_row, const int64_t* M_dev,
                                        const int32_t* row_cell, const int64_t* chunk_base, int Cmax, int64_t pairs, int N,
                                        float* out_l, float* out_r, int64_t* pair_off, void* workspace, size_t workspace_bytes,
                                        pats_stream_t stream) {
    return matches_by_pair_impl(matches_l, matches_r, match_row, M_dev, row_cell, chunk_base, Cmax, pairs, N, out_l, out_r, pair_off, nullptr,
                                nullptr, workspace, workspace_bytes, stream);
}
// The same with the step's counters appended: pair_off has pairs + 4 entries - the pairs + 1 offsets, then M, P (*P_dev, the
// third-level problem count; may be null -> 0) and the row table's status - so that a batch's hand-over is ONE device-to-host copy.
extern "C" int pats_matches_by_pair_summary_f32(const float* matches_l, const float* matches_r, const int32_t* match_row,
                                                const int64_t* M_dev, const int32_t* row_cell, const int64_t* chunk_base, int Cmax,
                                                int64_t pairs, int N, float* out_l, float* out_r, int64_t* pair_off,
                                                const int64_t* P_dev, const int32_t* status, void* workspace, size_t workspace_bytes,
                                                pats_stream_t stream) {
    PATS_REQUIRE(status, "matches_by_pair_summary: null status");
    return matches_by_pair_impl(matches_l, matches_r, match_row, M_dev, row_cell, chunk_base, Cmax, pairs, N, out_l, out_r, pair_off, P_dev,
                                status, workspace, workspace_bytes, stream);
}
static int matches_by_pair_impl(const float* matches_l, const float* matches_r, const int32_t* match_row, const int64_t* M_dev,
                                const int32_t* row_cell, const int64_t* chunk_base, int Cmax, int64_t pairs, int N,
                                float* out_l, float* out_r, int64_t* pair_off, const int64_t* P_dev, const int32_t* status,
                                void* workspace, size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(Cmax >= 1 && pairs >= 1 && N >= 1, "matches_by_pair: bad shape");
    PATS_REQUIRE(matches_l && matches_r && match_row && M_dev && row_cell && chunk_base && out_l && out_r && pair_off,
                 "matches_by_pair: null pointer");
    PATS_REQUIRE(workspace && workspace_bytes >= pats_matches_by_pair_workspace_bytes(Cmax, pairs), "matches_by_pair: workspace too small");
    hipStream_t st = as_stream(stream);
    int64_t* ws = reinterpret_cast<int64_t*>(workspace);
    const int64_t nseg = (int64_t)Cmax * pairs;
    if (fill_bytes(ws, 0, sizeof(int64_t) * (size_t)(2 * nseg), st)) return PATS_ERR_LAUNCH;        // runs without matches: lo = hi = 0
    ByPairArgs g{matches_l, matches_r, match_row, M_dev, row_cell, chunk_base, Cmax, pairs, N, out_l, out_r, pair_off, ws, ws + nseg,
                 ws + 2 * nseg, P_dev, status};
    hipLaunchKernelGGL(bypair_runs_kernel, dim3(2048), dim3(256), 0, st, g);
    hipLaunchKernelGGL(bypair_offsets_kernel, dim3(1), dim3(256), 0, st, g);
    hipLaunchKernelGGL(bypair_copy_kernel, dim3(2048), dim3(256), 0, st, g);
    return check_launch("matches_by_pair kernels");
}

extern "C" size_t pats_compact_workspace_bytes(int64_t n) { return n < 0 ? 0 : scan_bytes(n) + 64; }

extern "C" int pats_third_inputs_f32(const uint8_t* if_nomatching, const float* pts, int64_t B, float* mkpts0,
                                     float* mkpts1, int64_t* b_ids, int64_t capacity, int64_t* count,
                                     void* workspace, size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(B >= 0 && capacity >= 0, "third_inputs: bad shape");
    PATS_REQUIRE(count, "third_inputs: null count");
    const int64_t n = B * 144;
    PATS_REQUIRE(n == 0 || (if_nomatching && pts && (capacity == 0 || (mkpts0 && mkpts1 && b_ids))), "third_inputs: null pointer");
    PATS_REQUIRE(workspace && workspace_bytes >= pats_compact_workspace_bytes(n), "third_inputs: workspace too small");
    char* ws = static_cast<char*>(workspace);
    Scan sc;
    int rc = run_scan(if_nomatching, n, count, &ws, &sc, as_stream(stream));
    if (rc != PATS_OK || n == 0) return rc;
    hipLaunchKernelGGL(third_inputs_kernel, dim3(blocks256(n)), dim3(256), 0, as_stream(stream), if_nomatching, pts, n, sc,
                       capacity, mkpts0, mkpts1, b_ids);
    return check_launch("third_inputs_kernel");
}

extern "C" int pats_refine_scatter_f32(const uint8_t* if_nomatching, const float* pts, const float* mkpts1_f,
                                       const float* label, int label_stride, int64_t B, int64_t P,
                                       uint8_t* if_nomatching16, float* pts16, void* workspace,
                                       size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(B >= 0 && P >= 0 && label_stride >= 1, "refine_scatter: bad shape");
    if (B == 0) return PATS_OK;
    PATS_REQUIRE(if_nomatching && pts && if_nomatching16 && pts16 && (P == 0 || (mkpts1_f && label)), "refine_scatter: null pointer");
    const int64_t n = B * 144;
    PATS_REQUIRE(workspace && workspace_bytes >= pats_compact_workspace_bytes(n), "refine_scatter: workspace too small");
    char* ws = static_cast<char*>(workspace);
    Scan sc;
    int rc = run_scan(if_nomatching, n, nullptr, &ws, &sc, as_stream(stream));
    if (rc != PATS_OK) return rc;
    hipLaunchKernelGGL(refine_scatter_kernel, dim3(blocks256(B * 2304)), dim3(256), 0, as_stream(stream), if_nomatching, pts,
                       mkpts1_f, label, label_stride, P, B * 2304, sc, if_nomatching16, pts16);
    return check_launch("refine_scatter_kernel");
}

extern "C" size_t pats_get_result_workspace_bytes(int64_t cells0, int64_t rows1, int64_t cells1) {
    if (cells0 < 0 || rows1 < 0 || cells1 < 0) return 0;
    return scan_bytes(cells0) + scan_bytes(rows1 * cells1) + (size_t)((rows1 + 4) & ~3ll) * sizeof(int32_t) + 64;
}

static int get_result_impl(int batch_size, const uint8_t* if_nomatching0, const uint8_t* if_nomatching1,
                                   int64_t rows1, const float* average_point0, const float* average_point1,
                                   const float* scale0, const float* scale1, int64_t scale1_cell_stride,
                                   const int* patch_size0, const int* patch_size1, const uint8_t* left_choice0,
                                   const uint8_t* left_choice1, float* matches_l, float* matches_r,
                                   int64_t capacity, int64_t* count, void* workspace, size_t workspace_bytes,
                                   pats_stream_t stream, int64_t period0, float ap0_div, float ap1_div, int32_t* match_row) {
    PATS_REQUIRE(batch_size >= 1 && rows1 >= 0 && capacity >= 0 && patch_size0 && patch_size1, "get_result: bad shape");
    PATS_REQUIRE(scale1_cell_stride == 0 || scale1_cell_stride == 2, "get_result: scale1_cell_stride must be 0 or 2");
    PATS_REQUIRE(count, "get_result: null count");
    const int64_t n0 = (int64_t)patch_size0[1] * patch_size0[2], n1 = (int64_t)patch_size1[1] * patch_size1[2];
    PATS_REQUIRE(n0 > 0 && n1 > 0 && patch_size0[0] > 0 && patch_size1[0] > 0, "get_result: bad patch_size");
    const int64_t cells0 = batch_size * n0;
    PATS_REQUIRE(workspace && workspace_bytes >= pats_get_result_workspace_bytes(cells0, rows1, n1), "get_result: workspace too small");
    PATS_REQUIRE(if_nomatching0 && average_point0 && scale0 && left_choice0, "get_result: null pointer");
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    Scan s0, s1;
    int64_t* rows0_dev = reinterpret_cast<int64_t*>(ws);     // 64 B header
    ws += 64;
    int rc = run_scan(if_nomatching0, cells0, rows0_dev, &ws, &s0, st);
    if (rc != PATS_OK) return rc;
    rc = run_scan(if_nomatching1, rows1 * n1, count, &ws, &s1, st);
    if (rc != PATS_OK || rows1 == 0) return rc;
    PATS_REQUIRE(if_nomatching1 && average_point1 && (scale1 || period0) && left_choice1 && (capacity == 0 || (matches_l && matches_r)),
                 "get_result: null pointer");
    int32_t* row_cell = reinterpret_cast<int32_t*>(ws);
    hipLaunchKernelGGL(result_rows_kernel, dim3(blocks256(cells0)), dim3(256), 0, st, if_nomatching0, cells0, s0, rows1, row_cell);
    ResultArgs g{if_nomatching1, average_point0, average_point1, scale0, scale1, scale1_cell_stride,
                 patch_size0[0], patch_size0[1], patch_size0[2], patch_size1[0], patch_size1[1], patch_size1[2],
                 left_choice0, left_choice1, rows0_dev, rows1, capacity, matches_l, matches_r, period0, ap0_div, ap1_div,
                 match_row};
    const int64_t blocks = rows1 * (((int64_t)n1 + 255) / 256);
    PATS_REQUIRE(blocks < (1ll << 31), "get_result: grid too large (split the batch)");
    hipLaunchKernelGGL(get_result_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, row_cell, s1);
    return check_launch("get_result_kernel");
}

extern "C" int pats_get_result_f32(int batch_size, const uint8_t* if_nomatching0, const uint8_t* if_nomatching1,
                                   int64_t rows1, const float* average_point0, const float* average_point1,
                                   const float* scale0, const float* scale1, int64_t scale1_cell_stride,
                                   const int* patch_size0, const int* patch_size1, const uint8_t* left_choice0,
                                   const uint8_t* left_choice1, float* matches_l, float* matches_r,
                                   int64_t capacity, int64_t* count, void* workspace, size_t workspace_bytes,
                                   pats_stream_t stream) {
    return get_result_impl(batch_size, if_nomatching0, if_nomatching1, rows1, average_point0, average_point1, scale0, scale1,
                           scale1_cell_stride, patch_size0, patch_size1, left_choice0, left_choice1, matches_l, matches_r,
                           capacity, count, workspace, workspace_bytes, stream, 0, 0.0f, 0.0f, nullptr);
}

// get_result of a batch of pairs in one call (pats.py:68-73 for every (chunk, pair) at once): level-0 batch = the
// Cmax * pairs chunk masks of batch.hip, level-1 rows = the row table; pts_new / scales are the PER-PAIR tensors
// [pairs, N, 2] (never expanded over the chunks), points un-flipped; match_row tells which row - hence which pair -
// every match belongs to.
extern "C" int pats_get_result_chunks_f32(int Cmax, int64_t pairs, const uint8_t* masks, const uint8_t* if_nomatching16,
                                          int64_t rows_cap, const float* pts_new, const float* pts16, const float* scales,
                                          const int* patch_size0, const int* patch_size1, const uint8_t* left_choice0,
                                          const uint8_t* left_choice1, float* matches_l, float* matches_r,
                                          int32_t* match_row, int64_t capacity, int64_t* count, void* workspace,
                                          size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(Cmax >= 1 && pairs >= 1 && patch_size0, "get_result_chunks: bad shape");
    PATS_REQUIRE((int64_t)Cmax * pairs < (1ll << 31), "get_result_chunks: batch too large");
    const int64_t period0 = pairs * (int64_t)patch_size0[1] * patch_size0[2];
    return get_result_impl((int)(Cmax * pairs), masks, if_nomatching16, rows_cap, pts_new, pts16, scales, nullptr, 0,
                           patch_size0, patch_size1, left_choice0, left_choice1, matches_l, matches_r, capacity, count,
                           workspace, workspace_bytes, stream, period0, 32.0f, 2.0f, match_row);
}
