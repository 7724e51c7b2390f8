// Patch-area expansion ("subdivision" geometry) for PATS on gfx950.
//
// Replaces Iterative_expand_matrix (utils/utils.py:1179-1297) + Compute_scaling (:1321-1340) +
// Compute_positions_and_ranges (:1527-1537).  The reference runs ~40 small ATen ops per growth
// iteration over index tensors [b, m, 4*width] and materialises a [hw, b, m, 2] ruler.
//
// Here one 16-lane DPP row owns one source patch (four patches per wave, sixteen per workgroup):
// the patch's row of exp(Z) is staged once in LDS (exp fused into the load when handed the log
// plan), lane t of the group gathers cell t (t+16, ...) of each of the four adjacent strips, and
// every sum / argmax is a 4-step row-level DPP reduction whose result lands in all 16 lanes - no
// cross-row traffic, no LDS shuffles; the growth decision is then taken redundantly per lane.
// Every quirk of the reference is kept literally (each has a fixture):
//   * sentinel index width*height+1 reading the appended 1e-14 (:1205-1208,1220-1221)
//   * `ranges` rows padded with 1e7 so out-of-span strip cells overflow to the sentinel (:1536)
//   * the left/right strips wrapping into the neighbouring grid row (only the SUM is zeroed at
//     the border, :1227-1230)
//   * the post-loop border sums reusing the LAST iteration's sequence_base (:1245-1249)
//   * width = max(h, w), height = h*w / width as the function derives them (:1181)
//   * if_nomatching compared against the ROW count (:1191)
#include "common.hpp"
#include "third_device.hpp"     // row16_sum / row16_argmax

namespace pats {

struct ExpandArgs {
    const float* P;
    int input_is_log;
    int64_t rows_total;   // batch * m
    int M, N;
    const float* scalex;
    const float* scaley;
    int lim3, h, w;
    float lower_bound;
    int iter_num;
    float *whole_cost, *core_cost, *average_point, *x_scale, *y_scale;
    int64_t* bound;
    uint8_t* row_nomatch;     // optional: scores.max(2).indices == N - 1 of the INPUT values (first_layer.py:162-164)
    const int64_t* live;      // optional: device-side batch count (counted launch); rows of problems >= *live are skipped
};

__device__ __forceinline__ float range_val(int d, int k) { return k <= d ? (float)k : 1e7f; }

// float -> long truncation, then anything outside [0, wh - 1] becomes the sentinel (:1220-1221).  Every f that reaches
// here is an integral float (an integer or the 1e7 padding, times the width, plus an integer offset), so the range
// test can be made on the float itself - no 64-bit conversion.
__device__ __forceinline__ int clamp_seq(float f, int wh, int S) {
    return (f >= 0.f && f <= (float)(wh - 1)) ? (int)f : S;
}


// SMALL: the grid is at most 16 wide (the fine level's 12 x 12), every strip is ONE cell per lane - no inner loops
template <bool SMALL>
__global__ void __launch_bounds__(256)
expand_kernel(ExpandArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int t = threadIdx.x & 15, slot = threadIdx.x >> 4;      // lane in group, group in workgroup
    const int ROWS_PER_WG = blockDim.x >> 4;                      // 16 (256 threads) or 4 for wide rows
    const int M = a.M, N = a.N, m = M - 1, n = N - 1;
    int64_t rows_total = a.rows_total;
    if (a.live) {                                                 // counted launch: the rows of the first *live problems
        const int64_t lv = *a.live * m;
        rows_total = lv < rows_total ? lv : rows_total;
        if ((int64_t)blockIdx.x * ROWS_PER_WG >= rows_total) return;      // workgroup-uniform
    }
    const int64_t gr_raw = (int64_t)blockIdx.x * ROWS_PER_WG + slot;
    const bool active = gr_raw < rows_total;
    const int64_t gr = active ? gr_raw : rows_total - 1;         // idle groups shadow the last row
    const int64_t b = gr / m;
    const int r = (int)(gr - b * m);
    const int ld = N + 3;
    float* prow = sm + (size_t)slot * ld;
    float* popp = sm + (size_t)ROWS_PER_WG * ld + (size_t)slot * ld;   // this batch's dustbin row (see below)
    const float* Pb = a.P + b * (int64_t)M * N;
    const float* src = Pb + (int64_t)r * N;
    const float* oppsrc = Pb + (int64_t)(M - 1) * N;              // scores_in[:, -1, :-1]   (:1183)
    const float* sx = a.scalex + b * (int64_t)n;
    const float* sy = a.scaley + b * (int64_t)n;
    // the caller's "no match" flag is the argmax of the values as handed in (the LOG plan when input_is_log:
    // exp can round two distinct logs to one float, so the flag must not come from the exponentiated row)
    float fv = -INFINITY;
    int fi = 0x7fffffff;
    // The dustbin row of the problem is the same for its m source rows: a workgroup's rows touch at most two problems
    // (when m >= rows per workgroup), so the whole workgroup stages those two rows ONCE (the first version had every
    // 16-lane group exponentiate its own copy: 16 x N exps per workgroup instead of 2 x N).
    const int64_t g0 = (int64_t)blockIdx.x * ROWS_PER_WG;
    const int64_t glast = min(g0 + ROWS_PER_WG - 1, rows_total - 1);
    const int64_t b0 = g0 / m;
    const bool shared_opp = glast / m - b0 <= 1;                  // workgroup-uniform
    if (shared_opp) {
        popp = sm + (size_t)ROWS_PER_WG * ld + (size_t)(b - b0) * ld;
        for (int e = threadIdx.x; e < 2 * N; e += blockDim.x) {
            const int which = e >= N, j = e - which * N;
            const int64_t bb = min(b0 + which, (int64_t)(rows_total - 1) / m);
            const float o = a.P[bb * (int64_t)M * N + (int64_t)(M - 1) * N + j];
            sm[(size_t)ROWS_PER_WG * ld + (size_t)which * ld + j] = a.input_is_log ? expf(o) : o;
        }
    }
    // the same pass takes the argmax over the real columns (:1182) and over all columns (:1191), first index on ties,
    // and the row sum (:1288) of the values it stages: lane t sees cells t, t + 16, ... in the order a second pass would
    float bv = -INFINITY, bva = -INFINITY;
    int bi = 0x7fffffff, bia = 0x7fffffff;
    float rowsum = 0.f;
    for (int j = t; j < N; j += 16) {
        const float x0 = src[j];
        if (x0 > fv || fi == 0x7fffffff) { fv = x0; fi = j; }
        const float x = a.input_is_log ? expf(x0) : x0;
        prow[j] = x;
        rowsum += x;
        if (j < N - 1 && (x > bv || bi == 0x7fffffff)) { bv = x; bi = j; }
        if (x > bva || bia == 0x7fffffff) { bva = x; bia = j; }
        if (!shared_opp) {
            const float o = oppsrc[j];
            popp[j] = a.input_is_log ? expf(o) : o;
        }
    }
    if (a.row_nomatch) {
        row16_argmax(fv, fi);
        if (t == 0 && active) a.row_nomatch[gr] = fi == N - 1;
    }
    // the sentinel slot: every strip index is in [0, wh - 1] or is S = wh + 1 = N, which reads the appended 1e-14 (:1205,1208)
    if (t == 0) { prow[N] = ZERO_F; popp[N] = ZERO_F; }           // (shared dustbin rows: several groups write the same value)
    wg_barrier();     // every thread gets here (idle groups shadow the last row)

    const int width = a.h > a.w ? a.h : a.w;
    const int height = (a.h * a.w) / width;
    const int wh = width * height, S = wh + 1;
    auto ES = [&](int idx) { return prow[idx]; };                                    // :1205
    auto ESC = [&](int idx) { return idx < n ? sx[idx] * sy[idx] : ZERO_F; };        // :1206-1207
    auto EOPP = [&](int idx) { return popp[idx]; };                                  // :1208

    row16_argmax(bv, bi);
    row16_argmax(bva, bia);
    const float the_scale = row16_sum(rowsum);                      // scores.sum(2)  (:1288)
    const int max0 = bi;
    const bool if_nomatching = (bia == m);
    float last_nomatching = EOPP(max0);                             // :1184
    int up = max0 / a.lim3, down = up, left = max0 % a.lim3, right = left;   // :1189-1197
    int bd0 = 0, bd1 = 0, sb0 = 0, sb1 = 0;
    float last_sum = ES(max0);                                      // :1209

    for (int it = 0; it < a.iter_num; ++it) {
        sb0 = bd0; sb1 = bd1;                                       // :1215
        const float off[4] = {(float)(left + up * width - width),   // :1217
                              (float)(left + down * width + width), // :1218
                              (float)(left + up * width - 1),       // :1219
                              (float)(right + up * width + 1)};     // :1220
        float es[4], nm[4], sc[4];
        float vcell[4];          // SMALL: this lane's cell of each strip, for the one nomatching sum that is consumed
        int scell[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float e = 0.f, q = 0.f, c = 0.f;
            auto cell = [&](int k) {
                const float f = d < 2 ? range_val(sb1, k) + off[d] : range_val(sb0, k) * (float)width + off[d];
                const int s = clamp_seq(f, wh, S);
                const float v = ES(s);
                e += v;
                if (SMALL) { vcell[d] = v; scell[d] = s; }
                else {
                    q += (v > a.lower_bound) ? EOPP(s) : ZERO_F;    // :1225
                    c += ESC(s);                                    // :1231 (never consumed, see below)
                }
            };
            if (SMALL) { vcell[d] = 0.f; scell[d] = S; if (t < width) cell(t); }
            else for (int k = t; k < width; k += 16) cell(k);
            es[d] = row16_sum(e);
            if (!SMALL) { nm[d] = row16_sum(q); sc[d] = row16_sum(c); }
        }
        if (up == 0) es[0] = ZERO_F;                                // :1227-1230
        if (down == height - 1) es[1] = ZERO_F;
        if (left == 0) es[2] = ZERO_F;
        if (right == width - 1) es[3] = ZERO_F;
        int arg = 0;
        float mx = es[0];
#pragma unroll
        for (int d = 1; d < 4; ++d)
            if (es[d] > mx) { mx = es[d]; arg = d; }                // :1232-1234
        float mnm;
        if (SMALL) {
            // the nomatching sum of the chosen strip only (the other three are never read, :1233): same cells, same order
            float v = vcell[0];
            int sidx = scell[0];
#pragma unroll
            for (int d = 1; d < 4; ++d)
                if (arg == d) { v = vcell[d]; sidx = scell[d]; }
            const float q = (t < width) ? ((v > a.lower_bound) ? EOPP(sidx) : ZERO_F) : 0.f;
            mnm = row16_sum(q);
        } else {
            mnm = nm[0];
#pragma unroll
            for (int d = 1; d < 4; ++d)
                if (arg == d) mnm = nm[d];
        }
        (void)sc;       // last_scale is accumulated by the reference (:1242) but never consumed
        float add_sum = ZERO_F, add_nm = ZERO_F;
        if (mx > a.lower_bound) {                                   // :1235-1238
            up -= (arg == 0);
            down += (arg == 1);
            left -= (arg == 2);
            right += (arg == 3);
            add_sum = mx;
            add_nm = mnm;
        }
        bd0 = down - up;                                            // :1239-1240
        bd1 = right - left;
        last_sum = last_sum + add_sum;                              // :1241
        last_nomatching = last_nomatching + add_nm;                 // :1243
    }
    const bool if_core_exist = (bd0 > 1) && (bd1 > 1);              // :1244

    // border strips of the final rectangle with the stale sequence_base (:1245-1253)
    float e4, s4;
    {
        const float eoff[4] = {(float)(left + up * width), (float)(left + down * width),
                               (float)(left + up * width), (float)(right + up * width)};
        float ed[4], sd[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float e = 0.f, c = 0.f;
            auto cell = [&](int k) {
                const float f = d < 2 ? range_val(sb1, k) + eoff[d]
                                      : range_val(sb0, k) * (float)width + eoff[d];
                const int s = clamp_seq(f, wh, S);
                e += ES(s);
                c += ESC(s);
            };
            if (SMALL) { if (t < width) cell(t); }
            else for (int k = t; k < width; k += 16) cell(k);
            ed[d] = row16_sum(e);
            sd[d] = row16_sum(c);
        }
        e4 = (ed[0] + ed[1]) + (ed[2] + ed[3]);
        s4 = (sd[0] + sd[1]) + (sd[2] + sd[3]);
    }

    // in-rectangle weights, centroid and scale (:1254-1273, Compute_scaling :1321-1340)
    float wx = 0.f, wy = 0.f, sumx = 0.f, sumy = 0.f, ws = 0.f, so = 0.f;
    // positions (:1528-1532) of cells t, t + 16, ... kept incrementally (one division for the whole loop); a 16-cell band that
    // no rectangle of the wave's four rows reaches adds exact zeros to every accumulator and is skipped as a whole
    const int q16 = 16 / a.w, r16 = 16 - q16 * a.w;
    int pyi = t / a.w, pxi = t - pyi * a.w;
    for (int p = t; p < n; p += 16, pyi += q16, pxi += r16) {
        if (pxi >= a.w) { pxi -= a.w; ++pyi; }
        const bool crit = pyi >= up && pyi <= down && pxi >= left && pxi <= right;
        if (!__any(crit)) continue;
        const float fx = sx[p], fy = sy[p];
        float ox = ZERO_F, oy = ZERO_F;
        if (crit) {                                                 // the rectangle is a few cells: most passes skip the sqrt and the two divisions
            const float root = sqrtf(prow[p] + 1e-7f);
            ox = root / fx;
            oy = root / fy;
        }
        wx += ox * (float)pxi;
        wy += oy * (float)pyi;
        sumx += ox;
        sumy += oy;
        const float o = ox * oy;
        ws += o * (fx * fy);
        so += o;
    }
    wx = row16_sum(wx); wy = row16_sum(wy); sumx = row16_sum(sumx); sumy = row16_sum(sumy);
    ws = row16_sum(ws); so = row16_sum(so);

    if (t == 0 && active) {
        // corners (:1279-1287)
        const int corner[4] = {up * width + left, up * width + right, down * width + left,
                               down * width + right};
        float cps = 0.f, css = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int s = corner[q];
            if (!(s >= 0)) s = S;
            if (!(s <= wh - 1)) s = S;
            cps += ES(s);
            css += ESC(s);
        }
        const float core_scale_sum = the_scale - s4 + css;          // :1289
        const float core_sum = last_sum - e4 + cps;                 // :1290
        a.core_cost[gr] = (if_core_exist && !if_nomatching)
                              ? fabsf((core_sum - core_scale_sum) / the_scale) : ZERO_F;
        a.whole_cost[gr] = if_nomatching ? ZERO_F
            : (fabsf(the_scale - last_sum) + last_nomatching / 4.0f) / the_scale;    // :1296
        a.average_point[gr * 2 + 1] = wx / sumx + 0.5f;             // :1270-1273
        a.average_point[gr * 2 + 0] = wy / sumy + 0.5f;
        const float average_scale = sqrtf(ws / so);                 // :1326
        a.x_scale[gr] = 1.0f / (average_scale / 1.0f);              // :1335-1340
        a.y_scale[gr] = 1.0f / (average_scale * 1.0f);
        a.bound[gr * 4 + 0] = up;
        a.bound[gr * 4 + 1] = down;
        a.bound[gr * 4 + 2] = left;
        a.bound[gr * 4 + 3] = right;
    }
}

}  // namespace pats

using namespace pats;

static int expand_impl(const float* P, int input_is_log, int64_t batch, int M, int N, const float* scalex, const float* scaley,
                       int lim3, int h, int w, float lower_bound, int iter_num, float* whole_cost, float* core_cost,
                       float* average_point, float* x_scale, float* y_scale, int64_t* bound, uint8_t* row_nomatch,
                       const int64_t* live, pats_stream_t stream);

extern "C" int pats_iterative_expand_f32(const float* P, int input_is_log, int64_t batch, int M,
                                         int N, const float* scalex, const float* scaley, int lim3,
                                         int h, int w, float lower_bound, int iter_num,
                                         float* whole_cost, float* core_cost, float* average_point,
                                         float* x_scale, float* y_scale, int64_t* bound,
                                         uint8_t* row_nomatch, pats_stream_t stream) {
    return expand_impl(P, input_is_log, batch, M, N, scalex, scaley, lim3, h, w, lower_bound, iter_num, whole_cost, core_cost,
                       average_point, x_scale, y_scale, bound, row_nomatch, nullptr, stream);
}

// launched over a capacity of batch_cap problems, *batch_dev of them in use (throughput mode); outputs of the others untouched
extern "C" int pats_iterative_expand_counted_f32(const float* P, int input_is_log, int64_t batch_cap, const int64_t* batch_dev,
                                                 int M, int N, const float* scalex, const float* scaley, int lim3, int h, int w,
                                                 float lower_bound, int iter_num, float* whole_cost, float* core_cost,
                                                 float* average_point, float* x_scale, float* y_scale, int64_t* bound,
                                                 uint8_t* row_nomatch, pats_stream_t stream) {
    PATS_REQUIRE(batch_dev, "iterative_expand_counted: null count");
    return expand_impl(P, input_is_log, batch_cap, M, N, scalex, scaley, lim3, h, w, lower_bound, iter_num, whole_cost, core_cost,
                       average_point, x_scale, y_scale, bound, row_nomatch, batch_dev, stream);
}

static int expand_impl(const float* P, int input_is_log, int64_t batch, int M, int N, const float* scalex, const float* scaley,
                       int lim3, int h, int w, float lower_bound, int iter_num, float* whole_cost, float* core_cost,
                       float* average_point, float* x_scale, float* y_scale, int64_t* bound, uint8_t* row_nomatch,
                       const int64_t* live, pats_stream_t stream) {
    PATS_REQUIRE(batch >= 0 && M > 1 && N > 1 && h > 0 && w > 0 && lim3 > 0 && iter_num >= 1,
                 "iterative_expand: bad argument");
    PATS_REQUIRE(h * w == N - 1, "iterative_expand: grid %dx%d does not match %d target columns", h,
                 w, N - 1);
    if (batch == 0) return PATS_OK;
    PATS_REQUIRE(P && scalex && scaley && whole_cost && core_cost && average_point && x_scale &&
                     y_scale && bound, "iterative_expand: null pointer");
    ExpandArgs a{P, input_is_log, batch * (int64_t)(M - 1), M, N, scalex, scaley, lim3, h, w,
                 lower_bound, iter_num, whole_cost, core_cost, average_point, x_scale, y_scale, bound, row_nomatch, live};
    int rows_per_wg = 16;
    size_t lds = 2 * (size_t)rows_per_wg * (size_t)(N + 3) * sizeof(float);
    if (lds > 48 * 1024) {
        rows_per_wg = 4;
        lds = 2 * (size_t)rows_per_wg * (size_t)(N + 3) * sizeof(float);
    }
    PATS_REQUIRE(lds <= 64 * 1024, "iterative_expand: N=%d too large", N);
    const int64_t blocks = ceil_div(a.rows_total, rows_per_wg);
    if (h <= 16 && w <= 16)
        hipLaunchKernelGGL(expand_kernel<true>, dim3((unsigned)blocks), dim3(16 * rows_per_wg), lds, as_stream(stream), a);
    else
        hipLaunchKernelGGL(expand_kernel<false>, dim3((unsigned)blocks), dim3(16 * rows_per_wg), lds, as_stream(stream), a);
    return check_launch("expand_kernel");
}
