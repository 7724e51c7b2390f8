/*
 * pats_oracle.c - CPU restatement of the PATS patch-area optimal-transport hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may call this; the product path (pats_amd/, libpats_amd.so) never does.
 *
 * Every function restates one piece of the reference algorithm in scalar fp32 C and cites the
 * reference file:line it follows (paths relative to /root/reference).  It is pinned against
 * tests/golden/ (.npz fixtures produced by running the reference's own Python / C++ in the build
 * container, tools/make_golden.py) by tests/test_oracle_golden.py.
 *
 * Numerics: values are fp32 like the reference (torch CPU fp32).  Reductions whose order ATen
 * leaves unspecified (logsumexp's sum, einsum/bmm dot products, .sum()) are accumulated in double
 * and rounded once to fp32 - the closest representable answer to what any fp32 summation order
 * approximates.  Index results (argmax, bounds, chunk plans) are exact integer arithmetic with
 * ATen-CPU's first-index tie-break.
 *
 * OpenMP is used only across independent problems (batch dimension) so the same code serves as
 * the bench's CPU baseline ("port") on the GPU box's host cores.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ZERO_F 1e-14f /* `zero = scores.new_tensor(1e-14)`, utils/utils.py:1201 */

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * a1-a3  cost build.  scores = einsum('bdn,bdm->bnm', d0, d1) / D**.5 ; then 0.1 * scores
 * models/first_layer.py:110-111,114  second_layer.py:100-101,104  third_layer.py:156-158
 * d0 [b,D,n], d1 [b,D,m] channel-major -> out [b,n,m]
 * ---------------------------------------------------------------------------------------- */
void oracle_cost(const float* d0, const float* d1, int64_t b, int D, int n, int m, float* out) {
    const float sq = (float)sqrt((double)D); /* `D ** .5` is a Python double, cast to fp32 by ATen */
#pragma omp parallel for schedule(static)
    for (int64_t bi = 0; bi < b; ++bi) {
        const float* a = d0 + bi * (int64_t)D * n;
        const float* c = d1 + bi * (int64_t)D * m;
        float* o = out + bi * (int64_t)n * m;
        double* acc = (double*)malloc(sizeof(double) * (size_t)m);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < m; ++j) acc[j] = 0.0;
            for (int d = 0; d < D; ++d) {
                const double av = a[(int64_t)d * n + i];
                const float* crow = c + (int64_t)d * m;
                for (int j = 0; j < m; ++j) acc[j] += av * (double)crow[j];
            }
            for (int j = 0; j < m; ++j) {
                float s = (float)acc[j];
                s = s / sq;
                o[(int64_t)i * m + j] = 0.1f * s;
            }
        }
        free(acc);
    }
}

/* torch.logsumexp over a strided vector of x[k] + add[k]  (ATen: amax, masked_fill(inf->0),
 * log(sum(exp(x - max))) + max) */
static inline float lse_strided(const float* x, int64_t xs, const float* add, int len) {
    float mx = -INFINITY;
    for (int k = 0; k < len; ++k) {
        float t = x[k * xs] + add[k];
        if (t > mx) mx = t;
    }
    float msub = isinf(mx) ? 0.0f : mx;
    double s = 0.0;
    for (int k = 0; k < len; ++k) {
        float t = x[k * xs] + add[k];
        s += (double)expf(t - msub);
    }
    return logf((float)s) + msub;
}

/* ------------------------------------------------------------------------------------------
 * a6  log_sinkhorn_iterations, models/modules.py:137-143
 *   u = v = 0;  iters x { u = log_mu - lse_j(Z + v);  v = log_nu - lse_i(Z + u) };  Z + u + v
 * Z [b,M,N], log_mu [b,M], log_nu [b,N] -> out [b,M,N].  If sub_norm != NULL, out -= norm[b]
 * (the `Z - norm` of modules.py:161,181).
 * ---------------------------------------------------------------------------------------- */
static void sinkhorn_one(const float* Z, int M, int N, const float* log_mu, const float* log_nu,
                         int iters, float* out, const float* sub_norm) {
    float* u = (float*)calloc((size_t)M, sizeof(float));
    float* v = (float*)calloc((size_t)N, sizeof(float));
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < M; ++i) u[i] = log_mu[i] - lse_strided(Z + (int64_t)i * N, 1, v, N);
        for (int j = 0; j < N; ++j) v[j] = log_nu[j] - lse_strided(Z + j, N, u, M);
    }
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            float z = (Z[(int64_t)i * N + j] + u[i]) + v[j];
            if (sub_norm) z = z - *sub_norm;
            out[(int64_t)i * N + j] = z;
        }
    free(u);
    free(v);
}

void oracle_sinkhorn(const float* Z, int64_t b, int M, int N, const float* log_mu,
                     const float* log_nu, int iters, float* out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t bi = 0; bi < b; ++bi)
        sinkhorn_one(Z + bi * (int64_t)M * N, M, N, log_mu + bi * M, log_nu + bi * N, iters,
                     out + bi * (int64_t)M * N, NULL);
}

/* ------------------------------------------------------------------------------------------
 * a4  log_optimal_transport, models/modules.py:145-162
 * scores [b,m,n], alpha scalar, ns [b,1,n] -> Z [b,m+1,n+1]
 * ---------------------------------------------------------------------------------------- */
void oracle_log_optimal_transport(const float* scores, int64_t b, int m, int n, float alpha,
                                  const float* ns, int iters, float* out) {
    const int M = m + 1, N = n + 1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t bi = 0; bi < b; ++bi) {
        const float* S = scores + bi * (int64_t)m * n;
        const float* nsb = ns + bi * (int64_t)n;
        float* C = (float*)malloc(sizeof(float) * (size_t)M * N);
        float* log_mu = (float*)malloc(sizeof(float) * (size_t)M);
        float* log_nu = (float*)malloc(sizeof(float) * (size_t)N);
        /* couplings = [[scores, alpha],[alpha, alpha]]  (modules.py:152-156) */
        for (int i = 0; i < m; ++i) {
            memcpy(C + (int64_t)i * N, S + (int64_t)i * n, sizeof(float) * (size_t)n);
            C[(int64_t)i * N + n] = alpha;
        }
        for (int j = 0; j < N; ++j) C[(int64_t)m * N + j] = alpha;
        /* norm = -log(ms + sum(ns))  (modules.py:157) */
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += (double)nsb[j];
        const float ns_sum = (float)acc;
        const float ms = (float)m;
        const float norm = -logf(ms + ns_sum);
        for (int j = 0; j < n; ++j) log_nu[j] = logf(nsb[j]) + norm; /* modules.py:158 */
        log_nu[n] = logf(ms) + norm;
        for (int i = 0; i < m; ++i) log_mu[i] = norm;                /* modules.py:159 */
        log_mu[m] = logf(ns_sum) + norm;
        sinkhorn_one(C, M, N, log_mu, log_nu, iters, out + bi * (int64_t)M * N, &norm);
        free(C);
        free(log_mu);
        free(log_nu);
    }
}

/* ------------------------------------------------------------------------------------------
 * a5  log_optimal_transport2, models/modules.py:165-182
 * scores [b,m,n] (last row/col = dustbin already), one = 1.0, ns [b,1,n-1] -> Z [b,m,n]
 * ---------------------------------------------------------------------------------------- */
void oracle_log_optimal_transport2(const float* scores, int64_t b, int m, int n, float one,
                                   const float* ns, int iters, float* out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t bi = 0; bi < b; ++bi) {
        const float* S = scores + bi * (int64_t)m * n;
        const float* nsb = ns + bi * (int64_t)(n - 1);
        float* log_mu = (float*)malloc(sizeof(float) * (size_t)m);
        float* log_nu = (float*)malloc(sizeof(float) * (size_t)n);
        const float ms = (float)(m - 1) * one;                       /* modules.py:169 */
        double acc = 0.0;
        for (int j = 0; j < n - 1; ++j) acc += (double)nsb[j];
        const float ns_sum = (float)acc;
        const float norm = -logf(ms + ns_sum);                       /* modules.py:176 */
        for (int j = 0; j < n - 1; ++j) log_nu[j] = logf(nsb[j]) + norm;
        log_nu[n - 1] = logf(ms) + norm;                             /* modules.py:178 */
        for (int i = 0; i < m - 1; ++i) log_mu[i] = norm;            /* modules.py:179 */
        log_mu[m - 1] = logf(ns_sum) + norm;
        sinkhorn_one(S, m, n, log_mu, log_nu, iters, out + bi * (int64_t)m * n, &norm);
        free(log_mu);
        free(log_nu);
    }
}

/* ------------------------------------------------------------------------------------------
 * a7  post-OT reductions.
 *   colmass: scales = sqrt(exp(Z[:, :-1, :-1]).sum(1) + 1e-8)   first_layer.py:117-118
 *   bias:    Z[:, :, -1] += log(k); Z[:, -1, :] += log(k)       second_layer.py:107-112
 *            (the corner gets it twice, as in the reference)
 * ---------------------------------------------------------------------------------------- */
void oracle_colmass_sqrt(const float* Z, int64_t b, int M, int N, float* out /*[b,N-1]*/) {
#pragma omp parallel for schedule(static)
    for (int64_t bi = 0; bi < b; ++bi) {
        const float* z = Z + bi * (int64_t)M * N;
        for (int j = 0; j < N - 1; ++j) {
            double s = 0.0;
            for (int i = 0; i < M - 1; ++i) s += (double)expf(z[(int64_t)i * N + j]);
            out[bi * (N - 1) + j] = sqrtf((float)s + 1e-8f);
        }
    }
}

void oracle_dustbin_bias(float* Z, int64_t b, int M, int N, float k) {
    const float lb = logf(1.0f * k); /* torch.log(self.one * 2) */
    for (int64_t bi = 0; bi < b; ++bi) {
        float* z = Z + bi * (int64_t)M * N;
        for (int i = 0; i < M; ++i) z[(int64_t)i * N + (N - 1)] += lb;
        for (int j = 0; j < N; ++j) z[(int64_t)(M - 1) * N + j] += lb;
    }
}

/* ------------------------------------------------------------------------------------------
 * a8  argmax with first-index tie-break: scores.max(2).indices / scores.max(1).indices
 * first_layer.py:162  second_layer.py:243.  Full [b,M] / [b,N] index vectors (the callers slice
 * [:, :-1]).
 * ---------------------------------------------------------------------------------------- */
void oracle_argmax(const float* Z, int64_t b, int M, int N, int64_t* row_arg /*[b,M]*/,
                   int64_t* col_arg /*[b,N]*/) {
#pragma omp parallel for schedule(static)
    for (int64_t bi = 0; bi < b; ++bi) {
        const float* z = Z + bi * (int64_t)M * N;
        if (row_arg)
            for (int i = 0; i < M; ++i) {
                int best = 0;
                float bv = z[(int64_t)i * N];
                for (int j = 1; j < N; ++j)
                    if (z[(int64_t)i * N + j] > bv) { bv = z[(int64_t)i * N + j]; best = j; }
                row_arg[bi * M + i] = best;
            }
        if (col_arg)
            for (int j = 0; j < N; ++j) {
                int best = 0;
                float bv = z[j];
                for (int i = 1; i < M; ++i)
                    if (z[(int64_t)i * N + j] > bv) { bv = z[(int64_t)i * N + j]; best = i; }
                col_arg[bi * N + j] = best;
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * a9  Compute_positions_and_ranges, utils/utils.py:1527-1537
 * positions[k] = (k / w, k % w) for the TRUE grid (h, w);  ranges is implicit:
 * ranges[d][k] = k <= d ? k : 1e7, row length max(h, w).
 * ---------------------------------------------------------------------------------------- */
void oracle_positions(int h, int w, float* positions /*[h*w,2]*/) {
    for (int k = 0; k < h * w; ++k) {
        positions[2 * k + 0] = (float)(k / w);
        positions[2 * k + 1] = (float)(k % w);
    }
}

static inline float range_val(int d, int k) { return k <= d ? (float)k : 1e7f; }

/* the `ranges` tensor itself, [n,n] with n = max(h, w): row i = [0 .. i] padded with 1e7 (utils.py:1532-1536) */
void oracle_ranges(int n, float* ranges /*[n,n]*/) {
    for (int d = 0; d < n; ++d)
        for (int k = 0; k < n; ++k) ranges[d * n + k] = range_val(d, k);
}

/* ------------------------------------------------------------------------------------------
 * a10 + a11  Iterative_expand_matrix (utils/utils.py:1179-1297) + Compute_scaling (:1321-1340)
 *
 * P       [b,M,N]   = exp(Z) incl. dustbin row (M-1) and dustbin column (N-1)
 * scalex, scaley [b,n] with n = N-1 (the reference passes [b,n,1])
 * lim3    = limitation[3] (true grid width, W // patch_scale), used for point0 only (:1189-1190)
 * h, w    = true grid; the function itself derives width = ranges.shape[0] = max(h,w) and
 *           height = positions.shape[0] // width (:1181) - the portrait swap quirk is kept.
 * outputs: whole_cost, core_cost [b,m]; average_point [b,m,2]; x_scale, y_scale [b,m];
 *          bound [b,m,4] int64 (up, down, left, right)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float* prow; /* this source patch's row of exp(Z), N entries */
    const float* scale; /* scalex*scaley, n entries */
    const float* opp;   /* dustbin row of exp(Z), n entries */
    int N, n, S;        /* S = sentinel index width*height+1 */
} expand_ctx;

/* expand_scores = cat([scores, 1e-14]) : N+1 entries (utils.py:1205) */
static inline float ES(const expand_ctx* c, int64_t idx) { return idx < c->N ? c->prow[idx] : ZERO_F; }
/* expand_scale = cat([scale, 1e-14, 1e-14]) : n+2 entries (utils.py:1206-1207) */
static inline float ESC(const expand_ctx* c, int64_t idx) { return idx < c->n ? c->scale[idx] : ZERO_F; }
/* opposite_nomatching_scores padded the same way (utils.py:1208) */
static inline float EOPP(const expand_ctx* c, int64_t idx) { return idx < c->n ? c->opp[idx] : ZERO_F; }

static inline int64_t clamp_seq(float f, int wh, int S) {
    int64_t s = (int64_t)f; /* float -> long assignment truncates (utils.py:1216-1219) */
    if (!(s >= 0)) s = S;   /* utils.py:1220 */
    if (!(s <= wh - 1)) s = S; /* utils.py:1221 */
    return s;
}

/* margin (optional, [b,m,2]): how close each row came to deciding differently - the classifier the parity reports
 * use to tell a summation-order tie from a real mismatch.  [0] = the smallest RELATIVE distance, over the initial
 * argmax and every growth step, between the quantities a decision compares (best strip sum against lower_bound,
 * best against second-best strip sum, best against second-best row entry): a rectangle can only differ from this
 * one's if an implementation's sums differ from these by at least that much.  [1] = the same for the per-element
 * test `expand_sum > lower_bound` of :1225, which feeds whole_cost only. */
static void iterative_expand_impl(const float* P, int64_t b, int M, int N, const float* scalex,
                             const float* scaley, int lim3, int h, int w, float lower_bound,
                             int iter_num, float* whole_cost, float* core_cost,
                             float* average_point, float* x_scale, float* y_scale,
                             int64_t* bound_out, float* margin) {
    const int m = M - 1, n = N - 1;
    const int width = h > w ? h : w;            /* ranges.shape[0]          (utils.py:1181) */
    const int height = (h * w) / width;         /* positions.shape[0] // width              */
    const int wh = width * height;
    const int S = wh + 1;
    float* positions = (float*)malloc(sizeof(float) * 2 * (size_t)(h * w));
    oracle_positions(h, w, positions);
#pragma omp parallel for schedule(static)
    for (int64_t bi = 0; bi < b; ++bi) {
        const float* Pb = P + bi * (int64_t)M * N;
        const float* sx = scalex + bi * (int64_t)n;
        const float* sy = scaley + bi * (int64_t)n;
        float* scale = (float*)malloc(sizeof(float) * (size_t)n);
        float* ox = (float*)malloc(sizeof(float) * (size_t)n);
        float* oy = (float*)malloc(sizeof(float) * (size_t)n);
        for (int j = 0; j < n; ++j) scale[j] = sx[j] * sy[j]; /* utils.py:1192 */
        const float* opp = Pb + (int64_t)(M - 1) * N;       /* scores_in[:, -1, :-1] :1183 */
        for (int r = 0; r < m; ++r) {
            expand_ctx c = {Pb + (int64_t)r * N, scale, opp, N, n, S};
            /* max0 over real columns (:1182); argmax over all columns (:1191) */
            int max0 = 0, maxall = 0;
            for (int j = 1; j < n; ++j) if (c.prow[j] > c.prow[max0]) max0 = j;
            for (int j = 1; j < N; ++j) if (c.prow[j] > c.prow[maxall]) maxall = j;
            const int if_nomatching = (maxall == m); /* `== scores.shape[1]` (:1191) */
            float mg_bound = INFINITY, mg_elem = INFINITY;
            for (int j = 0; j < n; ++j)
                if (j != max0) mg_bound = fminf(mg_bound, (c.prow[max0] - c.prow[j]) / fabsf(c.prow[max0]));
            float last_nomatching = opp[max0];       /* :1184 */
            int64_t up = max0 / lim3, down = up, left = max0 % lim3, right = left; /* :1189-1197 */
            int bd0 = 0, bd1 = 0, sb0 = 0, sb1 = 0;  /* bound_difference; sb* = the copy the LAST
                                                        iteration's sequence_base was built from */
            float last_sum = ES(&c, max0);           /* :1209 */
            float last_scale = ESC(&c, max0);        /* :1210 */
            for (int it = 0; it < iter_num; ++it) {
                sb0 = bd0; sb1 = bd1;                /* sequence_base = ranges[bound_difference] :1215 */
                double e_sum[4] = {0, 0, 0, 0}, nm_sum[4] = {0, 0, 0, 0}, sc_sum[4] = {0, 0, 0, 0};
                const float off[4] = {
                    (float)(left + up * width - width),   /* strip above   :1217 */
                    (float)(left + down * width + width), /* strip below   :1218 */
                    (float)(left + up * width - 1),       /* strip left    :1219 */
                    (float)(right + up * width + 1)};     /* strip right   :1220 */
                for (int d = 0; d < 4; ++d)
                    for (int k = 0; k < width; ++k) {
                        float f = d < 2 ? range_val(sb1, k) + off[d]
                                        : range_val(sb0, k) * (float)width + off[d];
                        int64_t s = clamp_seq(f, wh, S);
                        float v = ES(&c, s);
                        e_sum[d] += (double)v;
                        nm_sum[d] += (double)(v > lower_bound ? EOPP(&c, s) : ZERO_F); /* :1225 */
                        mg_elem = fminf(mg_elem, fabsf(v - lower_bound) / lower_bound);
                        sc_sum[d] += (double)ESC(&c, s);                               /* :1231 */
                    }
                float es[4];
                for (int d = 0; d < 4; ++d) es[d] = (float)e_sum[d];
                if (up == 0) es[0] = ZERO_F;              /* :1227-1230 */
                if (down == height - 1) es[1] = ZERO_F;
                if (left == 0) es[2] = ZERO_F;
                if (right == width - 1) es[3] = ZERO_F;
                int arg = 0;
                for (int d = 1; d < 4; ++d) if (es[d] > es[arg]) arg = d; /* :1232 */
                const float max_sum = es[arg];
                float add_sum = ZERO_F, add_scale = ZERO_F, add_nm = ZERO_F;
                mg_bound = fminf(mg_bound, fabsf(max_sum - lower_bound) / lower_bound);
                if (max_sum > lower_bound)
                    for (int d = 0; d < 4; ++d)
                        if (d != arg) mg_bound = fminf(mg_bound, (max_sum - es[d]) / max_sum);
                if (max_sum > lower_bound) {              /* :1235-1238 */
                    if (arg == 0) up -= 1; else if (arg == 1) down += 1;
                    else if (arg == 2) left -= 1; else right += 1;
                    add_sum = max_sum;
                    add_scale = (float)sc_sum[arg];
                    add_nm = (float)nm_sum[arg];
                }
                bd0 = (int)(down - up);                   /* :1239-1240 */
                bd1 = (int)(right - left);
                last_sum = last_sum + add_sum;            /* :1241-1243 */
                last_scale = last_scale + add_scale;
                last_nomatching = last_nomatching + add_nm;
            }
            (void)last_scale;
            const int if_core_exist = (bd0 > 1) && (bd1 > 1); /* :1244 */
            /* border strips of the final rectangle, with the STALE sequence_base (:1245-1253) */
            double edge_sum[4] = {0, 0, 0, 0}, edge_scale[4] = {0, 0, 0, 0};
            const float eoff[4] = {(float)(left + up * width), (float)(left + down * width),
                                   (float)(left + up * width), (float)(right + up * width)};
            for (int d = 0; d < 4; ++d)
                for (int k = 0; k < width; ++k) {
                    float f = d < 2 ? range_val(sb1, k) + eoff[d]
                                    : range_val(sb0, k) * (float)width + eoff[d];
                    int64_t s = clamp_seq(f, wh, S);
                    edge_sum[d] += (double)ES(&c, s);
                    edge_scale[d] += (double)ESC(&c, s);
                }
            /* in-rectangle weights (:1254-1260) and weighted centroid (:1264-1273) */
            double wx = 0, wy = 0, sumx = 0, sumy = 0, ws = 0, so = 0;
            for (int p = 0; p < n; ++p) {
                const float py = positions[2 * p], px = positions[2 * p + 1];
                const int crit = (py >= (float)up) && (py <= (float)down) && (px >= (float)left) &&
                                 (px <= (float)right);
                const float root = sqrtf(c.prow[p] + 1e-7f);
                ox[p] = crit ? root / sx[p] : ZERO_F;
                oy[p] = crit ? root / sy[p] : ZERO_F;
                wx += (double)ox[p] * (double)px;
                wy += (double)oy[p] * (double)py;
                sumx += (double)ox[p];
                sumy += (double)oy[p];
                const float o = ox[p] * oy[p];            /* Compute_scaling :1323 */
                ws += (double)o * (double)scale[p];
                so += (double)o;
            }
            const int64_t o2 = (bi * m + r);
            average_point[o2 * 2 + 1] = (float)wx / (float)sumx + 0.5f;
            average_point[o2 * 2 + 0] = (float)wy / (float)sumy + 0.5f;
            /* corners (:1279-1287) */
            const int64_t corner[4] = {up * width + left, up * width + right, down * width + left,
                                       down * width + right};
            double cps = 0, css = 0;
            for (int q = 0; q < 4; ++q) {
                int64_t s = corner[q];
                if (!(s >= 0)) s = S;
                if (!(s <= wh - 1)) s = S;
                cps += (double)ES(&c, s);
                css += (double)ESC(&c, s);
            }
            double ts = 0;
            for (int j = 0; j < N; ++j) ts += (double)c.prow[j];
            const float the_scale = (float)ts;            /* :1288 */
            const float e4 = (float)((double)(float)edge_sum[0] + (double)(float)edge_sum[1] +
                                     (double)(float)edge_sum[2] + (double)(float)edge_sum[3]);
            const float s4 = (float)((double)(float)edge_scale[0] + (double)(float)edge_scale[1] +
                                     (double)(float)edge_scale[2] + (double)(float)edge_scale[3]);
            const float core_scale_sum = the_scale - s4 + (float)css;  /* :1289 */
            const float core_sum = last_sum - e4 + (float)cps;         /* :1290 */
            core_cost[o2] = (if_core_exist && !if_nomatching)
                                ? fabsf((core_sum - core_scale_sum) / the_scale) : ZERO_F; /* :1291 */
            whole_cost[o2] = if_nomatching ? ZERO_F
                : (fabsf(the_scale - last_sum) + last_nomatching / 4.0f) / the_scale;      /* :1296 */
            const float average_scale = sqrtf((float)ws / (float)so);  /* :1326 */
            x_scale[o2] = 1.0f / (average_scale / 1.0f);               /* :1335-1340, ratio = 1.0 */
            y_scale[o2] = 1.0f / (average_scale * 1.0f);
            bound_out[o2 * 4 + 0] = up;
            bound_out[o2 * 4 + 1] = down;
            bound_out[o2 * 4 + 2] = left;
            bound_out[o2 * 4 + 3] = right;
            if (margin) {
                margin[o2 * 2 + 0] = mg_bound;
                margin[o2 * 2 + 1] = mg_elem;
            }
        }
        free(scale);
        free(ox);
        free(oy);
    }
    free(positions);
}

void oracle_iterative_expand(const float* P, int64_t b, int M, int N, const float* scalex,
                             const float* scaley, int lim3, int h, int w, float lower_bound,
                             int iter_num, float* whole_cost, float* core_cost,
                             float* average_point, float* x_scale, float* y_scale,
                             int64_t* bound_out) {
    iterative_expand_impl(P, b, M, N, scalex, scaley, lim3, h, w, lower_bound, iter_num, whole_cost, core_cost,
                          average_point, x_scale, y_scale, bound_out, NULL);
}

void oracle_iterative_expand_margin(const float* P, int64_t b, int M, int N, const float* scalex,
                                    const float* scaley, int lim3, int h, int w, float lower_bound,
                                    int iter_num, float* whole_cost, float* core_cost,
                                    float* average_point, float* x_scale, float* y_scale,
                                    int64_t* bound_out, float* margin) {
    iterative_expand_impl(P, b, M, N, scalex, scaley, lim3, h, w, lower_bound, iter_num, whole_cost, core_cost,
                          average_point, x_scale, y_scale, bound_out, margin);
}

/* ------------------------------------------------------------------------------------------
 * a12  split_patches, utils/utils.py:152-181.  sum_cycle [h*w] int32 cumsum of matched flags.
 * Python's negative index `sum_cycle[i*width - 1]` at i == 0 reads the LAST element; kept.
 * second/third: caller-provided [h+1][2] int64.  Returns cycle_num.
 * ---------------------------------------------------------------------------------------- */
int oracle_split_patches(const int32_t* sum_cycle, int h, int w, int max_once_used,
                         int64_t* second, int64_t* third) {
    const int L = h * w;
#define SC(i) ((int64_t)sum_cycle[((i) % L + L) % L])
    int cycle = 0, last_second = 0, last_third = 0;
    for (int i = 0; i < h; ++i) {
        const int64_t num = SC((i + 1) * w - 1);
        if (num > (int64_t)max_once_used * (cycle + 1)) {
            const int64_t origin = last_second == 0 ? 0 : SC(last_second * w - 1);
            second[2 * cycle] = origin;
            second[2 * cycle + 1] = SC((i + 1) * w - 1);
            third[2 * cycle] = SC(last_third * w) - origin;
            third[2 * cycle + 1] = SC((i + 1) * w - 1) - SC(i * w - 1);
            cycle += 1;
            last_second = i;
            last_third = i + 1;
        }
    }
    const int64_t origin = last_second == 0 ? 0 : SC(last_second * w - 1);
    second[2 * cycle] = origin;
    second[2 * cycle + 1] = (int64_t)h * w;
    const int64_t end_num = (last_third == h) ? origin : SC(last_third * w);
    third[2 * cycle] = end_num - origin;
    third[2 * cycle + 1] = 0;
    cycle += 1;
#undef SC
    return cycle;
}

/* ------------------------------------------------------------------------------------------
 * a13  Compute_imgs bounds, utils/utils.py:1350-1382 (margin = 128, patch_scale = 32).
 * In : x_scale, y_scale [N]; average_point [N,2]; if_nomatching [N] (uint8); grid (height,width)
 * Out: bound5 [K,5] int64 (y0,y1,x0,x1,seq) for matched patches in order;
 *      x_scale_new, y_scale_new [N,2]; average_new [N,2].   Returns K.   (batch element img)
 * ---------------------------------------------------------------------------------------- */
int64_t oracle_compute_imgs_bounds(const float* x_scale, const float* y_scale,
                                   const float* average_point, const uint8_t* if_nomatching,
                                   int Np, int height, int width, int img, int64_t* bound5,
                                   float* x_scale_new, float* y_scale_new, float* average_new) {
    const float ps = 32.0f, margin = 128.0f;
    const float board1 = (float)(32 * height - 1), board3 = (float)(32 * width); /* :1351 */
    int64_t K = 0;
    for (int k = 0; k < Np; ++k) {
        const float ay = average_point[2 * k], ax = average_point[2 * k + 1];
        float b0 = (ay - y_scale[k] * 3.0f / 2.0f) * ps + margin; /* :1360-1363 */
        float b1 = (ay + y_scale[k] * 3.0f / 2.0f) * ps + margin;
        float b2 = (ax - x_scale[k] * 3.0f / 2.0f) * ps + margin;
        float b3 = (ax + x_scale[k] * 3.0f / 2.0f) * ps + margin;
        b0 = b0 >= 0 ? b0 : 0.0f;                                  /* :1364 */
        b1 = b1 >= 0 ? b1 : 0.0f;
        b2 = b2 >= 0 ? b2 : 0.0f;
        b3 = b3 >= 0 ? b3 : 0.0f;
        b1 = (b1 < (float)(32 * height + 256)) ? b1 : board1;       /* :1365 */
        b3 = (b3 < (float)(32 * width + 256)) ? b3 : board3;        /* :1366 */
        x_scale_new[2 * k] = (b1 - b0 + 1.0f) / 96.0f;              /* :1367 (row extent!) */
        x_scale_new[2 * k + 1] = 1.0f;                              /* :1378-1381 */
        y_scale_new[2 * k] = (b3 - b2 + 1.0f) / 96.0f;              /* :1368 */
        y_scale_new[2 * k + 1] = 1.0f;
        const int64_t l0 = (int64_t)b0, l1 = (int64_t)b1, l2 = (int64_t)b2, l3 = (int64_t)b3; /* :1369 */
        average_new[2 * k + 1] = (float)(l1 + l0) / 2.0f - 128.0f + 0.5f; /* :1371 */
        average_new[2 * k + 0] = (float)(l2 + l3) / 2.0f - 128.0f + 0.5f; /* :1372 */
        if (!if_nomatching[k]) {
            bound5[K * 5 + 0] = l0;
            bound5[K * 5 + 1] = l1;
            bound5[K * 5 + 2] = l2;
            bound5[K * 5 + 3] = l3;
            bound5[K * 5 + 4] = (int64_t)img * 10000 + k;           /* :1374-1377 */
            K += 1;
        }
    }
    return K;
}

/* ------------------------------------------------------------------------------------------
 * a13  left crops: origin_extract on the 32-px zero-padded left image, utils/utils.py:1300-1318,
 * caller :1383-1384.  left [H,W,3] HWC -> out [K,96,96,3] for matched patches (grid h x w).
 * ---------------------------------------------------------------------------------------- */
int64_t oracle_left_crops(const float* left, int H, int W, const uint8_t* if_nomatching, int h,
                          int w, float* out) {
    int64_t K = 0;
    for (int k = 0; k < h * w; ++k) {
        if (if_nomatching[k]) continue;
        const int r = k / w, c = k % w;
        float* o = out + K * 96 * 96 * 3;
        for (int y = 0; y < 96; ++y)
            for (int x = 0; x < 96; ++x) {
                const int iy = r * 32 + y - 32, ix = c * 32 + x - 32;
                for (int ch = 0; ch < 3; ++ch) {
                    float v = 0.0f;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = left[((int64_t)iy * W + ix) * 3 + ch];
                    o[((int64_t)y * 96 + x) * 3 + ch] = v;
                }
            }
        K += 1;
    }
    return K;
}

/* ------------------------------------------------------------------------------------------
 * a14  tensor_resize, setup/library.cpp:47-66.
 * input [n_img,C,Hp,Wp]; bound [K,5] int64 (y0,y1,x0,x1,seq); out [K,C,96,96].
 * crop rows [y0,y1), cols [x0,x1]  (height y1-y0, width x1-x0+1, library.cpp:56-59);
 * upsample_bilinear2d(align_corners=true) (library.cpp:60) = ATen's
 * area_pixel_compute_scale / source index: scale = (in-1)/(out-1), src = scale*dst.
 * Returns 0, or -1 if a crop is empty / out of range (torch raises there).
 * ---------------------------------------------------------------------------------------- */
int oracle_tensor_resize(const float* input, int n_img, int C, int Hp, int Wp, const int64_t* bound,
                         int64_t K, float* out) {
    int err = 0;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < K; ++i) {
        const int64_t y0 = bound[i * 5], y1 = bound[i * 5 + 1], x0 = bound[i * 5 + 2],
                      x1 = bound[i * 5 + 3], img = bound[i * 5 + 4] / 10000; /* library.cpp:55-56 */
        const int64_t ih = y1 - y0, iw = x1 - x0 + 1;
        if (ih <= 0 || iw <= 0 || y0 < 0 || x0 < 0 || y0 + ih > Hp || x0 + iw > Wp || img < 0 ||
            img >= n_img) {
#pragma omp atomic write
            err = -1;
            continue;
        }
        const float sh = (float)(ih - 1) / 95.0f, sw = (float)(iw - 1) / 95.0f;
        for (int c = 0; c < C; ++c) {
            const float* src = input + (((int64_t)img * C + c) * Hp + y0) * Wp + x0;
            float* o = out + ((i * C + c) * 96) * 96;
            for (int oy = 0; oy < 96; ++oy) {
                const float fy = sh * (float)oy;
                const int64_t h1 = (int64_t)fy;
                const int64_t h1p = (h1 < ih - 1) ? 1 : 0;
                const float l1 = fy - (float)h1, l0 = 1.0f - l1;
                for (int ox = 0; ox < 96; ++ox) {
                    const float fx = sw * (float)ox;
                    const int64_t w1 = (int64_t)fx;
                    const int64_t w1p = (w1 < iw - 1) ? 1 : 0;
                    const float m1 = fx - (float)w1, m0 = 1.0f - m1;
                    const float* p = src + h1 * Wp + w1;
                    o[oy * 96 + ox] = l0 * (m0 * p[0] + m1 * p[w1p]) +
                                      l1 * (m0 * p[h1p * Wp] + m1 * p[h1p * Wp + w1p]);
                }
            }
        }
    }
    return err;
}

/* ------------------------------------------------------------------------------------------
 * a17 + a18  ThirdLayer.Compute_result (third_layer.py:184-217) and the match label
 * (third_layer.py:161-170).  W = 8, T = 5.
 * scores [P,65,65] = exp(Z);  scale_x, scale_y [P,64];  p_s, p_t [P,2] int64
 * -> mkpts0_f, mkpts1_f [P,16,2];  whole_loss [P,16];  label [P*16,2];  if_matching1 [P,16]
 * ---------------------------------------------------------------------------------------- */
void oracle_compute_result(const float* scores, int64_t P, const float* scale_x,
                           const float* scale_y, const int64_t* p_s, const int64_t* p_t,
                           int outdoor, float* mkpts0_f, float* mkpts1_f, float* whole_loss,
                           float* label, uint8_t* if_matching1) {
    const int W = 8, T = 5, NN = 65;
    int64_t count = 0;
#pragma omp parallel for schedule(static) reduction(+ : count)
    for (int64_t p = 0; p < P; ++p) {
        const float* Sp = scores + p * (int64_t)NN * NN;
        const float* sx = scale_x + p * 64;
        const float* sy = scale_y + p * 64;
        for (int q = 0; q < 16; ++q) {
            const int qy = q / 4 + 2, qx = q % 4 + 2; /* [:, 2:6, 2:6] (:186,188) */
            const float* row = Sp + (int64_t)(qy * W + qx) * NN;
            int max0 = 0;
            for (int j = 1; j < 64; ++j) if (row[j] > row[max0]) max0 = j; /* :188 */
            const int mx = max0 % W, my = max0 / W;
            double wpx = 0, wpy = 0, sumx = 0, sumy = 0, unfold = 0;
            for (int t = 0; t < T * T; ++t) {
                const int tx = t % T, ty = t / T;
                /* index3 = y3*(W+4) + x3 into the pad-2 map (:189-191) */
                const int ux = mx + tx - 2, uy = my + ty - 2; /* unpadded coordinates */
                const int inside = (ux >= 0 && ux < W && uy >= 0 && uy < W);
                const float sb = inside ? row[uy * W + ux] : 0.0f;        /* ZeroPad2d(2)  :185 */
                const float scx = inside ? sx[uy * W + ux] : 1e-2f;       /* pad_1 = 1e-2  :195 */
                const float scy = inside ? sy[uy * W + ux] : 1e-2f;
                const float root = sqrtf(sb + 1e-7f);
                const float fx = root / scx, fy = root / scy;              /* :197-198 */
                const float posx = (float)tx * 2.0f - (float)(T - 1);      /* :199 */
                const float posy = (float)ty * 2.0f - (float)(T - 1);
                wpx += (double)fx * (double)posx;
                wpy += (double)fy * (double)posy;
                sumx += (double)fx;
                sumy += (double)fy;
                unfold += (double)sb;
            }
            const int64_t o = (p * 16 + q) * 2;
            float m1x = (float)wpx / (float)sumx + ((float)mx + 0.5f - (float)W / 2) * 2.0f; /* :206 */
            float m1y = (float)wpy / (float)sumy + ((float)my + 0.5f - (float)W / 2) * 2.0f; /* :207 */
            mkpts1_f[o + 0] = m1x + (float)p_t[p * 2 + 0];                                   /* :208 */
            mkpts1_f[o + 1] = m1y + (float)p_t[p * 2 + 1];
            mkpts0_f[o + 0] = (float)p_s[p * 2 + 0] + (float)(q % 4) * 2.0f - 3.0f;           /* :209-210 */
            mkpts0_f[o + 1] = (float)p_s[p * 2 + 1] + (float)(q / 4) * 2.0f - 3.0f;
            double rs = 0;
            for (int j = 0; j < NN; ++j) rs += (double)row[j];
            const float wl = (float)rs - (float)unfold;                                       /* :213 */
            whole_loss[p * 16 + q] = wl;
            if (wl >= 1e-2f) count += 1;
            /* label (:161-170) */
            int amax = 0;
            float bv = row[0] + 1e-8f;
            for (int j = 1; j < NN; ++j) {
                const float v = row[j] + 1e-8f;
                if (v > bv) { bv = v; amax = j; }
            }
            const int matching = (amax != W * W);
            if_matching1[p * 16 + q] = (uint8_t)matching;
            float l0 = 1e8f;
            if (!outdoor) {
                const int64_t a = (p * 16 + q) % 16;
                const int select = (a == 5 || a == 15 || a == 7 || a == 13);
                l0 = select ? l0 : -10.0f;
            } else {
                l0 = matching ? l0 : -10.0f;
            }
            label[o + 0] = l0;
            label[o + 1] = 1e8f;
        }
    }
    /* whole_loss = where(wl >= 1e-2, wl, 0) / (count + 10) / 10   (:215) */
    const float denom = (float)count + 10.0f;
    for (int64_t k = 0; k < P * 16; ++k) {
        const float wl = whole_loss[k];
        whole_loss[k] = (wl >= 1e-2f ? wl : 0.0f) / denom / 10.0f;
    }
}

/* ------------------------------------------------------------------------------------------
 * a15  fine-level descriptor sampling, models/second_layer.py:71-86
 * f0 [2B,64,48,48], f1 [2B,64,24,24], f2 [2B,128,12,12], title [B,8], rubbish [B,264]
 * -> desc [2,B,264,145]
 * ---------------------------------------------------------------------------------------- */
void oracle_fine_descriptors(const float* f0, const float* f1, const float* f2, const float* title,
                             const float* rubbish, int64_t B, float* desc) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < 2 * B; ++n) {
        const int64_t b = n % B;
        float* o = desc + n * 264 * 145;
        for (int ch = 0; ch < 264; ++ch)
            for (int p = 0; p < 145; ++p) {
                float v;
                if (p == 144) v = rubbish[b * 264 + ch];                   /* :83,85 */
                else if (ch < 8) v = title[b * 8 + ch];                    /* :82,84 */
                else {
                    const int r = p / 12, c = p % 12;                      /* positions (k//12, k%12) :45-49 */
                    if (ch < 72) {        /* AvgPool2d(2,1,1) of the 48x48 map at (4r+2, 4c+2)  :73-79 */
                        const float* m = f0 + (n * 64 + (ch - 8)) * 48 * 48;
                        const int y = 4 * r + 1, x = 4 * c + 1;
                        v = (((m[y * 48 + x] + m[y * 48 + x + 1]) + m[(y + 1) * 48 + x]) + m[(y + 1) * 48 + x + 1]) / 4.0f;
                    } else if (ch < 136) {   /* 24x24 map pooled, at (2r+1, 2c+1) */
                        const float* m = f1 + (n * 64 + (ch - 72)) * 24 * 24;
                        const int y = 2 * r, x = 2 * c;
                        v = (((m[y * 24 + x] + m[y * 24 + x + 1]) + m[(y + 1) * 24 + x]) + m[(y + 1) * 24 + x + 1]) / 4.0f;
                    } else {
                        v = f2[(n * 128 + (ch - 136)) * 144 + p];
                    }
                }
                o[ch * 145 + p] = v;
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * a16  third-level window gather, models/third_layer.py:121-146
 * ff0, ff1 [B,128,52,52]; mk0, mk1 [P,2] (x,y); b_ids [P]; kenc [128,64]; rubbish [B,128,144]
 * -> out0, out1 [P,128,65]; ps, pt [P,2] int64 (points rounded to the 4-px lattice)
 * Returns -1 if an index leaves the map (torch.gather raises there), else 0.
 * ---------------------------------------------------------------------------------------- */
static long long floordiv2(long long v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }

int oracle_third_descriptors(const float* ff0, const float* ff1, const float* mk0, const float* mk1,
                             const int64_t* b_ids, const float* kenc, const float* rubbish, int64_t P,
                             int64_t B, float* out0, float* out1, int64_t* ps, int64_t* pt) {
    const int W = 8, M = 52, C = 128;
    int err = 0;
    for (int64_t p = 0; p < P; ++p) {
        const int64_t b = b_ids[p];
        const long long s0 = (long long)rintf(mk0[p * 2] / 4.0f) * 4, s1 = (long long)rintf(mk0[p * 2 + 1] / 4.0f) * 4; /* :124 */
        float t0 = mk1[p * 2], t1 = mk1[p * 2 + 1];
        t0 = t0 >= 96.f ? 96.f : t0; t1 = t1 >= 96.f ? 96.f : t1;            /* :128 */
        t0 = t0 <= 0.f ? 0.f : t0;   t1 = t1 <= 0.f ? 0.f : t1;              /* :129 */
        const long long q0 = (long long)rintf(t0 / 4.0f) * 4, q1 = (long long)rintf(t1 / 4.0f) * 4;                   /* :130 */
        ps[p * 2] = s0; ps[p * 2 + 1] = s1; pt[p * 2] = q0; pt[p * 2 + 1] = q1;
        const long long x2 = (long long)rintf((float)s0 / 8.0f), y2 = (long long)rintf((float)s1 / 8.0f);              /* :141-142 */
        /* index2 = b*144 + y2*12 + x2 is a ROW of rubbish.permute(0,2,1).reshape(-1,128) (:143-144): cell 11 of a patch
         * (round(92/8) = 12) reads the next patch's feature; only leaving the whole tensor raises */
        const long long i2 = b * 144 + y2 * 12 + x2;
        if (i2 < 0 || i2 >= B * 144 || b < 0 || b >= B) { err = -1; continue; }
        for (int t = 0; t < 64; ++t) {
            const int wx = t % W, wy = t / W;
            const long long x0 = floordiv2(s0) + wx - W / 2 + 2, y0 = floordiv2(s1) + wy - W / 2 + 2;   /* :125-126 */
            const long long x1 = floordiv2(q0) + wx - W / 2 + 2, y1 = floordiv2(q1) + wy - W / 2 + 2;   /* :131-132 */
            const long long i0 = b * M * M + y0 * M + x0, i1 = b * M * M + y1 * M + x1;                   /* :127,133 */
            if (i0 < 0 || i0 >= B * M * M || i1 < 0 || i1 >= B * M * M) { err = -1; continue; }
            const long long bb0 = i0 / (M * M), r0 = i0 % (M * M), bb1 = i1 / (M * M), r1 = i1 % (M * M);
            for (int ch = 0; ch < C; ++ch) {
                out0[(p * C + ch) * 65 + t] = ff0[(bb0 * C + ch) * (M * M) + r0] + kenc[ch * 64 + t];    /* :139 */
                out1[(p * C + ch) * 65 + t] = ff1[(bb1 * C + ch) * (M * M) + r1] + kenc[ch * 64 + t];    /* :140 */
            }
        }
        for (int ch = 0; ch < C; ++ch) {
            const float rb = rubbish[((i2 / 144) * C + ch) * 144 + i2 % 144];                                              /* :143-146 */
            out0[(p * C + ch) * 65 + 64] = rb;
            out1[(p * C + ch) * 65 + 64] = rb;
        }
    }
    return err;
}

/* ==========================================================================================
 * SURVEY.md section 8(f) "next" rows: the steps either side of the OT path.
 * ======================================================================================== */

/* ------------------------------------------------------------------------------------------
 * f1  merge_patches_new / merge_patches_old, models/second_layer.py:137-238.
 * Every 8-px cell of the 1/8-resolution grid (4h x 4w cells) is covered by the 96x96 windows of up
 * to 9 coarse patches; the merge keeps one candidate per cell.  Restated step by step on full
 * arrays in the reference's own layouts:
 *   owner layout  A[bt][4*hh + r][4*ww + s][a*3 + c]  = window cell (a*4 + r, c*4 + s) of coarse
 *   patch (hh, ww)  (second_layer.py:160,162-164 / :211-214), geometrically at fine cell
 *   (4*(hh + a - 1) + r, 4*(ww + c - 1) + s).
 * trust [B,144] and ifn_L2 [B,144] are modified in place like the reference does (:143-147 /
 * :194-199); scores_back [batch,h*w,16,9] (fp64) receives this chunk's scores (:163 / :213).
 * `argsort(...)[..., 0]` (:175 / :232) is restated as the first index of the minimum (ATen's CPU
 * sort is stable); pinned by tests/golden/merge_*.npz which contain exact ties.
 * The final scatter (:190 / :238) is sequential in source order, last write wins (ATen CPU).
 * returns 0, -1 if the number of unmasked coarse patches != B, -2 if a scatter index leaves the
 * tensor (the reference raises there).
 * ---------------------------------------------------------------------------------------- */
int oracle_merge_patches(int merge_new, int64_t B, float* trust, int H, int W, int batch_num,
                         const uint8_t* ifn_L1, uint8_t* ifn_L2, double* scores_back, uint8_t* out) {
    const int h = H / 32, w = W / 32, h4 = 4 * h, w4 = 4 * w;
    const int64_t NP = (int64_t)batch_num * h * w, per = (int64_t)h4 * w4 * 9;
    for (int64_t b = 0; b < B; ++b)
        for (int cell = 0; cell < 144; ++cell) {
            const int x = cell % 12, y = cell / 12;
            float t = trust[b * 144 + cell];
            for (int i = 0; i < 3; ++i)                                   /* :143-147 / :194-198 */
                if (x < 3 - i || x > 7 + i || y < 3 - i || y > 7 + i) t *= 2.0f;
            uint8_t f = ifn_L2[b * 144 + cell];
            if (t > 2.0f) f = 1;                                          /* :148 / :199 */
            if (x < 1 || x > 10 || y < 1 || y > 10) f = 1;                /* :149 / :200 */
            if (merge_new && !f) t -= 10000.0f;                           /* :201 */
            trust[b * 144 + cell] = t;
            ifn_L2[b * 144 + cell] = f;
        }
    int64_t* slot = (int64_t*)malloc(sizeof(int64_t) * (size_t)NP);
    int64_t cnt = 0;
    for (int64_t q = 0; q < NP; ++q) slot[q] = ifn_L1[q] ? -1 : cnt++;
    if (cnt != B) { free(slot); return -1; }
    uint8_t* ifm = (uint8_t*)calloc((size_t)(batch_num * per), 1);
    double* use = (double*)malloc(sizeof(double) * (size_t)(batch_num * per));
    for (int64_t q = 0; q < NP; ++q) {
        const int bt = (int)(q / (h * w)), hh = (int)(q % (h * w)) / w, ww = (int)(q % (h * w)) % w;
        for (int r = 0; r < 4; ++r)
            for (int s = 0; s < 4; ++s)
                for (int a = 0; a < 3; ++a)
                    for (int c = 0; c < 3; ++c) {
                        const int64_t o = bt * per + ((int64_t)(hh * 4 + r) * w4 + ww * 4 + s) * 9 + a * 3 + c;
                        double* sb = scores_back + (q * 16 + r * 4 + s) * 9 + a * 3 + c;
                        if (slot[q] >= 0) {
                            const int64_t src = slot[q] * 144 + (a * 4 + r) * 12 + c * 4 + s;
                            ifm[o] = !ifn_L2[src];                        /* :157-160 / :206-209 */
                            *sb = (double)trust[src];                     /* :161-163 / :210-211 */
                        }
                        use[o] = *sb;                                     /* :164 / :212 */
                    }
    }
    uint8_t* res = (uint8_t*)malloc((size_t)(batch_num * per));
    memset(res, 1, (size_t)(batch_num * per));                           /* :189 / :237 ones */
    int rc = 0;
    if (merge_new) {
        uint8_t* ifm2 = (uint8_t*)malloc((size_t)(batch_num * per));
        memcpy(ifm2, ifm, (size_t)(batch_num * per));                     /* :226 */
        for (int bt = 0; bt < batch_num; ++bt)
            for (int Y = 0; Y < h4; ++Y)
                for (int X = 0; X < w4; ++X)
                    for (int k = 0; k < 9; ++k) {
                        const int a = k / 3, c = k % 3;
                        const int by = Y + 4 * (a - 1), bx = X + 4 * (c - 1);       /* :216-220 */
                        if (by < 0 || by >= h4 || bx < 0 || bx >= w4)
                            use[bt * per + ((int64_t)Y * w4 + X) * 9 + k] += 100000.0; /* :221-223 */
                    }
        for (int i = 0; i < 9; ++i) {                                     /* :227-231 */
            const int dy = -(i % 3 - 1), dx = -(i / 3 - 1);
            const int y0 = 4 * (dx > 0 ? dx : 0), y1 = h4 + (dx < 0 ? dx : 0) * 4;
            const int x0 = 4 * (dy > 0 ? dy : 0), x1 = w4 + (dy < 0 ? dy : 0) * 4;
            for (int bt = 0; bt < batch_num; ++bt)
                for (int Y = y0; Y < y1; ++Y)
                    for (int X = x0; X < x1; ++X)
                        ifm2[bt * per + ((int64_t)Y * w4 + X) * 9 + i] =
                            ifm[bt * per + ((int64_t)(Y - 4 * dx) * w4 + (X - 4 * dy)) * 9 + 8 - i];
        }
        for (int bt = 0; bt < batch_num && !rc; ++bt)
            for (int64_t n = 0; n < (int64_t)h4 * w4; ++n) {
                const double* u = use + bt * per + n * 9;
                int sbi = 0;                                              /* :232: argsort of scores_back_use */
                for (int k = 1; k < 9; ++k) if (u[k] < u[sbi]) sbi = k;
                const uint8_t m = ifm2[bt * per + n * 9 + sbi];           /* :233 */
                const int64_t s2 = 8 - sbi + n * 9 + (int64_t)(sbi % 3 - 1) * 4 * 9 +
                                   (int64_t)(sbi / 3 - 1) * 4 * w4 * 9;    /* :234-236 */
                if (s2 < 0 || s2 >= per) { rc = -2; break; }
                res[bt * per + s2] = !m;                                  /* :237-238 */
            }
        free(ifm2);
    } else {
        double* nu = (double*)malloc(sizeof(double) * (size_t)(batch_num * per));
        uint8_t* nm = (uint8_t*)malloc((size_t)(batch_num * per));
        memcpy(nu, use, sizeof(double) * (size_t)(batch_num * per));
        memcpy(nm, ifm, (size_t)(batch_num * per));
        for (int i = 0; i < 9; ++i) {                                     /* :166-171 (channel i only, src cloned) */
            const int dy = i % 3 - 1, dx = i / 3 - 1;
            const int y0 = 4 * (dx > 0 ? dx : 0), y1 = h4 + (dx < 0 ? dx : 0) * 4;
            const int x0 = 4 * (dy > 0 ? dy : 0), x1 = w4 + (dy < 0 ? dy : 0) * 4;
            for (int bt = 0; bt < batch_num; ++bt)
                for (int Y = y0; Y < y1; ++Y)
                    for (int X = x0; X < x1; ++X) {
                        const int64_t d = bt * per + ((int64_t)Y * w4 + X) * 9 + i;
                        const int64_t s = bt * per + ((int64_t)(Y - 4 * dx) * w4 + (X - 4 * dy)) * 9 + i;
                        nu[d] = use[s];
                        nm[d] = ifm[s];
                    }
        }
        for (int64_t e = 0; e < batch_num * per; ++e) if (nm[e]) nu[e] -= 10000.0;   /* :173 */
        for (int bt = 0; bt < batch_num; ++bt)
            for (int64_t n = 0; n < (int64_t)h4 * w4; ++n) {
                const double* u = nu + bt * per + n * 9;
                int sbi = 0;                                              /* :174 */
                for (int k = 1; k < 9; ++k) if (u[k] < u[sbi]) sbi = k;
                uint8_t m = nm[bt * per + n * 9 + sbi];                   /* :175 */
                int64_t s2 = sbi + n * 9 - (int64_t)(sbi % 3 - 1) * 4 * 9 -
                             (int64_t)(sbi / 3 - 1) * 4 * w4 * 9;          /* :176-178 */
                const int64_t hy = n / w / 4 - (sbi / 3 - 1) * 4;          /* :179-180 */
                const int64_t wx = n % w4 - (sbi % 3 - 1) * 4;             /* :181-182 */
                if (hy < 0 || hy >= h4 || wx < 0 || wx >= w4) m = 0;       /* :183-186 */
                if (s2 < 0) s2 = 0;                                        /* :187 */
                if (s2 > per - 1) s2 = per - 1;
                res[bt * per + s2] = !m;                                   /* :188-189, in source order */
            }
        free(nu);
        free(nm);
    }
    if (!rc)
        for (int64_t q = 0; q < NP; ++q) {                                /* :190-191 / :239-240 */
            if (slot[q] < 0) continue;
            const int bt = (int)(q / (h * w)), hh = (int)(q % (h * w)) / w, ww = (int)(q % (h * w)) % w;
            for (int a = 0; a < 3; ++a)
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 3; ++c)
                        for (int s = 0; s < 4; ++s)
                            out[slot[q] * 144 + (a * 4 + r) * 12 + c * 4 + s] =
                                res[bt * per + ((int64_t)(hh * 4 + r) * w4 + ww * 4 + s) * 9 + a * 3 + c];
        }
    free(slot); free(ifm); free(use); free(res);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * f3a  third-level inputs, models/pats.py:53-58: for every surviving L2 cell the source point
 * (cell centre on the 96-px patch, in 1/2-res units x 2) and the rounded target point, (x, y) order,
 * plus the patch row it belongs to.  torch.round = round-half-even (rintf under the default mode).
 * ifn2 [B,144], pts [B,144,2] (row, col) -> mk0, mk1 [P,2], b_ids [P]; returns P.
 * ---------------------------------------------------------------------------------------- */
int64_t oracle_third_inputs(const uint8_t* ifn2, const float* pts, int64_t B, float* mk0, float* mk1,
                            int64_t* b_ids) {
    int64_t P = 0;
    for (int64_t b = 0; b < B; ++b)
        for (int cell = 0; cell < 144; ++cell) {
            if (ifn2[b * 144 + cell]) continue;
            mk0[P * 2 + 0] = (float)(cell % 12 * 4 + 2) * 2.0f;
            mk0[P * 2 + 1] = (float)(cell / 12 * 4 + 2) * 2.0f;
            mk1[P * 2 + 0] = rintf(pts[(b * 144 + cell) * 2 + 1] * 4.0f) * 2.0f;
            mk1[P * 2 + 1] = rintf(pts[(b * 144 + cell) * 2 + 0] * 4.0f) * 2.0f;
            b_ids[P++] = b;
        }
    return P;
}

/* ------------------------------------------------------------------------------------------
 * f3b  scatter of the third-level results back onto the L2 grid, models/pats.py:59-67:
 * every L2 cell becomes 4x4 sub-cells; surviving cells take mkpts1_f / the label flag, the rest keep
 * the L2 point and stay "no match"; layout [B,12,12,4,4] -> [B,12,4,12,4] = 48x48 row-major.
 * ifn2 [B,144], pts [B,144,2], mkpts1 [P,16,2], label0 [P*16] (= label[:,0]) ->
 * ifn16 [B,2304], pts16 [B,2304,2]
 * ---------------------------------------------------------------------------------------- */
void oracle_refine_scatter(const uint8_t* ifn2, const float* pts, const float* mkpts1, const float* label0,
                           int64_t B, uint8_t* ifn16, float* pts16) {
    int64_t P = 0;
    for (int64_t b = 0; b < B; ++b)
        for (int cell = 0; cell < 144; ++cell) {
            const int cy = cell / 12, cx = cell % 12;
            const int nom = ifn2[b * 144 + cell];
            for (int sub = 0; sub < 16; ++sub) {
                const int sy = sub / 4, sx = sub % 4;
                const int64_t o = b * 2304 + (cy * 4 + sy) * 48 + cx * 4 + sx;
                float py = pts[(b * 144 + cell) * 2], px = pts[(b * 144 + cell) * 2 + 1];
                uint8_t f = 1;
                if (!nom) {
                    py = mkpts1[(P * 16 + sub) * 2];
                    px = mkpts1[(P * 16 + sub) * 2 + 1];
                    f = label0[P * 16 + sub] < -9.9f;
                }
                pts16[o * 2] = py;
                pts16[o * 2 + 1] = px;
                ifn16[o] = f;
            }
            if (!nom) ++P;
        }
}

/* ------------------------------------------------------------------------------------------
 * f2  get_result (layer_num = 2), utils/utils.py:189-213: composes the coarse patch origin, the
 * area scale and the fine offset into absolute pixel coordinates for every surviving sub-cell.
 * Level 0: rows = bs, n0 = size0[1]*size0[2] cells of size0[0] px; level 1: rows = K surviving
 * level-0 cells (in order), n1 = size1[1]*size1[2] sub-cells.  fp32, operation order as written
 * there.  Returns M and writes matches_l / matches_r [M,2].
 * ---------------------------------------------------------------------------------------- */
int64_t oracle_get_result(int bs, const uint8_t* ifn0, const uint8_t* ifn1, const float* ap0,
                          const float* ap1, const float* sc0, const float* sc1, const int* size0,
                          const int* size1, const uint8_t* choice0, const uint8_t* choice1, float* ml,
                          float* mr) {
    const int n0 = size0[1] * size0[2], n1 = size1[1] * size1[2];
    int64_t K = 0, M = 0;
    for (int bt = 0; bt < bs; ++bt)
        for (int i = 0; i < n0; ++i) {
            const int64_t e = (int64_t)bt * n0 + i;
            if (ifn0[e]) continue;
            float l0[2], r0[2];
            for (int d = 0; d < 2; ++d) {
                const float pos = (float)((d == 0 ? i / size0[2] : i % size0[2]) * size0[0]);
                float dl = pos + 0.5f * (float)size0[0];                              /* :204 */
                dl = dl - (1.5f * sc0[e * 2 + 1]) * (float)size0[0];                  /* :206 */
                const float dr = (ap0[e * 2 + d] - 1.5f * sc0[e * 2 + 0]) * (float)size0[0]; /* :207 */
                l0[d] = 0.0f + (choice0[bt] ? dl : dr);                               /* :211-212 */
                r0[d] = 0.0f + (choice0[bt] ? dr : dl);
            }
            for (int j = 0; j < n1; ++j) {
                const int64_t f = K * n1 + j;
                if (ifn1[f]) continue;
                for (int d = 0; d < 2; ++d) {
                    const float pos = (float)((d == 0 ? j / size1[2] : j % size1[2]) * size1[0]);
                    float dl = pos + 0.5f * (float)size1[0];
                    dl = dl * sc1[f * 2 + 1];                                         /* :209 */
                    const float dr = (ap1[f * 2 + d] * (float)size1[0]) * sc1[f * 2 + 0]; /* :210 */
                    ml[M * 2 + d] = l0[d] + (choice1[K] ? dl : dr);
                    mr[M * 2 + d] = r0[d] + (choice1[K] ? dr : dl);
                }
                ++M;
            }
            ++K;
        }
    return M;
}

/* ------------------------------------------------------------------------------------------
 * f4  attention(query, key, value), models/modules.py:84-88 (the core of MultiHeadedAttention,
 * :100-105, inside AttentionalGNN):
 *     scores = einsum('bdhn,bdhm->bhnm', query, key) / dim**.5
 *     prob   = softmax(scores, dim=-1)
 *     out    = einsum('bhnm,bdhm->bdhn', prob, value)
 * query [b,dim,heads,n], key / value [b,dim,heads,m] -> out [b,dim,heads,n], prob [b,heads,n,m]
 * (prob may be NULL).  Dot products and the softmax denominator accumulate in double and round once
 * (ATen leaves their order unspecified); exp / divide are fp32 like ATen's softmax.
 * ---------------------------------------------------------------------------------------- */
void oracle_attention(const float* q, const float* k, const float* v, int64_t b, int dim, int heads, int n,
                      int m, float* out, float* prob) {
    const float sq = (float)sqrt((double)dim);
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t bi = 0; bi < b; ++bi)
        for (int h = 0; h < heads; ++h) {
            float* row = (float*)malloc(sizeof(float) * (size_t)m);
            double* acc = (double*)malloc(sizeof(double) * (size_t)dim);
            for (int i = 0; i < n; ++i) {
                float mx = -INFINITY;
                for (int j = 0; j < m; ++j) {
                    double s = 0.0;
                    for (int d = 0; d < dim; ++d)
                        s += (double)q[((bi * dim + d) * heads + h) * (int64_t)n + i] *
                             (double)k[((bi * dim + d) * heads + h) * (int64_t)m + j];
                    row[j] = (float)s / sq;
                    if (row[j] > mx) mx = row[j];
                }
                double den = 0.0;
                for (int j = 0; j < m; ++j) {
                    row[j] = expf(row[j] - mx);
                    den += (double)row[j];
                }
                const float fden = (float)den;
                for (int d = 0; d < dim; ++d) acc[d] = 0.0;
                for (int j = 0; j < m; ++j) {
                    row[j] = row[j] / fden;
                    if (prob) prob[((bi * heads + h) * (int64_t)n + i) * m + j] = row[j];
                    for (int d = 0; d < dim; ++d)
                        acc[d] += (double)row[j] * (double)v[((bi * dim + d) * heads + h) * (int64_t)m + j];
                }
                for (int d = 0; d < dim; ++d) out[((bi * dim + d) * heads + h) * (int64_t)n + i] = (float)acc[d];
            }
            free(row);
            free(acc);
        }
}

/* ---------------------------------------------------------------------------------------------
 * AttentionalPropagation (reference models/modules.py:91-117) and the residual of AttentionalGNN (:131-133).
 *   conv1x1: y[b][o][t] = bias[o] + sum_c W[o][c] x[b][c][t]           nn.Conv1d(kernel_size=1)  (MLP :57-69, :98-99)
 *   MultiHeadedAttention.forward (:100-105): q, k, v = proj[i](.) viewed [b, dim, heads, n]; attention; merge
 *   AttentionalPropagation.forward (:114-117): mlp(cat([x, message], 1)), mlp = conv(2C,2C) BN ReLU conv(2C,C)
 * Weights in the reference's own (untransposed) layout.  BatchNorm1d: eval = running statistics, train = batch
 * statistics over (b, n) with the biased variance (what F.batch_norm normalises with).  Sums accumulate in double.
 * ---------------------------------------------------------------------------------------- */
static void oracle_conv1x1(const float* W, const float* bias, const float* x, int64_t b, int Cin, int Cout, int n,
                           float* y) {
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t bi = 0; bi < b; ++bi)
        for (int o = 0; o < Cout; ++o)
            for (int t = 0; t < n; ++t) {
                double s = 0.0;
                for (int c = 0; c < Cin; ++c) s += (double)W[(int64_t)o * Cin + c] * (double)x[(bi * Cin + c) * (int64_t)n + t];
                y[(bi * Cout + o) * (int64_t)n + t] = (float)s + bias[o];
            }
}

/* nn.Conv1d(kernel_size=1) on its own: final_proj (first_layer.py:34-36,105; second_layer.py:40-42,91) and the layers of
 * MLP (modules.py:57-69).  bias may not be NULL (pass zeros). */
void oracle_conv1d(const float* W, const float* bias, const float* x, int64_t b, int Cin, int Cout, int n, float* y) {
    oracle_conv1x1(W, bias, x, b, Cin, Cout, n, y);
}

/* BatchNorm1d followed by ReLU, in place on h [b][C][n] (the middle of MLP, modules.py:64-67): eval = running
 * statistics, train = batch statistics over (b, n) with the biased variance. */
void oracle_bn_relu(float* h, int64_t b, int C, int n, const float* gamma, const float* beta, const float* rmean,
                    const float* rvar, float eps, int bn_train) {
    for (int c = 0; c < C; ++c) {
        double mean = rmean[c], var = rvar[c];
        if (bn_train) {
            double s = 0.0, s2 = 0.0;
            for (int64_t bi = 0; bi < b; ++bi)
                for (int t = 0; t < n; ++t) s += h[(bi * C + c) * (int64_t)n + t];
            mean = s / (double)(b * n);
            for (int64_t bi = 0; bi < b; ++bi)
                for (int t = 0; t < n; ++t) {
                    const double d = h[(bi * C + c) * (int64_t)n + t] - mean;
                    s2 += d * d;
                }
            var = s2 / (double)(b * n);
        }
        const float inv = 1.0f / sqrtf((float)var + eps);
        for (int64_t bi = 0; bi < b; ++bi)
            for (int t = 0; t < n; ++t) {
                float* v = &h[(bi * C + c) * (int64_t)n + t];
                const float y = (*v - (float)mean) * inv * gamma[c] + beta[c];
                *v = y > 0.f ? y : 0.f;
            }
    }
}

void oracle_attentional_propagation(const float* x, const float* source, int64_t b, int C, int heads, int n, int m,
                                    const float* wq, const float* bq, const float* wk, const float* bk,
                                    const float* wv, const float* bv, const float* wm, const float* bm,
                                    const float* w1, const float* b1, const float* gamma, const float* beta,
                                    const float* rmean, const float* rvar, float eps, int bn_train,
                                    const float* w2, const float* b2, const float* residual, float* out) {
    const size_t qn = (size_t)b * C * n, kn = (size_t)b * C * m;
    float* q = (float*)malloc(sizeof(float) * qn);
    float* k = (float*)malloc(sizeof(float) * kn);
    float* v = (float*)malloc(sizeof(float) * kn);
    float* att = (float*)malloc(sizeof(float) * qn);
    float* msg = (float*)malloc(sizeof(float) * qn);
    float* cat = (float*)malloc(sizeof(float) * 2 * qn);
    float* hid = (float*)malloc(sizeof(float) * 2 * qn);
    oracle_conv1x1(wq, bq, x, b, C, C, n, q);
    oracle_conv1x1(wk, bk, source, b, C, C, m, k);
    oracle_conv1x1(wv, bv, source, b, C, C, m, v);
    oracle_attention(q, k, v, b, C / heads, heads, n, m, att, NULL);        /* .view(b, dim, heads, -1): same memory */
    oracle_conv1x1(wm, bm, att, b, C, C, n, msg);
    for (int64_t bi = 0; bi < b; ++bi)
        for (int c = 0; c < 2 * C; ++c)
            for (int t = 0; t < n; ++t)
                cat[(bi * 2 * C + c) * (int64_t)n + t] = c < C ? x[(bi * C + c) * (int64_t)n + t]
                                                                : msg[(bi * C + (c - C)) * (int64_t)n + t];
    oracle_conv1x1(w1, b1, cat, b, 2 * C, 2 * C, n, hid);
    oracle_bn_relu(hid, b, 2 * C, n, gamma, beta, rmean, rvar, eps, bn_train);
    oracle_conv1x1(w2, b2, hid, b, 2 * C, C, n, out);
    if (residual)
        for (size_t e = 0; e < qn; ++e) out[e] = residual[e] + out[e];
    free(q); free(k); free(v); free(att); free(msg); free(cat); free(hid);
}

/* ---------------------------------------------------------------------------------------------
 * The scale head: nn.Conv2d(C, 1, kernel_size=3, padding=1) on the descriptor grid, then
 * exp(sigmoid(v) * log(256) - log(256) / 2); two heads multiply (reference models/first_layer.py:106-107,
 * models/second_layer.py:92-98, models/third_layer.py:151-152).  x [b][C][ld], the grid is its first h*w columns;
 * weight [heads][C][3][3]; out [b][h*w].  The stencil accumulates in double.
 * ---------------------------------------------------------------------------------------- */
void oracle_scale_head(const float* x, int64_t b, int C, int ld, int h, int w, const float* weight, const float* bias,
                       int heads, float* out) {
    const float ln256 = (float)5.545177444479562, half = (float)2.772588722239781;
#pragma omp parallel for schedule(static)
    for (int64_t bi = 0; bi < b; ++bi)
        for (int py = 0; py < h; ++py)
            for (int px = 0; px < w; ++px) {
                float s = 1.0f;
                for (int hd = 0; hd < heads; ++hd) {
                    double acc = 0.0;
                    for (int c = 0; c < C; ++c)
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                const int yy = py + ky - 1, xx = px + kx - 1;
                                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                                acc += (double)weight[(((int64_t)hd * C + c) * 3 + ky) * 3 + kx] *
                                       (double)x[(bi * C + c) * (int64_t)ld + yy * w + xx];
                            }
                    const float v = (float)acc + bias[hd];
                    const float sig = 1.0f / (1.0f + expf(-v));
                    const float e = expf(sig * ln256 - half);
                    s = hd == 0 ? e : s * e;
                }
                out[bi * (int64_t)(h * w) + py * w + px] = s;
            }
}
