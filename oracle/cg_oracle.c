/*
 * cg_oracle.c -- CPU restatement of the reference CG path.  TEST INFRASTRUCTURE
 * ONLY (see cg_oracle.h).  Citations are file:line under /root/reference.
 *
 * Arithmetic contract: every `acc = acc + a*b` below is a rounded multiply
 * followed by a rounded add, in the order written; reductions run strictly
 * left to right.  Compile with -ffp-contract=off.
 */
#include "cg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ BLAS-1 */

/* src/vec.c:39-45 */
void orc_vec_sub(const double *a, const double *b, double *out, int n)
{
    for (int k = 0; k < n; ++k) out[k] = a[k] - b[k];
}

/* src/vec.c:48-54 */
void orc_vec_add(const double *a, const double *b, double *out, int n)
{
    for (int k = 0; k < n; ++k) out[k] = a[k] + b[k];
}

/* src/vec.c:57-64 (a memcpy) */
void orc_vec_assign(double *dst, const double *src, int n)
{
    memcpy(dst, src, (size_t)n * sizeof *dst);
}

/* src/vec.c:67-75: strict left-to-right sum of rounded products */
double orc_vec_dot(const double *a, const double *b, int n)
{
    double s = 0.0;
    for (int k = 0; k < n; ++k) { double t = a[k] * b[k]; s = s + t; }
    return s;
}

/* src/vec.c:77-85 */
double orc_vec_l1norm(const double *a, int n)
{
    double s = 0.0;
    for (int k = 0; k < n; ++k) s = s + fabs(a[k]);
    return s;
}

/* src/vec.c:88-94; out may alias a or b (element-wise, so aliasing is benign) */
void orc_vec_axpy(const double *a, const double *b, double alpha, double *out, int n)
{
    for (int k = 0; k < n; ++k) { double t = alpha * b[k]; out[k] = a[k] + t; }
}

static double sum_sq_diff(const double *a, const double *b, int n)
{
    double s = 0.0;
    for (int k = 0; k < n; ++k) { double d = a[k] - b[k]; double q = d * d; s = s + q; }
    return s;
}

/* src/vec.c:97-104 */
double orc_rmse(const double *a, const double *b, int n) { return sqrt(sum_sq_diff(a, b, n) / n); }

/* src/vec.c:107-114 */
double orc_euclidean_distance(const double *a, const double *b, int n) { return sqrt(sum_sq_diff(a, b, n)); }

/* src/vec.c:117-126 */
double orc_max_difference(const double *a, const double *b, int n)
{
    double m = 0.0;
    for (int k = 0; k < n; ++k) { double d = fabs(a[k] - b[k]); if (d > m) m = d; }
    return m;
}

/* -------------------------------------------------------------------- SpMV */

/* y = 0 then y += diag*x: the add to +0.0 is kept so that a -0.0 product
 * becomes +0.0 exactly as in the reference (memset/zero loop, then +=). */
static void diag_pass(int n, const double *diag, const double *x, double *y)
{
    for (int i = 0; i < n; ++i) { double t = diag[i] * x[i]; y[i] = 0.0 + t; }
}

/* src/coo.c:202-233 */
void orc_mv_coo_symmetric(int n, int nnz_upper, const double *diag, const double *val,
                          const int *rows, const int *cols, const double *v, double *out)
{
    diag_pass(n, diag, v, out);
    for (int e = 0; e < nnz_upper; ++e) {
        int i = rows[e], j = cols[e];
        double a = val[e];
        double t = a * v[j]; out[i] = out[i] + t;
        double u = a * v[i]; out[j] = out[j] + u;
    }
}

/* src/csr.c:190-229 */
void orc_mv_csr_symmetric(int n, const double *diag, const double *upper, const int *col_index,
                          const int *row_ptr, const double *v, double *out)
{
    diag_pass(n, diag, v, out);
    for (int i = 0; i < n; ++i)
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            int j = col_index[e];
            double a = upper[e];
            double t = a * v[j]; out[i] = out[i] + t;
            double u = a * v[i]; out[j] = out[j] + u;
        }
}

/* src/ell.c:136-159: generic row-major ELL, pad = negative column */
void orc_mv_ell(int n, int K, const double *ev, const int *ec, const double *x, double *y)
{
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int s = 0; s < K; ++s) {
            int c = ec[i * K + s];
            if (c >= 0) { double t = ev[i * K + s] * x[c]; acc = acc + t; }
        }
        y[i] = acc;
    }
}

/* src/ell.c:168-211: row-major, upper only, scatter; stops (returns) at the
 * first stored column that is neither -1 nor > row. */
int orc_mv_ell_symmetric_upper(int n, int K, const double *diag, const double *ev, const int *ec,
                               const double *x, double *y)
{
    for (int i = 0; i < n; ++i) y[i] = 0.0;
    for (int i = 0; i < n; ++i) {
        { double t = diag[i] * x[i]; y[i] = y[i] + t; }
        for (int s = 0; s < K; ++s) {
            int c = ec[i * K + s];
            if (c == -1) continue;
            if (c <= i) return 1;
            double a = ev[i * K + s];
            double t = a * x[c]; y[i] = y[i] + t;
            double u = a * x[i]; y[c] = y[c] + u;
        }
    }
    return 0;
}

/* src/ell.c:213-251: same without the upper check */
void orc_mv_ell_symmetric_upper_opt(int n, int K, const double *diag, const double *ev, const int *ec,
                                    const double *x, double *y)
{
    for (int i = 0; i < n; ++i) y[i] = 0.0;
    for (int i = 0; i < n; ++i) {
        { double t = diag[i] * x[i]; y[i] = y[i] + t; }
        for (int s = 0; s < K; ++s) {
            int c = ec[i * K + s];
            if (c == -1) continue;
            double a = ev[i * K + s];
            double t = a * x[c]; y[i] = y[i] + t;
            double u = a * x[i]; y[c] = y[c] + u;
        }
    }
}

/* src/ell.c:946-999: slot-major, upper only, scatter; slot-outer / row-inner */
int orc_mv_ell_symmetric_upper_colmajor(int n, int K, const double *diag, const double *ev,
                                        const int *ec, const double *x, double *y)
{
    diag_pass(n, diag, x, y);
    for (int s = 0; s < K; ++s)
        for (int i = 0; i < n; ++i) {
            int c = ec[s * n + i];
            if (c == -1) continue;
            if (c <= i) return 1;
            double a = ev[s * n + i];
            double t = a * x[c]; y[i] = y[i] + t;
            double u = a * x[i]; y[c] = y[c] + u;
        }
    return 0;
}

/* src/ell.c:1004-1040: slot-major, full rows, gather only.  Per row the order
 * is diag, slot 0, slot 1, ... with pads skipped, which is what the
 * slot-outer/row-inner loop of the reference produces for each y[i]. */
void orc_mv_ell_symmetric_full_colmajor(int n, int K, const double *diag, const double *ev,
                                        const int *ec, const double *x, double *y)
{
    diag_pass(n, diag, x, y);
    for (int s = 0; s < K; ++s)
        for (int i = 0; i < n; ++i) {
            int c = ec[s * n + i];
            if (c == -1) continue;
            double t = ev[s * n + i] * x[c];
            y[i] = y[i] + t;
        }
}

/* src/ell.c:1045-1081: uint64_t columns; the reference narrows to int before
 * the pad test (`int col = ell_cols[offset]`), so (uint64_t)-1 reads as -1. */
void orc_mv_ell_symmetric_full_colmajor_u64(int n, int K, const double *diag, const double *ev,
                                            const uint64_t *ec, const double *x, double *y)
{
    diag_pass(n, diag, x, y);
    for (int s = 0; s < K; ++s)
        for (int i = 0; i < n; ++i) {
            int c = (int)ec[s * n + i];
            if (c == -1) continue;
            double t = ev[s * n + i] * x[c];
            y[i] = y[i] + t;
        }
}

/* ------------------------------------------------------ format construction */

/* src/ell.c:315-352 */
int orc_compute_max_nnz_row_upper(int n, int nnz_upper, const int *rows, const int *cols)
{
    int *cnt = calloc((size_t)(n > 0 ? n : 1), sizeof *cnt);
    if (!cnt) return -1;
    for (int e = 0; e < nnz_upper; ++e) {
        if (cols[e] <= rows[e]) { free(cnt); return -1; }
        cnt[rows[e]]++;
    }
    int best = 0;
    for (int i = 0; i < n; ++i) if (cnt[i] > best) best = cnt[i];
    free(cnt);
    return best;
}

/* src/ell.c:360-396 */
int orc_compute_max_nnz_row_full(int n, int nnz_upper, const int *rows, const int *cols)
{
    int *cnt = calloc((size_t)(n > 0 ? n : 1), sizeof *cnt);
    if (!cnt) return -1;
    for (int e = 0; e < nnz_upper; ++e) {
        if (cols[e] <= rows[e]) { free(cnt); return -1; }
        cnt[rows[e]]++;
        cnt[cols[e]]++;
    }
    int best = 0;
    for (int i = 0; i < n; ++i) if (cnt[i] > best) best = cnt[i];
    free(cnt);
    return best;
}

static void ell_clear(double *ev, int *ec, long total)
{
    for (long k = 0; k < total; ++k) { ev[k] = 0.0; ec[k] = -1; }
}

/* src/ell.c:262-309: row-major, upper only.  NOTE the reference does not check
 * for slot overflow in this variant. */
int orc_coo_to_ell_symmetric_upper(int n, int nnz_upper, const double *val, const int *rows,
                                   const int *cols, double *ev, int *ec, int K)
{
    ell_clear(ev, ec, (long)n * K);
    int *fill = calloc((size_t)(n > 0 ? n : 1), sizeof *fill);
    if (!fill) return 1;
    for (int e = 0; e < nnz_upper; ++e) {
        int i = rows[e], j = cols[e];
        if (j <= i) { free(fill); return 1; }
        int at = i * K + fill[i]++;
        ev[at] = val[e];
        ec[at] = j;
    }
    free(fill);
    return 0;
}

/* src/ell.c:714-776: slot-major, upper only */
int orc_coo_to_ell_symmetric_upper_colmajor(int n, int nnz_upper, const double *val, const int *rows,
                                            const int *cols, double *ev, int *ec, int K)
{
    int *fill = calloc((size_t)(n > 0 ? n : 1), sizeof *fill);
    if (!fill) return 1;
    ell_clear(ev, ec, (long)n * K);
    for (int e = 0; e < nnz_upper; ++e) {
        int i = rows[e], j = cols[e];
        if (j <= i) { free(fill); return 1; }
        int s = fill[i]++;
        if (s >= K) { free(fill); return 1; }
        ev[s * n + i] = val[e];
        ec[s * n + i] = j;
    }
    free(fill);
    return 0;
}

/* src/ell.c:791-858: slot-major, both triangles.  Entry e=(i,j,v) lands first
 * in row i (column j), then in row j (column i); each row's next free slot is
 * a running counter, so a slot index is the rank of the placement among all
 * earlier placements into that row.  Stops at the first non-upper entry or the
 * first placement whose slot would be >= K, leaving what was written. */
int orc_coo_to_ell_symmetric_full_colmajor(int n, int nnz_upper, const double *val, const int *rows,
                                           const int *cols, double *ev, int *ec, int K)
{
    int *fill = calloc((size_t)(n > 0 ? n : 1), sizeof *fill);
    if (!fill) return 1;
    ell_clear(ev, ec, (long)n * K);
    for (int e = 0; e < nnz_upper; ++e) {
        int i = rows[e], j = cols[e];
        double v = val[e];
        if (j <= i) { free(fill); return 1; }
        int si = fill[i]++;
        if (si >= K) { free(fill); return 1; }
        ev[si * n + i] = v; ec[si * n + i] = j;
        int sj = fill[j]++;
        if (sj >= K) { free(fill); return 1; }
        ev[sj * n + j] = v; ec[sj * n + j] = i;
    }
    free(fill);
    return 0;
}

/* src/ell.c:873-940 */
int orc_coo_to_ell_symmetric_full_colmajor_u64(int n, int nnz_upper, const double *val, const int *rows,
                                               const int *cols, double *ev, uint64_t *ec, int K)
{
    int *fill = calloc((size_t)(n > 0 ? n : 1), sizeof *fill);
    if (!fill) return 1;
    for (long k = 0; k < (long)n * K; ++k) { ev[k] = 0.0; ec[k] = (uint64_t)-1; }
    for (int e = 0; e < nnz_upper; ++e) {
        int i = rows[e], j = cols[e];
        double v = val[e];
        if (j <= i) { free(fill); return 1; }
        int si = fill[i]++;
        if (si >= K) { free(fill); return 1; }
        ev[si * n + i] = v; ec[si * n + i] = (uint64_t)j;
        int sj = fill[j]++;
        if (sj >= K) { free(fill); return 1; }
        ev[sj * n + j] = v; ec[sj * n + j] = (uint64_t)i;
    }
    free(fill);
    return 0;
}

/* src/csr.c:233-272: row_ptr of a ROW-SORTED upper COO (values/cols are reused
 * in place by the caller).  row_ptr[r+1] = index of the first entry whose row
 * is > r; trailing rows get nnz_upper.  (The reference also prints a line per
 * trailing row; the oracle does not print.) */
int orc_coo_to_csr(int n, int nnz_upper, const int *rows, int *row_ptr)
{
    int r = 0;
    row_ptr[0] = 0;
    for (int e = 0; e < nnz_upper; ++e)
        while (r < rows[e]) row_ptr[++r] = e;
    while (r < n) row_ptr[++r] = nnz_upper;
    return 0;
}

/* src/csr.c:287-336: stable counting sort by row */
void orc_sorting_coo_to_csr(int n, int nnz, const double *val, const int *rows, const int *cols,
                            int *row_ptr, int *csr_cols, double *csr_vals)
{
    for (int i = 0; i <= n; ++i) row_ptr[i] = 0;
    for (int e = 0; e < nnz; ++e) row_ptr[rows[e] + 1]++;
    for (int i = 0; i < n; ++i) row_ptr[i + 1] += row_ptr[i];
    int *next = malloc((size_t)(n > 0 ? n : 1) * sizeof *next);
    if (!next) exit(EXIT_FAILURE);
    memcpy(next, row_ptr, (size_t)n * sizeof *next);
    for (int e = 0; e < nnz; ++e) {
        int at = next[rows[e]]++;
        csr_cols[at] = cols[e];
        csr_vals[at] = val[e];
    }
    free(next);
}

/* ---------------------------------------------------------------------- CG */

static void apply(int fmt, int n, const double *diag, int count, const double *vals,
                  const int *ia, const int *ib, int K, const double *in, double *out)
{
    switch (fmt) {
    case ORC_FMT_COO: /* src/cg.c:47,68: (rows, cols) */
        orc_mv_coo_symmetric(n, count, diag, vals, ia, ib, in, out); break;
    case ORC_FMT_CSR: /* src/cg.c:148,171: ia = row_ptr, ib = cols */
        orc_mv_csr_symmetric(n, diag, vals, ib, ia, in, out); break;
    case ORC_FMT_ELL_UPPER_ROWMAJOR: /* src/cg.c:250,270 */
        orc_mv_ell_symmetric_upper(n, K, diag, vals, ia, in, out); break;
    case ORC_FMT_ELL_UPPER_COLMAJOR: /* src/cg.c:345,366 */
        orc_mv_ell_symmetric_upper_colmajor(n, K, diag, vals, ia, in, out); break;
    default: /* src/cg.c:441,463 */
        orc_mv_ell_symmetric_full_colmajor(n, K, diag, vals, ia, in, out); break;
    }
}

/* src/cg.c:21-118 (COO), :121-222 (CSR), :225-316 (ELL row-major upper),
 * :319-411 (ELL slot-major upper), :414-508 (ELL slot-major full).  The five
 * loops are the same text around a different SpMV:
 *   x = 0; r = b - A x; p = r; rr = r.r
 *   for iter = 1 .. max_iter-1:
 *     Ap = A p; alpha = rr / (p.Ap); x += alpha p; r -= alpha Ap; rr' = r.r
 *     if l1(r)/2 < tol: return iter+1
 *     beta = rr'/rr; p = r + beta p; rr = rr'
 *   return -1
 */
int orc_cg(int fmt, int n, const double *diag, int count, const double *vals,
           const int *idx_a, const int *idx_b, int K,
           const double *b, double *x, int max_iter, double tol, double *traj)
{
    const double half = 2.0; /* norm_factor, src/cg.c:427 */
    double *r = malloc((size_t)n * sizeof *r);
    double *p = malloc((size_t)n * sizeof *p);
    double *Ap = malloc((size_t)n * sizeof *Ap);
    int ret = -1;

    for (int i = 0; i < n; ++i) x[i] = 0.0;
    apply(fmt, n, diag, count, vals, idx_a, idx_b, K, x, r);
    orc_vec_sub(b, r, r, n);
    orc_vec_assign(p, r, n);
    double rr = orc_vec_dot(r, r, n);
    if (traj) { traj[0] = orc_vec_l1norm(r, n) / half; traj[1] = rr; traj[2] = NAN; traj[3] = NAN; }

    for (int iter = 1; iter < max_iter; ++iter) {
        apply(fmt, n, diag, count, vals, idx_a, idx_b, K, p, Ap);
        double alpha = rr / orc_vec_dot(p, Ap, n);
        orc_vec_axpy(x, p, alpha, x, n);
        orc_vec_axpy(r, Ap, -alpha, r, n);
        double rr_new = orc_vec_dot(r, r, n);
        double res = orc_vec_l1norm(r, n) / half;
        if (traj) { traj[4 * iter] = res; traj[4 * iter + 1] = rr_new; traj[4 * iter + 2] = alpha; traj[4 * iter + 3] = NAN; }
        if (res < tol) { ret = iter + 1; break; }
        double beta = rr_new / rr;
        if (traj) traj[4 * iter + 3] = beta;
        orc_vec_axpy(r, p, beta, p, n);
        rr = rr_new;
    }
    free(r); free(p); free(Ap);
    return ret;
}
