/* cg_omp.c -- the reference's full-ELL CG loop (src/cg.c:414-508: same statements, same stop test) with its sweeps
 * parallelised over rows with OpenMP.  TEST / BENCHMARK INFRASTRUCTURE ONLY: bench.py times it as the "all host cores"
 * CPU figure beside the reference's own single-threaded code.  It is NOT the reference's execution model (the reference
 * is single-threaded) and NOT bit-identical to it: the dot products and the L1 norm are OpenMP reductions (per-thread
 * partial sums), the SpMV is per-row exact (diag first, slots 0..K-1, pads skipped: src/ell.c:1004-1040). */
#include <math.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_omp_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void spmv_rows(long n, int K, const double *diag, const double *ev, const int *ec, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double acc = 0.0 + diag[i] * x[i];
        for (int s = 0; s < K; ++s) {
            const int c = ec[(long)s * n + i];
            if (c != -1) acc += ev[(long)s * n + i] * x[c];
        }
        y[i] = acc;
    }
}

/* returns iter + 1 at convergence, -1 otherwise; x is overwritten (zero initial guess, cg.c:433) */
int orc_cg_ell_full_omp(int n_, const double *diag, const double *ev, const int *ec, int K, const double *b, double *x,
                        int max_iter, double tol)
{
    const long n = n_;
    double    *r = malloc((size_t)n * sizeof *r), *p = malloc((size_t)n * sizeof *p), *Ap = malloc((size_t)n * sizeof *Ap);
    if (!r || !p || !Ap) {
        free(r); free(p); free(Ap);
        return -1;
    }
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) x[i] = 0.0;
    spmv_rows(n, K, diag, ev, ec, x, Ap);
    double rr = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rr)
    for (long i = 0; i < n; ++i) {
        r[i] = b[i] - Ap[i];
        p[i] = r[i];
        rr += r[i] * r[i];
    }
    int ret = -1;
    for (int iter = 1; iter < max_iter; ++iter) {
        spmv_rows(n, K, diag, ev, ec, p, Ap);
        double pAp = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : pAp)
        for (long i = 0; i < n; ++i) pAp += p[i] * Ap[i];
        const double alpha = rr / pAp;
        double       rr_new = 0.0, l1 = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rr_new, l1)
        for (long i = 0; i < n; ++i) {
            x[i] = x[i] + alpha * p[i];
            r[i] = r[i] + (-alpha) * Ap[i];
            rr_new += r[i] * r[i];
            l1 += fabs(r[i]);
        }
        if (l1 / 2.0 < tol) {
            ret = iter + 1;
            break;
        }
        const double beta = rr_new / rr;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];
        rr = rr_new;
    }
    free(r); free(p); free(Ap);
    return ret;
}
