/*
 * cg_oracle.h -- CPU restatement of the reference's CG hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under conjugate_gradient_amd/ may call,
 * link or load this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker.
 *
 * Every function restates one reference function (cited per function in
 * cg_oracle.c, paths relative to /root/reference) with the same floating-point
 * operation ORDER: rounded multiply then rounded add (never an FMA), strict
 * left-to-right reductions.  Build with -ffp-contract=off (the reference is
 * built -std=c99, which implies it).
 *
 * Pinned: tests/test_oracle.py checks this file against (i) every known-answer
 * vector the reference's own tests hold (tests/test_ell.c, test_csr.c,
 * test_coo.c) and (ii) goldens produced by the reference itself compiled from
 * /root/reference/src (oracle/_ref, recipe in oracle/Makefile, generator
 * oracle/gen_golden.py), committed under tests/golden/.
 */
#ifndef CG_ORACLE_H
#define CG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BLAS-1 (src/vec.c) ---- */
void   orc_vec_sub(const double *a, const double *b, double *out, int n);
void   orc_vec_add(const double *a, const double *b, double *out, int n);
void   orc_vec_assign(double *dst, const double *src, int n);
double orc_vec_dot(const double *a, const double *b, int n);
double orc_vec_l1norm(const double *a, int n);
void   orc_vec_axpy(const double *a, const double *b, double alpha, double *out, int n);
double orc_rmse(const double *a, const double *b, int n);
double orc_euclidean_distance(const double *a, const double *b, int n);
double orc_max_difference(const double *a, const double *b, int n);

/* ---- SpMV (src/coo.c, src/csr.c, src/ell.c) ---- */
void orc_mv_coo_symmetric(int n, int nnz_upper, const double *diag, const double *val,
                          const int *rows, const int *cols, const double *v, double *out);
void orc_mv_csr_symmetric(int n, const double *diag, const double *upper, const int *col_index,
                          const int *row_ptr, const double *v, double *out);
void orc_mv_ell(int n, int K, const double *ev, const int *ec, const double *x, double *y);
/* return 0 ok, 1 = stopped early on a non-upper entry (the reference prints and returns) */
int  orc_mv_ell_symmetric_upper(int n, int K, const double *diag, const double *ev, const int *ec,
                                const double *x, double *y);
void orc_mv_ell_symmetric_upper_opt(int n, int K, const double *diag, const double *ev, const int *ec,
                                    const double *x, double *y);
int  orc_mv_ell_symmetric_upper_colmajor(int n, int K, const double *diag, const double *ev,
                                         const int *ec, const double *x, double *y);
void orc_mv_ell_symmetric_full_colmajor(int n, int K, const double *diag, const double *ev,
                                        const int *ec, const double *x, double *y);
void orc_mv_ell_symmetric_full_colmajor_u64(int n, int K, const double *diag, const double *ev,
                                            const uint64_t *ec, const double *x, double *y);

/* ---- format construction (src/ell.c, src/csr.c) ---- */
int  orc_compute_max_nnz_row_upper(int n, int nnz_upper, const int *rows, const int *cols);
int  orc_compute_max_nnz_row_full(int n, int nnz_upper, const int *rows, const int *cols);
/* converters return 0 ok / 1 stopped early (partial output, as the reference leaves it) */
int  orc_coo_to_ell_symmetric_upper(int n, int nnz_upper, const double *val, const int *rows,
                                    const int *cols, double *ev, int *ec, int K);
int  orc_coo_to_ell_symmetric_upper_colmajor(int n, int nnz_upper, const double *val, const int *rows,
                                             const int *cols, double *ev, int *ec, int K);
int  orc_coo_to_ell_symmetric_full_colmajor(int n, int nnz_upper, const double *val, const int *rows,
                                            const int *cols, double *ev, int *ec, int K);
int  orc_coo_to_ell_symmetric_full_colmajor_u64(int n, int nnz_upper, const double *val, const int *rows,
                                                const int *cols, double *ev, uint64_t *ec, int K);
int  orc_coo_to_csr(int n, int nnz_upper, const int *rows, int *row_ptr);
void orc_sorting_coo_to_csr(int n, int nnz, const double *val, const int *rows, const int *cols,
                            int *row_ptr, int *csr_cols, double *csr_vals);

/* ---- CG (src/cg.c) ----
 * One loop, five SpMV flavours.  `traj` (may be NULL) receives 4 doubles per
 * executed iteration k = 1.. : { l1norm(r)/2, r.r, alpha, beta } (beta = NaN on
 * the converging iteration, where the reference never computes it); traj[0..3]
 * is iteration "0": { initial l1/2, r0.r0, NaN, NaN }.  traj must hold
 * 4*max_iter doubles.  Returns what the reference returns: iter+1 or -1.
 */
enum { ORC_FMT_COO = 0, ORC_FMT_CSR = 1, ORC_FMT_ELL_UPPER_ROWMAJOR = 2,
       ORC_FMT_ELL_UPPER_COLMAJOR = 3, ORC_FMT_ELL_FULL_COLMAJOR = 4 };

int orc_cg(int fmt, int n, const double *diag, int count, const double *vals,
           const int *idx_a, const int *idx_b, int K,
           const double *b, double *x, int max_iter, double tol, double *traj);

#ifdef __cplusplus
}
#endif
#endif
