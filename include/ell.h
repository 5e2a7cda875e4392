/*
 * ell.h -- ELLPACK pieces of the reference API (reference include/ell.h:75-158,
 * src/ell.c), host pointers, served by libcg_hip.  Layout names follow the
 * reference: "colmajor" = slot-major, entry (row i, slot s) at s*n+i; pad = value 0.0, col -1.
 * Functions the reference declares but never defines (csr_to_ell_*, coo_to_ell_symmetric_full,
 * ell.h:37-72 there) are not provided.  Its host-side inspection helpers are (csrc/ell_debug.c: plain
 * host loops, same stdout), so that the reference's own tests link against this library unchanged.
 */
#ifndef CG_HIP_DROPIN_ELL_H
#define CG_HIP_DROPIN_ELL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* reference ell.h:106-111, ell.c:315-396: -1 (and a message) on a non-upper entry */
int compute_max_nnz_row_upper(int n, int upper_nnz, int *coo_rows, int *coo_cols);
int compute_max_nnz_row_full(int n, int upper_nnz, const int *coo_rows, const int *coo_cols);

/* reference ell.h:75-82, ell.c:262-309 (row-major, upper only) */
void coo_to_ell_symmetric_upper(int n, int upper_nnz, double *coo_values, int *coo_rows, int *coo_cols,
                                double *ell_values, int *ell_cols, int max_nnz_row);
/* reference ell.h:115-123, ell.c:714-776 (slot-major, upper only) */
void coo_to_ell_symmetric_upper_colmajor(int n, int upper_nnz, double *coo_values_upper, int *coo_rows,
                                         int *coo_cols, double *ell_values, int *ell_cols, int max_nnz_row);
/* reference ell.h:125-132, ell.c:791-858 (slot-major, both triangles): index arrays bit-exact */
void coo_to_ell_symmetric_full_colmajor(int n, int upper_nnz, double *coo_values_upper, int *coo_rows,
                                        int *coo_cols, double *ell_values, int *ell_cols, int max_nnz_row);
/* reference ell.h:27-34, ell.c:873-940 (uint64_t column ids) */
void coo_to_ell_symmetric_full_colmajor_sdtint(int n, int upper_nnz, double *coo_values_upper, int *coo_rows,
                                               int *coo_cols, double *ell_values, uint64_t *ell_cols,
                                               int max_nnz_row);

/* reference ell.h:85-87, ell.c:136-159: generic row-major ELL, no separate diagonal */
void mv_ell(int n, int max_nnz_row, double *ell_values, int *ell_cols, double *x, double *y);
/* reference ell.h:99-104, ell.c:168-211 and ell.h:89-96, ell.c:213-251: row-major upper-only ELL */
void mv_ell_symmetric_upper(int n, int max_nnz_row, double *diag, double *ell_values, int *ell_cols,
                            double *x, double *y);
void mv_ell_symmetric_upper_opt(int n, int max_nnz_row, double *diag, double *ell_values, int *ell_cols,
                                double *x, double *y);
/* reference ell.h:134-141, ell.c:946-999: slot-major upper-only ELL */
void mv_ell_symmetric_upper_colmajor(int n, int max_nnz_row, double *diag, double *ell_values, int *ell_cols,
                                     double *x, double *y);
/* reference ell.h:143-150, ell.c:1004-1040: slot-major full ELL (the hot SpMV) */
void mv_ell_symmetric_full_colmajor(int n, int max_nnz_row, double *diag, double *ell_values, int *ell_cols,
                                    double *x, double *y);
/* reference ell.h:152-158, ell.c:1045-1081 */
void mv_ell_symmetric_full_colmajor_sdtint(int n, int max_nnz_row, double *diag, double *ell_values,
                                           uint64_t *ell_cols, double *x, double *y);

/* ---- host-side inspection helpers (no GPU work; same text on stdout as the reference) ---- */
/* reference ell.h:7, ell.c:11-19 */
void print_dense_matrix(double **A, int n);
/* reference ell.h:9-11, ell.c:21-55: upper-only ELL (row-major / slot-major) mirrored into a dense matrix the caller zeroed */
void convert_ell_to_dense(int n, double *diag, double *ell_values, int *ell_cols, int max_nnz_row, double **A);
void convert_ell_to_dense_colmajor(int n, double *diag, double *ell_values, int *ell_cols, int max_nnz_row, double **A);
/* reference ell.h:13-25, ell.c:66-134: full ELL -> dense (A is cleared first) */
void convert_ell_full_to_dense_rowmajor(int n, const double diag[], const double ell_values[], const int ell_cols[],
                                        int max_nnz_row, double **A);
void convert_ell_full_to_dense_colmajor(int n, const double diag[], const double ell_values[], const int ell_cols[],
                                        int max_nnz_row, double **A);
/* reference ell.h:113, ell.c:401-520: row-major ELL dump + padding statistics */
void analyze_ell_matrix(int n, int nnz_max, double *ell_values, int *ell_col_idx);
/* reference ell.h:160-162, ell.c:529-620: slot-major ELL dump + statistics */
void analyze_ell_matrix_colmajor(int n, int nnz_max, const double *ell_values, const int *ell_col_idx);
/* reference ell.h:164-166, ell.c:622-707: slot-major full ELL, statistics only; returns the padded fraction
 * (called by the reference's tests/tests_cg.c:1214) */
double analyze_ell_matrix_full_colmajor(int n, int nnz_max, const double *ell_values, const int *ell_col_idx);
#ifdef __cplusplus
}
#endif
#endif
