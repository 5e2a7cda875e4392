/*
 * ell.h -- ELLPACK pieces of the reference API (reference include/ell.h:75-158,
 * src/ell.c), host pointers, served by libcg_hip.  Layout names follow the
 * reference: "colmajor" = slot-major, entry (row i, slot s) at s*n+i; pad = value 0.0, col -1.
 * Functions the reference declares but never defines (csr_to_ell_*, coo_to_ell_symmetric_full,
 * ell.h:37-72 there) and its debug printers are not provided.
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
#ifdef __cplusplus
}
#endif
#endif
