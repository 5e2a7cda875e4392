/*
 * cg.h -- the CG solvers of the reference API, served by libcg_hip (MI355X).
 * Same names, argument order and return convention as reference include/cg.h:4-56
 * (definitions: reference src/cg.c:21,121,225,319,414): all pointers are HOST
 * pointers owned by the caller; x is overwritten (it is zeroed first, the initial
 * guess is ignored); returns iter+1 (>= 2) on convergence of  l1norm(r)/2 < tol,
 * -1 otherwise (also on any GPU error, which is reported on stderr).
 */
#ifndef CG_HIP_DROPIN_CG_H
#define CG_HIP_DROPIN_CG_H
#include <stdio.h>
#include <stdlib.h>
#ifdef __cplusplus
extern "C" {
#endif

/* reference include/cg.h:4-6, src/cg.c:21-118: diag + strictly-upper COO (rows, cols) */
int conjugate_gradient_coo(const int n, double *diag, int coo_length, double *upper,
                           int *rows, int *cols, double *b, double *x,
                           int max_iter, double tol);

/* reference include/cg.h:8-19, src/cg.c:121-222: diag + strictly-upper CSR (rows_ptr[n+1], cols) */
int conjugate_gradient_csr(const int n, double *diag, int upper_count, double *upper,
                           int *rows_ptr, int *cols, double *b, double *x,
                           int max_iter, double tol);

/* reference include/cg.h:21-32, src/cg.c:225-316: row-major upper-only ELL */
int conjugate_gradient_ell(const int n, double *diag, int upper_count, double *ell_values,
                           int *ell_col, int max_nnz_row, double *b, double *x,
                           int max_iter, double tol);

/* reference include/cg.h:34-44, src/cg.c:319-411: slot-major ("column-major") upper-only ELL */
int conjugate_gradient_ell_colmajor(const int n, double *diag, int upper_count, double *ell_values,
                                    int *ell_col, int max_nnz_row, double *b, double *x,
                                    int max_iter, double tol);

/* reference include/cg.h:46-56, src/cg.c:414-508: slot-major full ELL (the hot path) */
int conjugate_gradient_ell_full_colmajor(const int n, double *diag, int upper_count, double *ell_values,
                                         int *ell_col, int max_nnz_row, double *b, double *x,
                                         int max_iter, double tol);

#ifdef __cplusplus
}
#endif
#endif
