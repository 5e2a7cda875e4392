/*
 * csr.h -- CSR pieces of the reference API (reference include/csr.h:26-53,
 * src/csr.c:190-336), host pointers, served by libcg_hip.
 */
#ifndef CG_HIP_DROPIN_CSR_H
#define CG_HIP_DROPIN_CSR_H
#include <stdio.h>
#include <stdlib.h>
#ifdef __cplusplus
extern "C" {
#endif
/* reference csr.h:26-31, csr.c:233-272: row_ptr[n+1] of a row-sorted upper COO; values/cols are reused in
 * place by the caller.  Prints "Update to csr_row_ptr[..]" per trailing empty row like the reference. */
int coo_to_csr(int n, int upper_count, double *coo_upper_values, int *coo_row_indx,
               int *coo_col_indx, int *csr_row_ptr);

/* reference csr.h:33-39, csr.c:190-229: out = A v, A = diag + upper CSR mirrored; bit-identical */
void mv_csr_symmetric(int n, const double *diag, const double *upper, const int *col_index,
                      const int *row_ptr, const double *v, double *out);

/* reference csr.h:47-53, csr.c:287-336: stable counting sort of any COO by row */
void sorting_coo_to_csr(int n, int nnz, const double *values, const int *row_idx, const int *col_idx,
                        int *csr_row_ptr, int *csr_col_idx, double *csr_values);
#ifdef __cplusplus
}
#endif
#endif
