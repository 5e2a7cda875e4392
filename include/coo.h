/*
 * coo.h -- COO (OpenFOAM LDU triple) SpMV of the reference API (reference
 * include/coo.h:23-30, src/coo.c:202-233), host pointers, served by libcg_hip.
 * The reference's rand()-based generators (coo.h:4-21) are not part of the hot
 * path and are not provided; seeded generators live in cg_hip.h.
 */
#ifndef CG_HIP_DROPIN_COO_H
#define CG_HIP_DROPIN_COO_H
#include <stdio.h>
#include <stdlib.h>
#ifdef __cplusplus
extern "C" {
#endif
/* out = A v with A = diag + strictly-upper (value, rows, columns) mirrored; bit-identical to the reference */
void mv_coo_symmetric(int n, int coo_length, const double *diag, const double *value,
                      const int *rows, const int *columns, const double *v, double *out);
#ifdef __cplusplus
}
#endif
#endif
