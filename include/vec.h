/*
 * vec.h -- BLAS-1 helpers of the reference API (reference include/vec.h:26-46,
 * src/vec.c:39-126), host pointers, served by libcg_hip.  Reductions use the
 * reference's strict left-to-right order, so results are bit-identical.
 */
#ifndef CG_HIP_DROPIN_VEC_H
#define CG_HIP_DROPIN_VEC_H
#include <stdlib.h>
#include <string.h>
#ifdef __cplusplus
extern "C" {
#endif
void   vec_sub(double *a, double *b, double *out, int n);                   /* out = a - b      vec.c:39 */
void   vec_add(double *a, double *b, double *out, int n);                   /* out = a + b      vec.c:48 */
void   vec_assign(double *a, double *b, int n);                             /* a = b            vec.c:57 */
double vec_dot(double *a, double *b, int n);                                /* sum a[i]*b[i]    vec.c:67 */
void   vec_axpy(double *a, double *b, double alpha, double *out, int n);    /* out = a+alpha*b  vec.c:88 */
double vec_l1norm(double *a, int n);                                        /* sum |a[i]|       vec.c:77 */
double rmse(const double *a, const double *b, int n);                       /* vec.c:97  */
double euclidean_distance(const double *a, const double *b, int n);         /* vec.c:107 */
double max_difference(const double *a, const double *b, int n);             /* vec.c:117 */
#ifdef __cplusplus
}
#endif
#endif
