/*
 * parser.h -- OpenFOAM LDU-dump readers of the reference API (reference include/parser.h:6-8,
 * src/parser.c:29-177).  Host-only C; part of libcg_hip so that a driver links one library.
 * Each call scans forward from the stream's current position for the keyword, so sections must be
 * requested in file order (source, diag, upper, upperAddr, lowerAddr), as the reference's drivers do
 * (tests/tests_cg.c:1095-1099).  The caller frees the returned array.  NULL on any error.
 */
#ifndef CG_HIP_DROPIN_PARSER_H
#define CG_HIP_DROPIN_PARSER_H
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif
double *parseDoubleArray(FILE *file, const char *target_word, int *out_count);
int    *parseIntArray(FILE *file, const char *target_word, int *out_count);
/* volScalarField reader: line containing target_word, then the count line, "(", one value per line, ")" */
double *parseDoubleArraySolution(FILE *file, const char *target_word, int *out_count);
#ifdef __cplusplus
}
#endif
#endif
