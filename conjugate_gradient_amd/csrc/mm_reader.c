/* mm_reader.c -- MatrixMarket coordinate reader producing the diag / strict-upper COO split (include/mm_reader.h).
 * Behaviour follows the inline reader of reference tests/test_vec.c:443-560; the banner / size parsing that the
 * reference delegates to NIST mmio is written here from the format's definition (one banner line starting with
 * "%%MatrixMarket", '%' comment lines, then "rows cols entries"). */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mm_reader.h"

static void lower(char *s)
{
    for (; *s; ++s) *s = (char)tolower((unsigned char)*s);
}

void mm_free(void *p) { free(p); }

int mm_read_symmetric_split(const char *path, int *n, int *upper_count, double **diag, double **upper, int **rows, int **cols,
                            int *is_symmetric)
{
    *n = *upper_count = 0;
    *diag = *upper = NULL;
    *rows = *cols = NULL;
    FILE *f = fopen(path, "r");
    if (!f) {
        perror("fopen");
        return -1;
    }
    char line[1100], banner[64], object[64], format[64], field[64], symmetry[64];
    if (!fgets(line, sizeof line, f) ||
        sscanf(line, "%63s %63s %63s %63s %63s", banner, object, format, field, symmetry) != 5 ||
        strcmp(banner, "%%MatrixMarket") != 0) {
        fprintf(stderr, "Could not process Matrix Market banner.\n");
        fclose(f);
        return -2;
    }
    lower(object), lower(format), lower(field), lower(symmetry);
    const int pattern = strcmp(field, "pattern") == 0;
    const int sym = strcmp(symmetry, "symmetric") == 0;
    if (strcmp(object, "matrix") || strcmp(format, "coordinate") ||
        !(pattern || !strcmp(field, "real") || !strcmp(field, "integer")) || !(sym || !strcmp(symmetry, "general"))) {
        fprintf(stderr, "Only sparse real matrices supported.\n");
        fclose(f);
        return -2;
    }
    int M = 0, N = 0, nz = 0, have_size = 0;
    while (fgets(line, sizeof line, f)) {
        const char *p = line;
        while (*p && isspace((unsigned char)*p)) ++p;
        if (*p == '%' || *p == '\0') continue;
        have_size = sscanf(p, "%d %d %d", &M, &N, &nz) == 3;
        break;
    }
    if (!have_size || M < 0 || nz < 0) {
        fprintf(stderr, "Failed to read matrix size.\n");
        fclose(f);
        return -3;
    }
    if (M != N) {
        fprintf(stderr, "Matrix must be square (M=%d, N=%d)\n", M, N);
        fclose(f);
        return -3;
    }
    double *d = calloc((size_t)(N > 0 ? N : 1), sizeof *d);
    double *uv = malloc((size_t)(nz > 0 ? nz : 1) * sizeof *uv);
    int    *ur = malloc((size_t)(nz > 0 ? nz : 1) * sizeof *ur);
    int    *uc = malloc((size_t)(nz > 0 ? nz : 1) * sizeof *uc);
    int     rc = 0, count = 0;
    if (!d || !uv || !ur || !uc) {
        perror("malloc");
        rc = -5;
    }
    for (int k = 0; rc == 0 && k < nz; ++k) {
        int       ii = 0, jj = 0;
        double    vv = 1.0;
        const int got = pattern ? fscanf(f, "%d %d", &ii, &jj) : fscanf(f, "%d %d %lg", &ii, &jj, &vv);
        if (got != (pattern ? 2 : 3) || ii < 1 || jj < 1 || ii > N || jj > N) {
            fprintf(stderr, "Matrix Market entry %d is malformed or out of range\n", k + 1);
            rc = -4;
            break;
        }
        const int i = ii - 1, j = jj - 1;
        if (i == j) {
            d[i] = vv;
        } else if (i < j) {
            ur[count] = i, uc[count] = j, uv[count] = vv, ++count;
        } else if (sym) {
            ur[count] = j, uc[count] = i, uv[count] = vv, ++count; /* flip into the strict upper triangle */
        }
    }
    fclose(f);
    if (rc) {
        free(d), free(uv), free(ur), free(uc);
        return rc;
    }
    *n = N;
    *upper_count = count;
    *diag = d;
    *upper = uv;
    *rows = ur;
    *cols = uc;
    if (is_symmetric) *is_symmetric = sym;
    return 0;
}
