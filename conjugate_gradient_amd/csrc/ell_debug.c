/* ell_debug.c -- the reference's host-side ELL inspection helpers, so that its own tests link against libcg_hip
 * unchanged (reference tests/tests_cg.c:1214 calls analyze_ell_matrix_full_colmajor; the sibling tests call the
 * other two printers and the dense converters).
 *
 * Replaces reference src/ell.c:11-134 (print_dense_matrix, convert_ell_*_to_dense*) and :401-707 (analyze_ell_matrix*).
 * Plain host loops over host arrays -- nothing here touches the GPU or the hot path; what matters is that the
 * observable behaviour is the reference's: the same text on stdout (tests/test_ell_debug.py compares it byte for byte
 * with the reference compiled from its sources), the same dense matrices, the same return value.
 * "Padding" is judged by VALUE == 0.0 (as the reference does), the column id only feeds the consistency warning.
 */
#include <stdio.h>
#include <stdlib.h>

#include "../../include/ell.h"

void print_dense_matrix(double **A, int n)
{
    for (int r = 0; r < n; ++r) {
        for (int c = 0; c < n; ++c) printf("%6.1f ", A[r][c]);
        printf("\n");
    }
}

/* stored entry (row r, slot k) of a row-major / slot-major ELL array */
static size_t at_rowmajor(int r, int k, int n, int width) { (void)n; return (size_t)r * (size_t)width + (size_t)k; }
static size_t at_slotmajor(int r, int k, int n, int width) { (void)width; return (size_t)k * (size_t)n + (size_t)r; }

/* A[r][r] = diag[r]; every stored (r, col, v) is mirrored into A[r][col] and A[col][r] (last writer wins). */
static void ell_into_dense(int n, const double *diag, const double *vals, const int *cols, int width, double **A,
                           size_t (*at)(int, int, int, int), int clear_first, int slots_outer)
{
    if (clear_first)
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) A[r][c] = 0.0;
    if (clear_first) {
        for (int r = 0; r < n; ++r) A[r][r] = diag[r];
    }
    /* the traversal order decides which duplicate wins, so both orders of the reference are kept */
    const int outer = slots_outer ? width : n, inner = slots_outer ? n : width;
    for (int a = 0; a < outer; ++a) {
        if (!clear_first && !slots_outer) A[a][a] = diag[a];
        for (int b = 0; b < inner; ++b) {
            const int    r = slots_outer ? b : a, k = slots_outer ? a : b;
            const size_t q = at(r, k, n, width);
            if (cols[q] == -1) continue;
            A[r][cols[q]] = vals[q];
            A[cols[q]][r] = vals[q];
        }
    }
}

void convert_ell_to_dense(int n, double *diag, double *ell_values, int *ell_cols, int max_nnz_row, double **A)
{
    ell_into_dense(n, diag, ell_values, ell_cols, max_nnz_row, A, at_rowmajor, 0, 0);
}

void convert_ell_to_dense_colmajor(int n, double *diag, double *ell_values, int *ell_cols, int max_nnz_row, double **A)
{
    ell_into_dense(n, diag, ell_values, ell_cols, max_nnz_row, A, at_slotmajor, 0, 0);
}

void convert_ell_full_to_dense_rowmajor(int n, const double diag[], const double ell_values[], const int ell_cols[],
                                        int max_nnz_row, double **A)
{
    ell_into_dense(n, diag, ell_values, ell_cols, max_nnz_row, A, at_rowmajor, 1, 0);
}

void convert_ell_full_to_dense_colmajor(int n, const double diag[], const double ell_values[], const int ell_cols[],
                                        int max_nnz_row, double **A)
{
    ell_into_dense(n, diag, ell_values, ell_cols, max_nnz_row, A, at_slotmajor, 1, 1);
}

/* what the three printers count */
struct ell_census {
    int pads;       /* stored values equal to 0.0 */
    int mismatches; /* ... whose column id is not -1 */
};

static void census_line(const struct ell_census *c, int total_slots, int with_utilization, const char *what)
{
    printf("\nTotal slots: %d\n", total_slots);
    printf("Padding count: %d\n", c->pads);
    printf("Padding percentage: %.2f%%\n", 100.0 * c->pads / total_slots);
    if (with_utilization) printf("Memory utilization: %.2f%%\n", 100.0 * (total_slots - c->pads) / total_slots);
    if (c->mismatches == 0)
        printf("All zero values correctly marked with column index -1.\n");
    else
        printf("Warning: %d padding %s have col_idx != -1.\n", c->mismatches, what);
}

static int row_nonzeros(const double *vals, int r, int n, int width, size_t (*at)(int, int, int, int))
{
    int live = 0;
    for (int k = 0; k < width; ++k) live += vals[at(r, k, n, width)] != 0.0;
    return live;
}

/* row-major ELL: dumps the arrays in storage order, nnz_max per line */
void analyze_ell_matrix(int n, int nnz_max, double *ell_values, int *ell_col_idx)
{
    const int         total = n * nnz_max;
    struct ell_census c = {0, 0};
    printf("ELL matrix values (ell_values):\n");
    for (int q = 0; q < total; ++q) {
        if (q != 0 && q % nnz_max == 0) printf("\n");
        printf("%8.2f ", ell_values[q]);
        if (ell_values[q] == 0.0) {
            ++c.pads;
            if (ell_col_idx[q] != -1) ++c.mismatches;
        }
    }
    printf("\n");
    printf("\nELL column indices (ell_col_idx):\n");
    for (int q = 0; q < total; ++q) {
        if (q != 0 && q % nnz_max == 0) printf("\n");
        printf("%4d ", ell_col_idx[q]);
    }
    printf("\n");
    census_line(&c, total, 1, "values");

    int full_rows = 0, rows_with_3 = 0;
    for (int r = 0; r < n; ++r) {
        const int live = row_nonzeros(ell_values, r, n, nnz_max, at_rowmajor);
        full_rows += live == nnz_max;
        rows_with_3 += live == 3;
    }
    printf("\nRows without padding (completely full rows):\n");
    printf("Total full rows without padding: %d out of %d\n", full_rows, n);
    printf("Percentage of full rows without padding: %.2f%%\n", 100.0 * full_rows / n);
    printf("\nRows with exactly 3 non-zero values:\n");
    printf("Total rows with exactly 3 non-zero values: %d of %d\n", rows_with_3, n);
    printf("Percentage of rows with exactly 3 non-zero values: %.2f%%\n", 100.0 * rows_with_3 / n);
}

/* slot-major ELL: one line per row */
void analyze_ell_matrix_colmajor(int n, int nnz_max, const double *ell_values, const int *ell_col_idx)
{
    struct ell_census c = {0, 0};
    printf("ELL values (row \xC3\x97 slot):\n");
    for (int r = 0; r < n; ++r) {
        for (int k = 0; k < nnz_max; ++k) {
            const size_t q = at_slotmajor(r, k, n, nnz_max);
            printf("%8.2f ", ell_values[q]);
            if (ell_values[q] == 0.0) {
                ++c.pads;
                if (ell_col_idx[q] != -1) ++c.mismatches;
            }
        }
        printf("\n");
    }
    printf("\nELL column indices (row \xC3\x97 slot):\n");
    for (int r = 0; r < n; ++r) {
        for (int k = 0; k < nnz_max; ++k) printf("%4d ", ell_col_idx[at_slotmajor(r, k, n, nnz_max)]);
        printf("\n");
    }
    census_line(&c, n * nnz_max, 1, "entries");

    int by_count[5] = {0, 0, 0, 0, 0};
    for (int r = 0; r < n; ++r) {
        const int live = row_nonzeros(ell_values, r, n, nnz_max, at_slotmajor);
        if (live >= 1 && live <= 4) ++by_count[live];
    }
    printf("\nRows with exactly 1 non-zero: %d of %d (%.2f%%)\n", by_count[1], n, 100.0 * by_count[1] / n);
    for (int k = 2; k <= 4; ++k)
        printf("Rows with exactly %d non-zeros: %d of %d (%.2f%%)\n", k, by_count[k], n, 100.0 * by_count[k] / n);
}

/* slot-major full ELL: statistics only; returns the padded fraction of the slots */
double analyze_ell_matrix_full_colmajor(int n, int nnz_max, const double *ell_values, const int *ell_col_idx)
{
    struct ell_census c = {0, 0};
    for (int r = 0; r < n; ++r)
        for (int k = 0; k < nnz_max; ++k) {
            const size_t q = at_slotmajor(r, k, n, nnz_max);
            if (ell_values[q] == 0.0) {
                ++c.pads;
                if (ell_col_idx[q] != -1) ++c.mismatches;
            }
        }
    const int total = n * nnz_max;
    census_line(&c, total, 0, "entries");
    return (double)c.pads / total;
}
