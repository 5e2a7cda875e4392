/* test_cg.c -- the reference's end-to-end check (tests/tests_cg.c:1061-1299, `make test_cg`) as a plain C
 * host program on top of libcg_hip: read an OpenFOAM LDU dump, build the slot-major full ELL, run CG on the
 * MI355X through the reference's own API (include/ell.h, cg.h, vec.h), compare with OpenFOAM's solution.
 *
 *   test_cg_hip <system-file> [<Phi-file>] [--format ell|coo|csr] [--max-iter N] [--tol T]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/cg.h"
#include "../../include/csr.h"
#include "../../include/ell.h"
#include "../../include/parser.h"
#include "../../include/vec.h"

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv)
{
    const char *sys_path = NULL, *phi_path = NULL, *format = "ell";
    int         max_iter = 1000;
    double      tol = 1e-6;
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "--format") && a + 1 < argc) format = argv[++a];
        else if (!strcmp(argv[a], "--max-iter") && a + 1 < argc) max_iter = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--tol") && a + 1 < argc) tol = atof(argv[++a]);
        else if (!sys_path) sys_path = argv[a];
        else if (!phi_path) phi_path = argv[a];
    }
    if (!sys_path) {
        fprintf(stderr, "usage: %s <system-file> [<Phi-file>] [--format ell|coo|csr] [--max-iter N] [--tol T]\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(sys_path, "r");
    if (!f) {
        perror("Error in opening file");
        return 1;
    }
    printf("File %s opened successfully.\n", sys_path);
    int     n = 0, n_upper = 0, n_lower = 0, n_phi = 0;
    double *source = parseDoubleArray(f, "source", &n);
    double *diag = parseDoubleArray(f, "diag", &n);
    double *upper = parseDoubleArray(f, "upper", &n_upper);
    int    *upperAddr = parseIntArray(f, "upperAddr", &n_upper);
    int    *lowerAddr = parseIntArray(f, "lowerAddr", &n_lower);
    fclose(f);
    if (!source || !diag || !upper || !upperAddr || !lowerAddr || n_upper != n_lower) {
        fprintf(stderr, "Error: could not read the linear system from %s\n", sys_path);
        return 1;
    }
    double *phi = NULL;
    if (phi_path) {
        FILE *g = fopen(phi_path, "r");
        if (!g) {
            perror("Error in opening file");
            return 1;
        }
        phi = parseDoubleArraySolution(g, "internalField   nonuniform List<scalar>", &n_phi);
        fclose(g);
        if (!phi || n_phi != n) {
            printf("Error: The number of elements in the solution vector does not match the number of elements in the source vector.\n");
            return 1;
        }
    }
    printf("n = %d, upper = %d\n", n, n_upper);

    double *y = malloc((size_t)n * sizeof *y);
    int     iterations = -1;
    double  t0 = 0.0, t1 = 0.0;
    if (!strcmp(format, "ell")) {
        /* row = lowerAddr, col = upperAddr (tests_cg.c:1195,1207) */
        const int nnz_max = compute_max_nnz_row_full(n, n_upper, lowerAddr, upperAddr);
        printf("nnz_max = %d\n", nnz_max);
        if (nnz_max < 0) return 1;
        double *ell_values = malloc((size_t)nnz_max * n * sizeof *ell_values);
        int    *ell_cols = malloc((size_t)nnz_max * n * sizeof *ell_cols);
        coo_to_ell_symmetric_full_colmajor(n, n_upper, upper, lowerAddr, upperAddr, ell_values, ell_cols, nnz_max);
        long pads = 0;
        for (long k = 0; k < (long)nnz_max * n; ++k) pads += ell_cols[k] == -1;
        printf("ELL slots = %ld, padding = %ld (%.4f)\n", (long)nnz_max * n, pads, (double)pads / ((double)nnz_max * n));
        t0 = now_s();
        iterations = conjugate_gradient_ell_full_colmajor(n, diag, n_upper, ell_values, ell_cols, nnz_max, source, y, max_iter, tol);
        t1 = now_s();
        free(ell_values);
        free(ell_cols);
    } else if (!strcmp(format, "coo")) {
        t0 = now_s();
        iterations = conjugate_gradient_coo(n, diag, n_upper, upper, lowerAddr, upperAddr, source, y, max_iter, tol);
        t1 = now_s();
    } else {
        /* the call pattern of the reference's src/main.c:234-262: row_ptr from the row-sorted COO, values/cols reused */
        int *row_ptr = malloc(((size_t)n + 1) * sizeof *row_ptr);
        coo_to_csr(n, n_upper, upper, lowerAddr, upperAddr, row_ptr);
        t0 = now_s();
        iterations = conjugate_gradient_csr(n, diag, n_upper, upper, row_ptr, upperAddr, source, y, max_iter, tol);
        t1 = now_s();
        free(row_ptr);
    }
    printf("Time spent: %f seconds\n", t1 - t0);
    if (iterations >= 0) {
        printf("Converged in %d iterations.\n\n", iterations);
        if (phi) {
            for (int i = 0; i < 5 && i < n; i++)
                printf("x[%d] = %.10f | OpenFoam: x[%d] = %.10f | difference: %f\n", i, y[i], i, phi[i], y[i] - phi[i]);
            printf("RMSE: %f\n", rmse(y, phi, n));
            printf("Euclidean distance: %f\n", euclidean_distance(y, phi, n));
            printf("Max component difference: %f\n", max_difference(y, phi, n));
        }
    } else {
        printf("Didnt reach convergences init max_iter.\n");
    }
    free(y);
    free(source);
    free(diag);
    free(upper);
    free(upperAddr);
    free(lowerAddr);
    free(phi);
    return 0;
}
