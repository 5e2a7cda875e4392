/*
 * mm_reader.h -- MatrixMarket coordinate files as the reference's diag / strict-upper COO triple.
 *
 * The reference has no reader API of its own: it calls NIST's mmio (src/mmio.c there, not reproduced here) for the
 * banner and size lines and does the rest inline in tests/test_vec.c:443-560 -- 1-based (i, j, v) lines, diag[i] = v
 * for i == j (last one wins), every entry with i < j kept as an upper entry in FILE order, and, in a `symmetric`
 * file, every entry with i > j flipped to (j, i) in its place; in a `general` file the lower entries are dropped.
 * mm_read_symmetric_split does exactly that split, so the result feeds compute_max_nnz_row_full /
 * coo_to_ell_symmetric_full_colmajor / sorting_coo_to_csr (include/ell.h, csr.h) as in test_vec.c:572-581.
 *
 * Accepted: "matrix coordinate {real|integer|pattern} {general|symmetric}", square.  `pattern` entries get value 1.0.
 * Host-only C, part of libcg_hip.  Arrays are malloc'd; the caller frees them (free()).
 * Returns 0, or: -1 cannot open, -2 bad banner / unsupported kind, -3 bad size line or not square,
 * -4 malformed or out-of-range entry, -5 out of memory.  Messages go to stderr, as the reference's do.
 */
#ifndef CG_HIP_MM_READER_H
#define CG_HIP_MM_READER_H
#ifdef __cplusplus
extern "C" {
#endif
int mm_read_symmetric_split(const char *path, int *n, int *upper_count, double **diag, double **upper, int **rows, int **cols,
                            int *is_symmetric);
void mm_free(void *p);
#ifdef __cplusplus
}
#endif
#endif
