// partition.cpp -- row-block partition of a symmetric matrix for multi-GPU CG (host logic, no GPU needed).
//
// New module: the reference is serial.  Rank g owns the contiguous rows [lo_g, hi_g) with
// all their off-diagonals (both triangles).  Its share is a slot-major ELL whose slot order is the
// reference's (ell.c:815-855: the k-th upper entry lands in row i, then in row j, each row filling
// its slots in arrival order), i.e. exactly the rows [lo,hi) of the serial full ELL, so a rank's
// y[lo:hi) is bit-identical to the serial SpMV.  Columns are renumbered locally:
//     owned column c            -> c - lo
//     off-block column c        -> n_loc + (rank of c among the sorted distinct off-block columns)
// so the halo part of the p vector is [columns owned by lower ranks ... | ... by higher ranks], each
// neighbour's share contiguous -- a neighbour's message is received in place, with no unpack.
// By symmetry of the pattern, what rank q needs from this rank is {my rows that reference a column of q},
// so the send lists are derived locally, without communication.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/cg_hip.h"
#include "partition.h"

static int owner_of(const std::vector<int64_t> &starts, int64_t row)
{
    return (int)(std::upper_bound(starts.begin(), starts.end(), row) - starts.begin()) - 1;
}

extern "C" {

cgh_partition *cgh_partition_from_coo_host(int64_t n, int64_t U, const double *diag, const double *upper, const int *rows,
                                           const int *cols, int nranks, int rank, const int64_t *row_starts)
{
    if (n <= 0 || U < 0 || nranks < 1 || rank < 0 || rank >= nranks || !diag || (U > 0 && (!upper || !rows || !cols))) return nullptr;
    cgh_partition *P = new cgh_partition();
    P->n_global = n;
    P->nranks = nranks;
    P->rank = rank;
    P->starts.resize((size_t)nranks + 1);
    for (int g = 0; g <= nranks; ++g) {
        int64_t s = row_starts ? row_starts[g] : (n * g / nranks) & ~(int64_t)1;  // even block starts
        if (g == nranks) s = n;
        if (g == 0) s = 0;
        P->starts[(size_t)g] = s;
    }
    for (int g = 0; g < nranks; ++g)
        if (P->starts[(size_t)g] > P->starts[(size_t)g + 1]) {
            delete P;
            return nullptr;
        }
    const int64_t lo = P->starts[(size_t)rank], hi = P->starts[(size_t)rank + 1], nl = hi - lo;
    P->lo = lo;
    P->n_loc = nl;

    // pass 1: per-row placement counts of the local rows
    std::vector<int> cnt((size_t)nl, 0);
    for (int64_t k = 0; k < U; ++k) {
        const int i = rows[k], j = cols[k];
        if (j <= i || i < 0 || j >= n) {
            delete P;
            return nullptr;  // the reference's compute_max_nnz_row_full returns -1 here (ell.c:377)
        }
        if (i >= lo && i < hi) cnt[(size_t)(i - lo)]++;
        if (j >= lo && j < hi) cnt[(size_t)(j - lo)]++;
    }
    int K = 0;
    for (int64_t r = 0; r < nl; ++r) K = std::max(K, cnt[(size_t)r]);
    P->K = K;
    P->diag.assign(diag + lo, diag + hi);
    P->vals.assign((size_t)nl * K, 0.0);
    P->cols.assign((size_t)nl * K, -1);
    std::vector<int64_t> gcols((size_t)nl * K, -1);
    std::fill(cnt.begin(), cnt.end(), 0);

    // pass 2: place, in entry order, with global columns
    for (int64_t k = 0; k < U; ++k) {
        const int64_t i = rows[k], j = cols[k];
        if (i >= lo && i < hi) {
            const int64_t r = i - lo;
            const size_t  at = (size_t)cnt[(size_t)r]++ * (size_t)nl + (size_t)r;
            P->vals[at] = upper[k];
            gcols[at] = j;
        }
        if (j >= lo && j < hi) {
            const int64_t r = j - lo;
            const size_t  at = (size_t)cnt[(size_t)r]++ * (size_t)nl + (size_t)r;
            P->vals[at] = upper[k];
            gcols[at] = i;
        }
    }

    // halo = sorted distinct off-block columns
    std::vector<int64_t> halo;
    for (int64_t c : gcols)
        if (c >= 0 && (c < lo || c >= hi)) halo.push_back(c);
    std::sort(halo.begin(), halo.end());
    halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
    P->halo_global = halo;

    // local column ids, send sets, interior range
    std::vector<std::vector<int>> send((size_t)nranks);
    int64_t                       a = 0, b = nl;  // interior = [a, b): rows before a touch the lower halo, rows from b the upper
    P->nnz_off = 0;
    for (int s = 0; s < K; ++s)
        for (int64_t r = 0; r < nl; ++r) {
            const size_t  at = (size_t)s * (size_t)nl + (size_t)r;
            const int64_t c = gcols[at];
            if (c < 0) continue;
            P->nnz_off++;
            if (c >= lo && c < hi) {
                P->cols[at] = (int)(c - lo);
            } else {
                const int64_t h = std::lower_bound(halo.begin(), halo.end(), c) - halo.begin();
                P->cols[at] = (int)(nl + h);
                send[(size_t)owner_of(P->starts, c)].push_back((int)r);
                if (c < lo) a = std::max(a, r + 1);
                else b = std::min(b, r);
            }
        }
    if (a > b) a = b = 0;  // interleaved: no interior range, everything waits for the halo
    a = (a + 1) & ~(int64_t)1;  // vector kernels want even range starts
    if (b < nl) b &= ~(int64_t)1;
    if (a > b) a = b = 0;
    P->int_begin = a;
    P->int_end = b;

    for (int q = 0; q < nranks; ++q) {
        if (q == rank) continue;
        std::vector<int> &sl = send[(size_t)q];
        std::sort(sl.begin(), sl.end());
        sl.erase(std::unique(sl.begin(), sl.end()), sl.end());
        const int64_t qlo = P->starts[(size_t)q], qhi = P->starts[(size_t)q + 1];
        const int64_t r0 = std::lower_bound(halo.begin(), halo.end(), qlo) - halo.begin();
        const int64_t r1 = std::lower_bound(halo.begin(), halo.end(), qhi) - halo.begin();
        if (sl.empty() && r1 == r0) continue;
        cgh_neighbour nb;
        nb.rank = q;
        nb.send_idx = sl;
        nb.recv_off = r0;
        nb.recv_cnt = r1 - r0;
        P->nbs.push_back(nb);
    }
    return P;
}

void cgh_partition_free(cgh_partition *P) { delete P; }

int64_t cgh_partition_rows(const cgh_partition *P) { return P->n_loc; }
int64_t cgh_partition_first_row(const cgh_partition *P) { return P->lo; }
int64_t cgh_partition_halo(const cgh_partition *P) { return (int64_t)P->halo_global.size(); }
int     cgh_partition_width(const cgh_partition *P) { return P->K; }
int     cgh_partition_neighbours(const cgh_partition *P) { return (int)P->nbs.size(); }
void    cgh_partition_interior(const cgh_partition *P, int64_t *begin, int64_t *end)
{
    *begin = P->int_begin;
    *end = P->int_end;
}

int cgh_partition_neighbour(const cgh_partition *P, int k, int *rank, int64_t *send_count, int64_t *recv_off, int64_t *recv_count)
{
    if (k < 0 || k >= (int)P->nbs.size()) return CGH_EINVAL;
    const cgh_neighbour &nb = P->nbs[(size_t)k];
    *rank = nb.rank;
    *send_count = (int64_t)nb.send_idx.size();
    *recv_off = nb.recv_off;
    *recv_count = nb.recv_cnt;
    return CGH_OK;
}

int cgh_partition_copy_send_list(const cgh_partition *P, int k, int *out)
{
    if (k < 0 || k >= (int)P->nbs.size()) return CGH_EINVAL;
    const std::vector<int> &sl = P->nbs[(size_t)k].send_idx;
    if (!sl.empty()) memcpy(out, sl.data(), sl.size() * sizeof(int));
    return CGH_OK;
}

int cgh_partition_copy_local(const cgh_partition *P, double *diag, double *ell_vals, int *ell_cols, int64_t *halo_global)
{
    if (diag) memcpy(diag, P->diag.data(), P->diag.size() * sizeof(double));
    if (ell_vals && !P->vals.empty()) memcpy(ell_vals, P->vals.data(), P->vals.size() * sizeof(double));
    if (ell_cols && !P->cols.empty()) memcpy(ell_cols, P->cols.data(), P->cols.size() * sizeof(int));
    if (halo_global && !P->halo_global.empty()) memcpy(halo_global, P->halo_global.data(), P->halo_global.size() * sizeof(int64_t));
    return CGH_OK;
}

} // extern "C"
