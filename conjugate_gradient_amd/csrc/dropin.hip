// dropin.hip -- the reference's own C API (include/cg.h, csr.h, ell.h, vec.h, coo.h), host pointers in,
// host pointers out, served by the HIP path:  upload -> cgh_* -> download.
//
// Signatures, return values, stdout text and early-return behaviour follow the reference
// (file:line cited per function, relative to /root/reference).  There is no CPU fallback: if no
// MI355X is usable the solvers print the error to stderr and return -1, the void functions return
// without touching their outputs.
//
// Environment: CG_HIP_DEVICE (default 0), CG_HIP_QUIET=1 (suppress the reference's printf chatter),
// CG_HIP_MODE=strict|fast (reduction order of the CG dots: strict = the reference's left-to-right sums,
// bit-identical trajectory, the default at every size -- one lane below 4096 rows, the order-exact parallel
// sums of xsum.hip above; fast = tree reductions, ~2.4x faster per iteration at 16.7M rows).
#include <cmath>
#include <new>
#include <vector>

#include "cgh_internal.h"

int cgh_convert_coo_to_ell_device(int64_t n, int64_t U, const double *d_vals, const int *d_rows, const int *d_cols, int K,
                                  bool both, bool row_major, bool wide_cols, double *d_ev, void *d_ec,
                                  unsigned long long *stop_seq, hipStream_t s);
int cgh_max_nnz_row_device(int64_t n, int64_t U, const int *d_rows, const int *d_cols, bool both, int64_t *bad_entry, hipStream_t s);
int cgh_sort_coo_to_csr_device(int64_t n, int64_t nnz, const double *d_vals, const int *d_rows, const int *d_cols,
                               int *d_row_ptr32, int *d_csr_cols, double *d_csr_vals, hipStream_t s);
int cgh_coo_row_ptr_device(int64_t n, int64_t U, const int *d_rows, int *d_row_ptr, hipStream_t s);

namespace {

constexpr unsigned long long NO_STOP = ~0ull;

bool quiet() { static const bool q = getenv("CG_HIP_QUIET") && atoi(getenv("CG_HIP_QUIET")) != 0; return q; }

bool ensure_ready()
{
    if (g_cgh.ready) return true;
    const int dev = getenv("CG_HIP_DEVICE") ? atoi(getenv("CG_HIP_DEVICE")) : 0;
    return cgh_init(dev) == CGH_OK;
}

int strict_for(int64_t n)
{
    const char *m = getenv("CG_HIP_MODE");
    (void)n;
    if (m && !strcmp(m, "fast")) return 0;
    return 1;
}

template <typename T>
struct dbuf {
    T     *p = nullptr;
    size_t count = 0;
    explicit dbuf(size_t c) : count(c) { p = (T *)cgh_alloc(sizeof(T) * (c ? c : 1)); }
    dbuf(const T *host, size_t c) : count(c)
    {
        p = (T *)cgh_alloc(sizeof(T) * (c ? c : 1));
        if (p && c && cgh_h2d(p, host, sizeof(T) * c) != CGH_OK) {
            cgh_free(p);
            p = nullptr;
        }
    }
    ~dbuf() { cgh_free(p); }
    dbuf(const dbuf &) = delete;
    dbuf &operator=(const dbuf &) = delete;
    bool ok() const { return p != nullptr; }
    int  get(T *host) const { return cgh_d2h(host, p, sizeof(T) * count); }
};

struct mat_guard {
    cgh_matrix *m;
    ~mat_guard() { cgh_matrix_free(m); }
};

// The entry list of the reference's COO SpMV / solver (host arrays): ANY (i, j) inside the matrix, as coo.c:202-233 there
// serves them -- each entry scatters to y[i] and y[j], an (i, i) entry therefore twice.
cgh_matrix *coo_any_from_host(int n, int count, const double *diag, const double *value, const int *rows, const int *cols, int layout)
{
    if (n <= 0 || count < 0 || !diag || (count > 0 && (!value || !rows || !cols))) return nullptr;
    dbuf<double> dd(diag, (size_t)n), dv(value, (size_t)count);
    dbuf<int>    dr(rows, (size_t)count), dc(cols, (size_t)count);
    if (!dd.ok() || !dv.ok() || !dr.ok() || !dc.ok()) return nullptr;
    return cgh_matrix_from_entries_device(n, count, dd.p, dv.p, dr.p, dc.p, layout, true);
}

// rows[] of a CSR-compressed upper triangle: rows[e] = r for row_ptr[r] <= e < row_ptr[r+1]
__global__ __launch_bounds__(256) void expand_row_ptr(int64_t n, const int *__restrict__ row_ptr, int *__restrict__ rows)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256)
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) rows[e] = (int)r;
}

// Compact the non-pad entries of an upper ELL (either orientation), in MEMORY order, into a COO list.
// Memory order is the order in which the reference's SpMV visits them (slot-major: ell.c:971-998,
// row-major: ell.c:182-210), and each visit adds into y[row] and then y[col] -- the same two placements
// per entry, in the same order, as the COO conversion.  flags/pos: 1 per non-pad entry and its exclusive scan.
// (entries at memory position >= limit count as padding: the reference stops there on a malformed entry)
__global__ __launch_bounds__(256) void ell_flags(int64_t total, int64_t limit, const int *__restrict__ ec, int *__restrict__ flag)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256)
        flag[k] = ec[k] != -1 && k < limit;
}
__global__ __launch_bounds__(256) void ell_compact(int64_t total, int64_t limit, int64_t n, int K, int row_major,
                                                  const double *__restrict__ ev, const int *__restrict__ ec,
                                                  const int64_t *__restrict__ pos, double *__restrict__ v, int *__restrict__ rows,
                                                  int *__restrict__ cols)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        if (ec[k] == -1 || k >= limit) continue;
        const int64_t at = pos[k];
        v[at] = ev[k];
        cols[at] = ec[k];
        rows[at] = (int)(row_major ? k / K : k % n);
    }
}

// row-major upper ELL of width K -> width K+1 with the diagonal as slot 0 of every row (see matrix_from_upper_ell)
__global__ __launch_bounds__(256) void ell_extend_rowmajor(int64_t n, int K, const double *__restrict__ diag,
                                                          const double *__restrict__ ev, const int *__restrict__ ec,
                                                          double *__restrict__ xv, int *__restrict__ xc)
{
    const int64_t total = n * (int64_t)(K + 1);
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t i = k / (K + 1);
        const int     s = (int)(k % (K + 1));
        xv[k] = s == 0 ? diag[i] : ev[i * K + s - 1];
        xc[k] = s == 0 ? (int)i : ec[i * K + s - 1];
    }
}

__global__ __launch_bounds__(256) void narrow_cols(int64_t total, const uint64_t *__restrict__ in, int *__restrict__ out)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) out[k] = (int)in[k];
}

// generic row-major ELL (pad = any negative column) -> slot-major with pad = -1; a row's slot order is kept
__global__ __launch_bounds__(256) void ell_rowmajor_to_slotmajor(int64_t n, int K, const double *__restrict__ ev, const int *__restrict__ ec,
                                                                double *__restrict__ sv, int *__restrict__ sc)
{
    const int64_t total = n * (int64_t)K;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t i = k % n;           // k walks the slot-major output: coalesced stores, strided loads
        const int     s = (int)(k / n);
        const int     c = ec[i * K + s];
        sv[k] = ev[i * K + s];
        sc[k] = c >= 0 ? c : -1;
    }
}

// an ELL matrix whose arrays are allocated but not filled (the callers fill them on the device)
cgh_matrix *ell_shell(int n, int K, int64_t nnz_off)
{
    cgh_matrix *m = new (std::nothrow) cgh_matrix();
    if (!m) return nullptr;
    m->n = n;
    m->K = K;
    m->layout = CGH_LAYOUT_ELL;
    m->stored = (int64_t)n * K;
    m->nnz_off = nnz_off;
    m->diag = (double *)cgh_alloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    m->vals = (double *)cgh_alloc(sizeof(double) * (size_t)(m->stored > 0 ? m->stored : 1));
    m->cols = (int *)cgh_alloc(sizeof(int) * (size_t)(m->stored > 0 ? m->stored : 1));
    if (!m->diag || !m->vals || !m->cols) {
        cgh_matrix_free(m);
        return nullptr;
    }
    return m;
}

__global__ __launch_bounds__(256) void sub_kernel(int64_t n, const double *a, const double *b, double *out, int add)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = add ? a[i] + b[i] : a[i] - b[i];
}

__global__ __launch_bounds__(256) void absdiff_max(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                  unsigned long long *out)
{
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double d = fabs(a[i] - b[i]);
        if (d > m) m = d;  // NaN never wins, as in the reference's `if (diff > max_diff)`
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(m, off, 64);
        if (o > m) m = o;
    }
    // non-negative doubles order like their bit patterns
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

inline int grid_for(int64_t work) { return cgh_stream_grid((work + 255) / 256, 8); }

int cgh_exclusive_scan_i32_to_i64_fwd(const int *d_in, int64_t *d_out, int64_t count, hipStream_t s);

// The reference's per-solver chatter (cg.c:426-451, :478, :496-501).
void print_prologue(const char *banner, const cgh_cg_report &rep, int64_t n)
{
    if (quiet()) return;
    printf("%s", banner);
    printf("r[0...4] = \n");
    for (int i = 0; i < 5; i++) printf("%f ", i < n ? rep.r0[i] : 0.0);  // (the reference reads 5 entries whatever n is)
    printf("\n");
    printf("Initial residual norm: %f\n", rep.initial_res);
}

// The solver's chatter comes out of the history as the solve hands it over, window by window: nothing here grows with
// max_iter (the reference's loop keeps no history at all).
struct chatter {
    const char *banner;
    int64_t     n;
    bool        prologue_done = false;
};

void print_window(void *ctx, int first, int count, const double *e, int converged, const cgh_cg_report *rep)
{
    chatter *c = (chatter *)ctx;
    if (quiet()) return;
    if (!c->prologue_done) {
        print_prologue(c->banner, *rep, c->n);
        c->prologue_done = true;
    }
    for (int k = 0; k < count; ++k) {
        const int it = first + k;
        if (it == 0) continue;                         // entry 0 is the initial residual (printed by the prologue)
        if (converged > 0 && it == converged) break;   // the converging iteration prints the final line instead
        printf("Iteration %d: Residual norm = %g\n", it, e[(size_t)4 * k]);
    }
}

int run_cg(const char *banner, const cgh_matrix *m, const double *b, double *x, int max_iter, double tol)
{
    try {
        const int64_t n = m->n;
        dbuf<double>  db(b, (size_t)n), dx((size_t)n);
        if (!db.ok() || !dx.ok()) return -1;
        cgh_cg_opts opts;
        memset(&opts, 0, sizeof opts);
        // the reference still zeroes x, computes r0 and prints when max_iter < 1; with no iterations it reports non-convergence
        opts.max_iter = max_iter < 1 ? 1 : max_iter;
        opts.tol = tol;
        opts.strict = strict_for(n);
        opts.check_every = 32;
        cgh_cg_report rep;
        chatter       c{banner, n};
        if (cgh_cg_solve_sink(m, db.p, dx.p, &opts, &rep, print_window, &c) != CGH_OK) return -1;
        if (dx.get(x) != CGH_OK) return -1;
        if (!quiet()) {
            if (!c.prologue_done) print_prologue(banner, rep, n);
            if (rep.ret > 0) {
                printf("\nResidual norm: %.g\n", rep.final_res);
            } else {
                printf("Maximum iterations reached without convergence.\n");
                printf("Residual norm: %g\n", rep.final_res);
            }
        }
        return rep.ret;
    } catch (const std::exception &e) {  // nothing may unwind through the extern "C" callers
        cgh_set_error("conjugate_gradient_*: %s", e.what());
        return -1;
    } catch (...) {
        cgh_set_error("conjugate_gradient_*: unknown failure");
        return -1;
    }
}

// stop: memory position (in the caller's ELL arrays) of the first malformed entry, -1 if none: everything from there on is
// left out, which is exactly what the reference's loops have accumulated when they return early.
cgh_matrix *matrix_from_upper_ell(int n, int K, const double *diag, const double *ev, const int *ec, bool row_major, int layout,
                                  int64_t stop = -1)
{
    // upper ELL -> COO in visiting order -> full layout (see ell_compact).  The row-major SpMV of the reference
    // (ell.c:182-210) adds diag[i]*x[i] when it reaches row i, i.e. AFTER the contributions scattered into y[i]
    // by earlier rows: there the diagonal travels inside the entry list, as slot 0 of its row.
    hipStream_t  s = g_cgh.stream;
    const int    KK = row_major ? K + 1 : K;
    const int64_t total = (int64_t)n * KK;
    // position of the stop entry in the (possibly diagonal-extended) list that is compacted below
    const int64_t limit = stop < 0 ? total : (row_major ? (stop / K) * KK + stop % K + 1 : stop);
    dbuf<double> dd(diag, (size_t)n), dv0(ev, (size_t)n * K), dv((size_t)(row_major ? total : 0));
    dbuf<int>    dc0(ec, (size_t)n * K), dc((size_t)(row_major ? total : 0)), flag((size_t)total + 1);
    dbuf<int64_t> pos((size_t)total + 2);
    if (!dd.ok() || !dv0.ok() || !dv.ok() || !dc0.ok() || !dc.ok() || !flag.ok() || !pos.ok()) return nullptr;
    const double *src_v = dv0.p;
    const int    *src_c = dc0.p;
    if (row_major) {
        hipLaunchKernelGGL(ell_extend_rowmajor, dim3(grid_for(total)), dim3(256), 0, s, (int64_t)n, K, (const double *)dd.p,
                           (const double *)dv0.p, (const int *)dc0.p, dv.p, dc.p);
        src_v = dv.p;
        src_c = dc.p;
    }
    if (hipMemsetAsync(flag.p + total, 0, sizeof(int), s) != hipSuccess) return nullptr;
    if (total > 0) hipLaunchKernelGGL(ell_flags, dim3(grid_for(total)), dim3(256), 0, s, total, limit, src_c, flag.p);
    if (cgh_exclusive_scan_i32_to_i64_fwd(flag.p, pos.p, total + 1, s) != CGH_OK) return nullptr;
    int64_t U = 0;
    if (cgh_d2h(&U, pos.p + total, sizeof(int64_t)) != CGH_OK) return nullptr;
    dbuf<double> cv((size_t)U);
    dbuf<int>    cr((size_t)U), cc((size_t)U);
    if (!cv.ok() || !cr.ok() || !cc.ok()) return nullptr;
    if (total > 0)
        hipLaunchKernelGGL(ell_compact, dim3(grid_for(total)), dim3(256), 0, s, total, limit, (int64_t)n, KK, row_major ? 1 : 0,
                           src_v, src_c, (const int64_t *)pos.p, cv.p, cr.p, cc.p);
    if (hipGetLastError() != hipSuccess) return nullptr;
    return cgh_matrix_from_entries_device(n, U, row_major ? nullptr : dd.p, cv.p, cr.p, cc.p, layout);
}

int spmv_host(const cgh_matrix *m, const double *x, double *y)
{
    dbuf<double> dx(x, (size_t)m->n), dy((size_t)m->n);
    if (!dx.ok() || !dy.ok()) return -1;
    // ONE product with a matrix that is dropped right after it: no derived copies (column dictionary, tile-blocked values,
    // staging plan -- several passes over the matrix and, for wide rows, its values a second time in HBM) are built for it
    cgh_spmv_opts so;
    so.no_alloc = true;
    if (cgh_launch_spmv(m, dx.p, dy.p, nullptr, nullptr, nullptr, g_cgh.stream, &so) != CGH_OK) return -1;
    return dy.get(y);
}

} // namespace

int cgh_exclusive_scan_i32_to_i64(const int *d_in, int64_t *d_out, int64_t count, hipStream_t s);
namespace {
int cgh_exclusive_scan_i32_to_i64_fwd(const int *d_in, int64_t *d_out, int64_t count, hipStream_t s)
{
    return cgh_exclusive_scan_i32_to_i64(d_in, d_out, count, s);
}
} // namespace

extern "C" {

// ------------------------------------------------------------------------------- include/cg.h
// reference src/cg.c:414-508
int conjugate_gradient_ell_full_colmajor(const int n, double *diag, int upper_count, double *ell_values, int *ell_col,
                                         int max_nnz_row, double *b, double *x, int max_iter, double tol)
{
    (void)upper_count;  // unused by the reference too
    if (!ensure_ready()) return -1;
    mat_guard g{cgh_matrix_from_ell_host(n, max_nnz_row, diag, ell_values, ell_col)};
    if (!g.m) return -1;
    return run_cg("CG (ELL) \n\n", g.m, b, x, max_iter, tol);
}

// reference src/cg.c:21-118
int conjugate_gradient_coo(const int n, double *diag, int coo_length, double *upper, int *rows, int *cols, double *b,
                           double *x, int max_iter, double tol)
{
    if (!ensure_ready()) return -1;
    mat_guard g{coo_any_from_host(n, coo_length, diag, upper, rows, cols, CGH_LAYOUT_ELL)};
    if (!g.m) return -1;
    return run_cg("CG (COO) \n\n", g.m, b, x, max_iter, tol);
}

// reference src/cg.c:121-222: (upper, rows_ptr, cols) is the upper triangle in CSR
int conjugate_gradient_csr(const int n, double *diag, int upper_count, double *upper, int *rows_ptr, int *cols, double *b,
                           double *x, int max_iter, double tol)
{
    if (!ensure_ready()) return -1;
    if (n <= 0) return -1;
    const int    U = rows_ptr[n];
    (void)upper_count;
    dbuf<double> dd(diag, (size_t)n), dv(upper, (size_t)U);
    dbuf<int>    drp(rows_ptr, (size_t)n + 1), dc(cols, (size_t)U), dr((size_t)U);
    if (!dd.ok() || !dv.ok() || !drp.ok() || !dc.ok() || !dr.ok()) return -1;
    hipLaunchKernelGGL(expand_row_ptr, dim3(grid_for(n)), dim3(256), 0, g_cgh.stream, (int64_t)n, (const int *)drp.p, dr.p);
    mat_guard g{cgh_matrix_from_entries_device(n, U, dd.p, dv.p, dr.p, dc.p, CGH_LAYOUT_CSR, true)};
    if (!g.m) return -1;
    g.m->exact_rows = true;  // the reference's order in every row, whatever its length
    return run_cg("CG (CSR) \n\n", g.m, b, x, max_iter, tol);
}

// reference src/cg.c:319-411: slot-major upper-only ELL
int conjugate_gradient_ell_colmajor(const int n, double *diag, int upper_count, double *ell_values, int *ell_col,
                                    int max_nnz_row, double *b, double *x, int max_iter, double tol)
{
    (void)upper_count;
    if (!ensure_ready()) return -1;
    mat_guard g{matrix_from_upper_ell(n, max_nnz_row, diag, ell_values, ell_col, false, CGH_LAYOUT_ELL)};
    if (!g.m) return -1;
    return run_cg("CG (ELL) \n\n", g.m, b, x, max_iter, tol);
}

// reference src/cg.c:225-316: row-major upper-only ELL
int conjugate_gradient_ell(const int n, double *diag, int upper_count, double *ell_values, int *ell_col, int max_nnz_row,
                           double *b, double *x, int max_iter, double tol)
{
    (void)upper_count;
    if (!ensure_ready()) return -1;
    mat_guard g{matrix_from_upper_ell(n, max_nnz_row, diag, ell_values, ell_col, true, CGH_LAYOUT_ELL)};
    if (!g.m) return -1;
    return run_cg("CG (ELL) \n\n", g.m, b, x, max_iter, tol);
}

// ------------------------------------------------------------------------------- include/ell.h
// reference src/ell.c:1004-1040
void mv_ell_symmetric_full_colmajor(int n, int max_nnz_row, double *diag, double *ell_values, int *ell_cols, double *x, double *y)
{
    if (!ensure_ready()) return;
    mat_guard g{cgh_matrix_from_ell_host(n, max_nnz_row, diag, ell_values, ell_cols)};
    if (g.m) spmv_host(g.m, x, y);
}

// reference src/ell.c:1045-1081 (uint64_t columns; the reference narrows them to int before use): the wide ids are
// uploaded once and narrowed on the device straight into the matrix
void mv_ell_symmetric_full_colmajor_sdtint(int n, int max_nnz_row, double *diag, double *ell_values, uint64_t *ell_cols,
                                           double *x, double *y)
{
    if (!ensure_ready() || n <= 0) return;
    const int64_t total = (int64_t)n * max_nnz_row;
    int64_t       live = 0;
    for (int64_t k = 0; k < total; ++k) live += (int)ell_cols[k] != -1;
    dbuf<uint64_t> wide(ell_cols, (size_t)total);
    mat_guard      g{ell_shell(n, max_nnz_row, live)};
    if (!wide.ok() || !g.m) return;
    if (cgh_h2d(g.m->diag, diag, sizeof(double) * (size_t)n) != CGH_OK || cgh_h2d(g.m->vals, ell_values, sizeof(double) * (size_t)total) != CGH_OK)
        return;
    if (total > 0)
        hipLaunchKernelGGL(narrow_cols, dim3(grid_for(total)), dim3(256), 0, g_cgh.stream, total, (const uint64_t *)wide.p, g.m->cols);
    if (hipGetLastError() != hipSuccess) return;
    spmv_host(g.m, x, y);
}

// reference src/ell.c:946-999.  A stored column <= its row makes the reference print and return with y partially
// accumulated: the message and that partial y are reproduced (the expansion stops at the offending entry).
void mv_ell_symmetric_upper_colmajor(int n, int max_nnz_row, double *diag, double *ell_values, int *ell_cols, double *x, double *y)
{
    if (!ensure_ready()) return;
    // a stored column <= its row: the reference prints this and returns with y = diag*x + the entries visited so far
    int64_t stop = -1;
    for (int s = 0; s < max_nnz_row && stop < 0; ++s)
        for (int i = 0; i < n; ++i) {
            const int c = ell_cols[(size_t)s * n + i];
            if (c != -1 && c <= i) {
                printf("Error: non-upper element at row=%d, col=%d\n", i, c);
                stop = (int64_t)s * n + i;
                break;
            }
        }
    mat_guard g{matrix_from_upper_ell(n, max_nnz_row, diag, ell_values, ell_cols, false, CGH_LAYOUT_ELL, stop)};
    if (g.m) spmv_host(g.m, x, y);
}

// reference src/ell.c:168-211.  As for the slot-major variant, a stored column <= its row is reported and y keeps what
// the reference has accumulated by then: rows before the offending one complete (with their diagonal), that row up
// to the entry, later rows only what earlier rows scattered into them.
void mv_ell_symmetric_upper(int n, int max_nnz_row, double *diag, double *ell_values, int *ell_cols, double *x, double *y)
{
    if (!ensure_ready()) return;
    int64_t stop = -1;
    for (int i = 0; i < n && stop < 0; ++i)
        for (int s = 0; s < max_nnz_row; ++s) {
            const int c = ell_cols[(size_t)i * max_nnz_row + s];
            if (c != -1 && c <= i) {
                printf("Error: Found a non-upper triangular element in ELL format: row=%d, col=%d\n", i, c);
                stop = (int64_t)i * max_nnz_row + s;
                break;
            }
        }
    mat_guard g{matrix_from_upper_ell(n, max_nnz_row, diag, ell_values, ell_cols, true, CGH_LAYOUT_ELL, stop)};
    if (g.m) spmv_host(g.m, x, y);
}

// reference src/ell.c:213-251: the same product without the upper check; identical on valid (upper-only) input,
// which is the only input served here (anything else is reported as above).
void mv_ell_symmetric_upper_opt(int n, int max_nnz_row, double *diag, double *ell_values, int *ell_cols, double *x, double *y)
{
    mv_ell_symmetric_upper(n, max_nnz_row, diag, ell_values, ell_cols, x, y);
}

// reference src/ell.c:136-159: generic row-major ELL (no separate diagonal, pad = negative column): uploaded as it is,
// turned slot-major on the device (zero diagonal: 0.0 + 0.0*x starts each row's sum); per-row order (slots 0..K-1) is kept
void mv_ell(int n, int max_nnz_row, double *ell_values, int *ell_cols, double *x, double *y)
{
    if (!ensure_ready() || n <= 0) return;
    const int64_t total = (int64_t)n * max_nnz_row;
    int64_t       live = 0;
    for (int64_t k = 0; k < total; ++k) live += ell_cols[k] >= 0;
    dbuf<double> rv(ell_values, (size_t)total);
    dbuf<int>    rc(ell_cols, (size_t)total);
    mat_guard    g{ell_shell(n, max_nnz_row, live)};
    if (!rv.ok() || !rc.ok() || !g.m) return;
    if (hipMemsetAsync(g.m->diag, 0, sizeof(double) * (size_t)n, g_cgh.stream) != hipSuccess) return;
    if (total > 0)
        hipLaunchKernelGGL(ell_rowmajor_to_slotmajor, dim3(grid_for(total)), dim3(256), 0, g_cgh.stream, (int64_t)n, max_nnz_row,
                           (const double *)rv.p, (const int *)rc.p, g.m->vals, g.m->cols);
    if (hipGetLastError() != hipSuccess) return;
    spmv_host(g.m, x, y);
}

static int max_nnz_row(int n, int upper_nnz, const int *rows, const int *cols, bool both, bool to_stderr)
{
    if (!ensure_ready()) return -1;
    dbuf<int> dr(rows, (size_t)upper_nnz), dc(cols, (size_t)upper_nnz);
    if (!dr.ok() || !dc.ok()) return -1;
    int64_t   bad = -1;
    const int k = cgh_max_nnz_row_device(n, upper_nnz, dr.p, dc.p, both, &bad, g_cgh.stream);
    if (k < 0 && bad >= 0) {
        if (to_stderr)
            fprintf(stderr, "Error: non-upper entry row=%d, col=%d\n", rows[bad], cols[bad]);  // ell.c:379
        else
            printf("Error: Found a non-upper triangular element in COO format: row=%d, col=%d\n", rows[bad], cols[bad]);  // ell.c:337
        return -1;
    }
    return k;
}

// reference src/ell.c:315-352
int compute_max_nnz_row_upper(int n, int upper_nnz, int *coo_rows, int *coo_cols)
{
    return max_nnz_row(n, upper_nnz, coo_rows, coo_cols, false, false);
}

// reference src/ell.c:360-396
int compute_max_nnz_row_full(int n, int upper_nnz, const int *coo_rows, const int *coo_cols)
{
    return max_nnz_row(n, upper_nnz, coo_rows, coo_cols, true, true);
}

static void coo_to_ell(int n, int upper_nnz, const double *v, const int *rows, const int *cols, double *ev, void *ec, int K,
                       bool both, bool row_major, bool wide, int variant)
{
    if (!ensure_ready()) return;
    const size_t total = (size_t)n * K;
    dbuf<double> dv(v, (size_t)upper_nnz), dev(total);
    dbuf<int>    dr(rows, (size_t)upper_nnz), dc(cols, (size_t)upper_nnz);
    dbuf<char>   dec(total * (wide ? 8 : 4));
    if (!dv.ok() || !dev.ok() || !dr.ok() || !dc.ok() || !dec.ok()) return;
    unsigned long long stop = NO_STOP;
    if (cgh_convert_coo_to_ell_device(n, upper_nnz, dv.p, dr.p, dc.p, K, both, row_major, wide, dev.p, dec.p, &stop,
                                      g_cgh.stream) != CGH_OK)
        return;
    dev.get(ev);
    cgh_d2h(ec, dec.p, total * (wide ? 8 : 4));
    if (stop != NO_STOP) {
        // what the reference prints when it stops early (ell.c:299, :753, :765, :825, :836, :849)
        const int64_t k = (int64_t)(stop >> 1);
        const int     i = rows[k], j = cols[k];
        if (j <= i) {
            if (variant == 0) printf("Error: Found a non-upper triangular element in COO format: row=%d, col=%d\n", i, j);
            else if (variant == 1) printf("Error: Found non-upper element in COO: row=%d, col=%d\n", i, j);
            else printf("Error: non-upper COO entry row=%d, col=%d\n", i, j);
        } else if (variant == 1) {
            printf("Error: row %d has more than %d non-zeros\n", i, K);
        } else {
            printf("Error: row %d has >%d off-diag entries\n", (stop & 1) ? j : i, K);
        }
    }
}

// reference src/ell.c:262-309 (row-major, upper only).  The reference does not check for slot overflow here
// (it writes into the next row's slots); this implementation stops at the overflowing entry instead.
void coo_to_ell_symmetric_upper(int n, int upper_nnz, double *coo_values, int *coo_rows, int *coo_cols, double *ell_values,
                                int *ell_cols, int max_nnz_row)
{
    coo_to_ell(n, upper_nnz, coo_values, coo_rows, coo_cols, ell_values, ell_cols, max_nnz_row, false, true, false, 0);
}

// reference src/ell.c:714-776
void coo_to_ell_symmetric_upper_colmajor(int n, int upper_nnz, double *coo_values_upper, int *coo_rows, int *coo_cols,
                                         double *ell_values, int *ell_cols, int max_nnz_row)
{
    coo_to_ell(n, upper_nnz, coo_values_upper, coo_rows, coo_cols, ell_values, ell_cols, max_nnz_row, false, false, false, 1);
}

// reference src/ell.c:791-858
void coo_to_ell_symmetric_full_colmajor(int n, int upper_nnz, double *coo_values_upper, int *coo_rows, int *coo_cols,
                                        double *ell_values, int *ell_cols, int max_nnz_row)
{
    coo_to_ell(n, upper_nnz, coo_values_upper, coo_rows, coo_cols, ell_values, ell_cols, max_nnz_row, true, false, false, 2);
}

// reference src/ell.c:873-940
void coo_to_ell_symmetric_full_colmajor_sdtint(int n, int upper_nnz, double *coo_values_upper, int *coo_rows, int *coo_cols,
                                               double *ell_values, uint64_t *ell_cols, int max_nnz_row)
{
    coo_to_ell(n, upper_nnz, coo_values_upper, coo_rows, coo_cols, ell_values, ell_cols, max_nnz_row, true, false, true, 2);
}

// ------------------------------------------------------------------------------- include/csr.h
// reference src/csr.c:233-272 (also prints one line per trailing empty row, :268)
int coo_to_csr(int n, int upper_count, double *coo_upper_values, int *coo_row_indx, int *coo_col_indx, int *csr_row_ptr)
{
    (void)coo_upper_values;
    (void)coo_col_indx;
    if (!ensure_ready()) return -1;
    dbuf<int> dr(coo_row_indx, (size_t)upper_count), drp((size_t)n + 1);
    if (!dr.ok() || !drp.ok()) return -1;
    if (cgh_coo_row_ptr_device(n, upper_count, dr.p, drp.p, g_cgh.stream) != CGH_OK) return -1;
    if (drp.get(csr_row_ptr) != CGH_OK) return -1;
    if (!quiet()) {
        int last = 0;
        for (int e = 0; e < upper_count; ++e)
            if (coo_row_indx[e] > last) last = coo_row_indx[e];
        for (int r = last; r < n; ++r) printf("Update to csr_row_ptr[%d] = %d \n", r + 2, upper_count);
    }
    return 0;
}

// reference src/csr.c:287-336
void sorting_coo_to_csr(int n, int nnz, const double *values, const int *row_idx, const int *col_idx, int *csr_row_ptr,
                        int *csr_col_idx, double *csr_values)
{
    if (!ensure_ready()) return;
    dbuf<double> dv(values, (size_t)nnz), ov((size_t)nnz);
    dbuf<int>    dr(row_idx, (size_t)nnz), dc(col_idx, (size_t)nnz), orp((size_t)n + 1), oc((size_t)nnz);
    if (!dv.ok() || !ov.ok() || !dr.ok() || !dc.ok() || !orp.ok() || !oc.ok()) return;
    if (cgh_sort_coo_to_csr_device(n, nnz, dv.p, dr.p, dc.p, orp.p, oc.p, ov.p, g_cgh.stream) != CGH_OK) return;
    orp.get(csr_row_ptr);
    oc.get(csr_col_idx);
    ov.get(csr_values);
}

// reference src/csr.c:190-229
void mv_csr_symmetric(int n, const double *diag, const double *upper, const int *col_index, const int *row_ptr, const double *v,
                      double *out)
{
    if (!ensure_ready() || n <= 0) return;
    const int    U = row_ptr[n];
    dbuf<double> dd(diag, (size_t)n), dv(upper, (size_t)U);
    dbuf<int>    drp(row_ptr, (size_t)n + 1), dc(col_index, (size_t)U), dr((size_t)U);
    if (!dd.ok() || !dv.ok() || !drp.ok() || !dc.ok() || !dr.ok()) return;
    hipLaunchKernelGGL(expand_row_ptr, dim3(grid_for(n)), dim3(256), 0, g_cgh.stream, (int64_t)n, (const int *)drp.p, dr.p);
    mat_guard g{cgh_matrix_from_entries_device(n, U, dd.p, dv.p, dr.p, dc.p, CGH_LAYOUT_CSR, true)};
    if (!g.m) return;
    g.m->exact_rows = true;
    spmv_host(g.m, v, out);
}

// ------------------------------------------------------------------------------- include/coo.h
// reference src/coo.c:202-233
void mv_coo_symmetric(int n, int coo_length, const double *diag, const double *value, const int *rows, const int *columns,
                      const double *v, double *out)
{
    if (!ensure_ready()) return;
    mat_guard g{coo_any_from_host(n, coo_length, diag, value, rows, columns, CGH_LAYOUT_CSR)};
    if (!g.m) return;
    g.m->exact_rows = true;
    spmv_host(g.m, v, out);
}

// ------------------------------------------------------------------------------- include/vec.h
static void binary_op(double *a, double *b, double *out, int n, int add)
{
    if (!ensure_ready() || n <= 0) return;
    dbuf<double> da(a, (size_t)n), db(b, (size_t)n), dout((size_t)n);
    if (!da.ok() || !db.ok() || !dout.ok()) return;
    hipLaunchKernelGGL(sub_kernel, dim3(grid_for(n)), dim3(256), 0, g_cgh.stream, (int64_t)n, (const double *)da.p,
                       (const double *)db.p, dout.p, add);
    dout.get(out);
}

void vec_sub(double *a, double *b, double *out, int n) { binary_op(a, b, out, n, 0); }  // reference src/vec.c:39-45
void vec_add(double *a, double *b, double *out, int n) { binary_op(a, b, out, n, 1); }  // reference src/vec.c:48-54

// reference src/vec.c:57-64: a plain memcpy between host buffers; there is nothing for a GPU to do
void vec_assign(double *a, double *b, int n)
{
    if (n > 0) memcpy(a, b, (size_t)n * sizeof(double));
}

// reference src/vec.c:67-75 (strict left-to-right order, as the reference)
double vec_dot(double *a, double *b, int n)
{
    if (!ensure_ready()) return NAN;
    if (n <= 0) return 0.0;
    dbuf<double> da(a, (size_t)n), db(b, (size_t)n);
    double       r = NAN;
    if (da.ok() && db.ok()) cgh_dot(da.p, db.p, n, 1, &r);
    return r;
}

// reference src/vec.c:77-85
double vec_l1norm(double *a, int n)
{
    if (!ensure_ready()) return NAN;
    if (n <= 0) return 0.0;
    dbuf<double> da(a, (size_t)n);
    double       r = NAN;
    if (da.ok()) cgh_l1norm(da.p, n, 1, &r);
    return r;
}

// reference src/vec.c:88-94; out may alias a or b
void vec_axpy(double *a, double *b, double alpha, double *out, int n)
{
    if (!ensure_ready() || n <= 0) return;
    dbuf<double> da(a, (size_t)n), db(b, (size_t)n), dout((size_t)n);
    if (!da.ok() || !db.ok() || !dout.ok()) return;
    if (cgh_axpy(da.p, db.p, alpha, dout.p, n) == CGH_OK) dout.get(out);
}

static double sum_sq_diff(const double *a, const double *b, int n)
{
    dbuf<double> da(a, (size_t)n), db(b, (size_t)n), dd((size_t)n);
    double       r = NAN;
    if (!da.ok() || !db.ok() || !dd.ok()) return r;
    hipLaunchKernelGGL(sub_kernel, dim3(grid_for(n)), dim3(256), 0, g_cgh.stream, (int64_t)n, (const double *)da.p,
                       (const double *)db.p, dd.p, 0);
    cgh_dot(dd.p, dd.p, n, 1, &r);
    return r;
}

// reference src/vec.c:97-104
double rmse(const double *a, const double *b, int n)
{
    if (!ensure_ready()) return NAN;
    return sqrt((n > 0 ? sum_sq_diff(a, b, n) : 0.0) / n);
}

// reference src/vec.c:107-114
double euclidean_distance(const double *a, const double *b, int n)
{
    if (!ensure_ready()) return NAN;
    return sqrt(n > 0 ? sum_sq_diff(a, b, n) : 0.0);
}

// reference src/vec.c:117-126
double max_difference(const double *a, const double *b, int n)
{
    if (!ensure_ready()) return NAN;
    if (n <= 0) return 0.0;
    dbuf<double>             da(a, (size_t)n), db(b, (size_t)n);
    dbuf<unsigned long long> m(1);
    if (!da.ok() || !db.ok() || !m.ok()) return NAN;
    cgh_memset(m.p, 0, sizeof(unsigned long long));
    hipLaunchKernelGGL(absdiff_max, dim3(grid_for(n)), dim3(256), 0, g_cgh.stream, (int64_t)n, (const double *)da.p,
                       (const double *)db.p, m.p);
    unsigned long long bits = 0;
    if (m.get(&bits) != CGH_OK) return NAN;
    double r;
    memcpy(&r, &bits, sizeof r);
    return r;
}

} // extern "C"
