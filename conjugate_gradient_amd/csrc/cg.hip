// cg.hip -- fused vector kernels and the device-resident CG driver.
//
// Replaces the loop of reference src/cg.c:414-508 (and its four textual twins :21,:121,:225,:319)
// plus the BLAS-1 sweeps it calls (src/vec.c:39-94).  Per iteration the reference makes
// 1 SpMV + 2 dot + 3 axpy + 2-3 l1norm sweeps (12K+136 B/row); here an iteration is three kernels
// moving the fused minimum 12K+96 B/row:
//
//   K1  spmv(+dot)   Ap = A p, per-workgroup partial of p.Ap                 (spmv_*.hip)
//   K2  update_r     alpha = rr/(p.Ap) from the partials (every workgroup, same order);
//                    r -= alpha Ap; partials of r.r and |r|_1
//   K3  update_xp    x += alpha p; rr' and l1 from the partials; convergence test; beta = rr'/rr; p = r + beta p
//                    (x is advanced here, not in K2: K3 reads p anyway, which saves one sweep over p per iteration)
//
// Scalars never visit the host: each consumer kernel re-derives them from the producer's
// partials in its prologue (<= 2048 doubles, L2-resident), workgroup 0 records them in the
// device-side history, and a device flag freezes x/r/p at the converging iteration so the host
// may queue iterations ahead and look at the flag only every `check_every` iterations.
//
// strict=1 replaces the tree reductions by single-lane left-to-right sums (seq_sums), which
// makes the whole trajectory bit-identical to the reference's; it is the parity mode, not the
// throughput mode.
#include <cmath>
#include <stdexcept>
#include <vector>

#include "cgh_internal.h"
#include "device_utils.h"

#include "cg_kernels.h"
#include "cg_persistent.h"

namespace {

struct scratch_buf {
    void *p = nullptr;
    explicit scratch_buf(size_t bytes) { p = cgh_alloc(bytes); }
    ~scratch_buf() { cgh_free(p); }
    scratch_buf(const scratch_buf &) = delete;
    scratch_buf &operator=(const scratch_buf &) = delete;
};

// Grid of the vector kernels: four workgroups per CU, each walking its chunks of 512 rows grid-stride.  (Eight per CU --
// every workgroup slot of the chip taken at once -- was the choice of rounds 1 and 2; measured in round 3, us per
// iteration of the throughput loop, 8 -> 4 per CU: 256^3 408 -> 400, 4.2M rows 104.7 -> 102, 1M rows 35.0 -> 31.8; three per CU
// the same, grids that are no multiple of the CU count (640, 896) lose 3 %.)
int vec_grid(int64_t n)
{
    return cgh_stream_grid((n + 2 * CGH_BLOCK - 1) / (2 * CGH_BLOCK), 4);
}

// Vectors of at least this many elements take the order-exact PARALLEL sums of xsum.hip in strict mode; shorter ones
// the single-lane seq_sums kernel (same bits either way).  CGH_XSUM_MIN_ROWS overrides (0 = always, -1 = never).
int64_t xsum_min_rows() { return g_cfg.xsum_min_rows < 0 ? INT64_MAX : g_cfg.xsum_min_rows; }

struct xsum_guard {
    cgh_xsum *w = nullptr;
    ~xsum_guard() { cgh_xsum_free(w); }
};

int reduce_pair(const double *a, const double *b, const double *c, int64_t n, int strict, double *h_out2)
{
    scratch_buf parts(sizeof(double) * 2 * CGH_MAX_PARTIALS), out(sizeof(double) * 2);
    if (!parts.p || !out.p) return CGH_ENOMEM;
    hipStream_t s = g_cgh.stream;
    xsum_guard  xg;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
    if (strict && n >= xsum_min_rows() && !(al & 15u)) {
        xg.w = cgh_xsum_create(n);
        if (!xg.w) return CGH_ENOMEM;
        CGH_HIP(hipMemsetAsync(out.p, 0, 2 * sizeof(double), s));
        const int rc = cgh_xsum_launch(xg.w, n, a, b, c, 0, nullptr, 0, 0.0, nullptr, (double *)out.p, s);
        if (rc) return rc;
        return cgh_d2h(h_out2, out.p, 2 * sizeof(double));  // synchronises before the workspace is released
    }
    if (strict) {
        hipLaunchKernelGGL(seq_sums, dim3(1), dim3(CGH_BLOCK), 0, s, n, a, b, c, 0, (double *)nullptr, 0, 0.0,
                           (int *)nullptr, (double *)out.p);
    } else {
        const int g = vec_grid(n);
        hipLaunchKernelGGL(dot_l1_partials, dim3(g), dim3(CGH_BLOCK), 0, s, n, a, b, c, (double *)parts.p);
        hipLaunchKernelGGL(finish_pair, dim3(1), dim3(CGH_BLOCK), 0, s, (const double *)parts.p, g, (double *)out.p);
    }
    CGH_HIP(hipGetLastError());
    return cgh_d2h(h_out2, out.p, 2 * sizeof(double));
}

} // namespace

struct cgh_solver {
    const cgh_matrix *m = nullptr;
    int64_t           n = 0;
    int               capacity = 0;   // iterations the history can hold
    int               queued = 0;     // iterations queued since start (or since the last rebase)
    int               base = 0;       // iterations finished before the current history window (cgh_solver_rebase)
    int               vgrid = 0;
    bool              strict = false;
    int               precond = CGH_PRECOND_NONE;
    double            tol = 0.0;
    double           *x = nullptr;    // caller's solution vector
    double           *r = nullptr, *p = nullptr, *Ap = nullptr;
    double           *partA = nullptr, *partB = nullptr, *hist = nullptr;
    int              *done = nullptr;
    cgh_xsum         *xs = nullptr;   // order-exact parallel sums (strict mode of long vectors), created on demand
    bool              no_alloc = false;  // cgh_cg_solve_noalloc: no launch of this solver may allocate
    // one resident launch per batch of iterations (cg_persistent.h): tagged images of r (pl) and of the workgroups' partials
    uint64_t         *pl = nullptr, *pa = nullptr, *pb = nullptr;
    unsigned          epoch = 0;         // of the current solve: part of every tag
    unsigned          tag_period = ~0u;  // epoch >> CGP_EPOCH_BITS when the images were last cleared (tags repeat beyond that)
    unsigned         *tag_period_home = nullptr;  // where that lives when the images outlive the solver (the no-allocation workspace)
    std::vector<hipEvent_t> kev;      // SpMV kernel brackets (time_kernels)
    // HIP graphs of `count` iterations (launch-bound sizes): kernels take the iteration number from the device
    std::vector<std::pair<int, hipGraphExec_t>> graphs;
};

// The strict-mode sums of the solver (modes: see seq_sums): parallel order-exact sums for long vectors, one lane otherwise.
static int strict_sums(cgh_solver *sv, const double *a, const double *b, const double *c, int mode, int iter, hipStream_t s,
                       bool presummed = false, double *xu_x = nullptr, const double *xu_p = nullptr, int *xu_taken = nullptr)
{
    if (xu_taken) *xu_taken = 0;
    if (sv->xs)
        return cgh_xsum_launch(sv->xs, sv->n, a, b, c, mode, sv->hist, iter, sv->tol, sv->done, nullptr, s, presummed, xu_x, xu_p, xu_taken);
    hipLaunchKernelGGL(seq_sums, dim3(1), dim3(CGH_BLOCK), 0, s, sv->n, a, b, c, mode, sv->hist, iter, sv->tol, sv->done,
                       (double *)nullptr);
    return CGH_OK;
}

extern "C" {

int cgh_dot(const double *d_a, const double *d_b, int64_t n, int strict, double *h_out)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    double two[2] = {0.0, 0.0};
    if (n > 0) {
        int rc = reduce_pair(d_a, d_b, nullptr, n, strict, two);
        if (rc) return rc;
    }
    *h_out = two[0];
    return CGH_OK;
}

int cgh_l1norm(const double *d_a, int64_t n, int strict, double *h_out)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    double two[2] = {0.0, 0.0};
    if (n > 0) {
        int rc = reduce_pair(nullptr, nullptr, d_a, n, strict, two);
        if (rc) return rc;
    }
    *h_out = two[1];
    return CGH_OK;
}

int cgh_axpy(const double *d_a, const double *d_b, double alpha, double *d_out, int64_t n)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (n <= 0) return CGH_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(cgh_stream_grid((n + CGH_BLOCK - 1) / CGH_BLOCK, 8)), dim3(CGH_BLOCK), 0,
                       g_cgh.stream, n, d_a, d_b, alpha, d_out);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

cgh_solver *cgh_solver_create(const cgh_matrix *m, int capacity)
{
    CGH_REQUIRE_READY(nullptr);
    if (!m || m->n <= 0 || capacity < 1) {
        cgh_set_error("cgh_solver_create: bad arguments");
        return nullptr;
    }
    cgh_solver *sv = new cgh_solver();
    sv->m = m;
    sv->n = m->n;
    sv->capacity = capacity;
    const size_t vbytes = sizeof(double) * (size_t)m->n;
    sv->r = (double *)cgh_alloc(vbytes);
    sv->p = (double *)cgh_alloc(vbytes);
    sv->Ap = (double *)cgh_alloc(vbytes);
    sv->partA = (double *)cgh_alloc(sizeof(double) * CGH_MAX_PARTIALS);
    sv->partB = (double *)cgh_alloc(sizeof(double) * 2 * CGH_MAX_PARTIALS);
    sv->hist = (double *)cgh_alloc(sizeof(double) * HIST * (size_t)(capacity + 1));
    sv->done = (int *)cgh_alloc(sizeof(int) * 4);
    sv->vgrid = vec_grid(m->n);
    if (!sv->r || !sv->p || !sv->Ap || !sv->partA || !sv->partB || !sv->hist || !sv->done) {
        cgh_solver_free(sv);
        return nullptr;
    }
    return sv;
}

void cgh_solver_free(cgh_solver *sv)
{
    if (!sv) return;
    for (hipEvent_t e : sv->kev) (void)hipEventDestroy(e);
    for (auto &g : sv->graphs) (void)hipGraphExecDestroy(g.second);
    cgh_free(sv->r);
    cgh_free(sv->p);
    cgh_free(sv->Ap);
    cgh_free(sv->partA);
    cgh_free(sv->partB);
    cgh_free(sv->hist);
    cgh_free(sv->done);
    cgh_xsum_free(sv->xs);
    cgh_free(sv->pl);
    cgh_free(sv->pa);
    cgh_free(sv->pb);
    delete sv;
}

// Preconditioner of the next cgh_solver_start (additive; the reference has none).
int cgh_solver_set_precond(cgh_solver *sv, int kind)
{
    if (!sv || (kind != CGH_PRECOND_NONE && kind != CGH_PRECOND_JACOBI)) {
        cgh_set_error("cgh_solver_set_precond: unknown preconditioner %d", kind);
        return CGH_EINVAL;
    }
    sv->precond = kind;
    return CGH_OK;
}

// x = 0; r = b - A x; p = r; rr = r.r (reference cg.c:433-456).  Asynchronous.
int cgh_solver_start(cgh_solver *sv, const double *d_b, double *d_x, double tol, int strict)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!sv || !d_b || !d_x) {
        cgh_set_error("cgh_solver_start: NULL argument");
        return CGH_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(d_b) | reinterpret_cast<uintptr_t>(d_x)) & 15u) {
        cgh_set_error("cgh_solver_start: d_b and d_x must be 16-byte aligned");
        return CGH_EINVAL;
    }
    if (strict && (sv->m->layout == CGH_LAYOUT_CSRB || sv->m->layout == CGH_LAYOUT_PB)) {
        cgh_set_error("cgh_solver_start: the column-blocked layouts sum a row block by block, not in the reference's order: they are "
                      "throughput layouts (strict=0); for a bit-identical trajectory use CGH_LAYOUT_CSR with exact rows, ELL or sliced ELL");
        return CGH_EINVAL;
    }
    hipStream_t   s = g_cgh.stream;
    const int64_t n = sv->n;
    sv->x = d_x;
    sv->tol = tol;
    sv->strict = strict != 0;
    sv->queued = 0;
    sv->base = 0;
    static unsigned solve_epoch = 0;
    sv->epoch = ++solve_epoch;
    for (hipEvent_t e : sv->kev) (void)hipEventDestroy(e);
    sv->kev.clear();
    for (auto &g : sv->graphs) (void)hipGraphExecDestroy(g.second);  // the captured launches carry x, b, tol, strict
    sv->graphs.clear();
    CGH_HIP(hipMemsetAsync(sv->done, 0, sizeof(int) * 4, s));
    CGH_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)n, s));
    cgh_spmv_opts so;
    so.no_alloc = sv->no_alloc;
    int rc = cgh_launch_spmv(sv->m, d_x, sv->Ap, nullptr, nullptr, nullptr, s, &so);
    if (rc) return rc;
    if (sv->precond == CGH_PRECOND_JACOBI) {
        if (sv->strict) {
            cgh_set_error("cgh_solver_start: the Jacobi preconditioner has no reference order to be strict about (use strict=0)");
            return CGH_EINVAL;
        }
        if (n & 1 ? false : (reinterpret_cast<uintptr_t>(sv->m->diag) & 15u) != 0) return CGH_EINVAL;
        hipLaunchKernelGGL(pcg_init_rp, dim3(sv->vgrid), dim3(CGH_BLOCK), 0, s, n, d_b, (const double *)sv->Ap,
                           (const double *)sv->m->diag, sv->r, sv->p, sv->partB);
    } else
        hipLaunchKernelGGL(cg_init_rp, dim3(sv->vgrid), dim3(CGH_BLOCK), 0, s, n, d_b, (const double *)sv->Ap, sv->r, sv->p,
                           sv->partB);
    if (sv->strict) {
        if (!sv->xs && n >= xsum_min_rows()) {
            sv->xs = cgh_xsum_create(n);
            if (!sv->xs) return CGH_ENOMEM;
        }
        rc = strict_sums(sv, sv->r, sv->r, sv->r, 3, 0, s);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(cg_finish_init, dim3(1), dim3(CGH_BLOCK), 0, s, (const double *)sv->partB, sv->vgrid, sv->hist);
    }
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

} // extern "C"

// Small systems in throughput mode: can this solver's iterations run as one resident launch (cg_persistent.h)?  At most
// 256 tiles (all workgroups resident at once, one lane per partial), the vector path of the ELL SpMV with a compile-time
// width, no preconditioner.  CGH_PERSISTENT=0 keeps the three launches per iteration.
static bool persistent_eligible(const cgh_solver *sv)
{
    const cgh_matrix *m = sv->m;
    if (!g_cfg.persistent || sv->strict || sv->precond != CGH_PRECOND_NONE || g_cfg.serpentine) return false;
    if (sv->no_alloc && !sv->pl) return false;  // a solver that must not allocate brings the images along or does without
    if (m->layout != CGH_LAYOUT_ELL || g_cfg.ell_generic || m->K < 1 || m->K > 8) return false;
    if ((m->n + 2 * CGH_BLOCK - 1) / (2 * CGH_BLOCK) > CGP_MAX_TILES) return false;
    if (!cgh_ell_vector_path(m, sv->p, sv->Ap) || (reinterpret_cast<uintptr_t>(sv->x) & 15u)) return false;
    // (dict_state 0: a solver that must not allocate has not built the column dictionary either -- int32 columns, as its SpMV)
    if (!(m->dict_state == 2 || (m->dict_state == 1 && m->dict_mode == 1) || (m->dict_state == 0 && sv->no_alloc))) return false;
    // A captured launch would be replayed with the SAME tags over the previous replay's images: not under capture
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(g_cgh.stream, &cap);
    return cap == hipStreamCaptureStatusNone;
}

static long long *g_persistent_trace = nullptr;  // tools: cgh_persistent_set_trace
static long long  g_persistent_launches = 0;     // tests: cgh_persistent_launches

template <int DICT>
static void launch_persistent(cgh_solver *sv, int tiles, int first_iter, int count, hipStream_t s)
{
    const cgh_matrix *m = sv->m;
    const void       *cols = DICT ? (const void *)m->cols8 : (const void *)m->cols;
#define CGH_PERSISTENT_CASE(KK)                                                                                                 \
    case KK:                                                                                                                    \
        hipLaunchKernelGGL((cg_persistent<KK, DICT>), dim3(tiles), dim3(CGH_BLOCK), 0, s, m->n, (const double *)m->diag,        \
                           (const double *)m->vals, cols, (const int *)m->off_tab, sv->x, sv->r, sv->p, sv->pl, sv->pa, sv->pb, \
                           sv->hist, first_iter, count, sv->tol, sv->done, sv->epoch, g_persistent_trace);                      \
        break;
    switch (m->K) {
        CGH_PERSISTENT_CASE(1)
        CGH_PERSISTENT_CASE(2)
        CGH_PERSISTENT_CASE(3)
        CGH_PERSISTENT_CASE(4)
        CGH_PERSISTENT_CASE(5)
        CGH_PERSISTENT_CASE(6)
        CGH_PERSISTENT_CASE(7)
        CGH_PERSISTENT_CASE(8)
    }
#undef CGH_PERSISTENT_CASE
}

static int enqueue_persistent(cgh_solver *sv, int count)
{
    hipStream_t s = g_cgh.stream;
    const int   tiles = (int)((sv->n + 2 * CGH_BLOCK - 1) / (2 * CGH_BLOCK));
    // images: two tagged words per entry of r (whole tiles), four per workgroup for p.Ap and for (r.r, |r|_1)
    const size_t pl_bytes = sizeof(uint64_t) * 2 * (size_t)tiles * 2 * CGH_BLOCK, part_bytes = sizeof(uint64_t) * 4 * CGP_MAX_TILES;
    if (!sv->pl) {
        sv->pl = (uint64_t *)cgh_alloc(pl_bytes);
        sv->pa = (uint64_t *)cgh_alloc(part_bytes);
        sv->pb = (uint64_t *)cgh_alloc(part_bytes);
        if (!sv->pl || !sv->pa || !sv->pb) return CGH_ENOMEM;
    }
    // a cleared buffer carries tag 0, which no exchange uses; tags repeat after 2^18 solves, so the images are cleared
    // whenever the epoch's higher bits differ from those of the last clearing (always the first time)
    unsigned *period = sv->tag_period_home ? sv->tag_period_home : &sv->tag_period;
    if (*period != sv->epoch >> CGP_EPOCH_BITS) {
        CGH_HIP(hipMemsetAsync(sv->pl, 0, pl_bytes, s));
        CGH_HIP(hipMemsetAsync(sv->pa, 0, part_bytes, s));
        CGH_HIP(hipMemsetAsync(sv->pb, 0, part_bytes, s));
        *period = sv->epoch >> CGP_EPOCH_BITS;
    }
    const int first = sv->queued + 1;
    if (sv->m->dict_state == 1 && sv->m->dict_mode == 1)
        launch_persistent<1>(sv, tiles, first, count, s);
    else
        launch_persistent<0>(sv, tiles, first, count, s);
    sv->queued += count;
    ++g_persistent_launches;
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

// Enqueue `count` iterations.  device_iter: the kernels read the iteration number from the device (graph capture).
static int enqueue_iterations(cgh_solver *sv, int count, int time_kernels, bool device_iter)
{
    if (count > 0 && !time_kernels && !device_iter && persistent_eligible(sv)) return enqueue_persistent(sv, count);
    hipStream_t   s = g_cgh.stream;
    const int64_t n = sv->n;
    const int     vg = sv->vgrid;
    double       *dr = sv->r, *dp = sv->p, *dAp = sv->Ap, *dA = sv->partA, *dB = sv->partB, *dh = sv->hist, *d_x = sv->x;
    int          *dd = sv->done;
    const double  tol = sv->tol;
    for (int c = 0; c < count; ++c) {
        const int iter = device_iter ? 0 : ++sv->queued;
        const int host_iter = device_iter ? sv->queued + c + 1 : iter;  // parity of the sweep direction only
        int       gridA = 0;
        // serpentine sweeps: each kernel starts where the previous one ended, so the tail of what was just
        // written/read could still be in the 256 MiB Infinity Cache.  Measured null on MI355X (0.4791 vs 0.4797
        // ms/iter, 256^3), so off by default; env CGH_SERPENTINE=1 enables it for experiments.
        const int serp = g_cfg.serpentine;
        const int rev1 = serp ? (host_iter & 1) : 0, rev2 = serp ? !rev1 : 0;
        const bool timed = time_kernels > 0 && c % time_kernels == 0;  // every time_kernels-th iteration of this call
        if (timed) {
            hipEvent_t a, b;
            CGH_HIP(hipEventCreate(&a));
            CGH_HIP(hipEventCreate(&b));
            sv->kev.push_back(a);
            sv->kev.push_back(b);
            CGH_HIP(hipEventRecord(a, s));
        }
        // strict mode with the parallel sums: the SpMV leaves p.Ap per 128-row cell (their binade prediction) if it can
        bool          pAp_presummed = false;
        cgh_spmv_opts so;
        so.cell_sums = sv->strict && sv->precond == CGH_PRECOND_NONE ? cgh_xsum_cellsums(sv->xs) : nullptr;
        so.cell_sums_written = &pAp_presummed;
        so.reverse = rev1;
        so.no_alloc = sv->no_alloc;
        int rc = cgh_launch_spmv(sv->m, dp, dAp, sv->strict ? nullptr : dA, &gridA, dd, s, &so);
        if (rc) return rc;
        if (timed) CGH_HIP(hipEventRecord(sv->kev.back(), s));
        if (sv->precond == CGH_PRECOND_JACOBI) {
            const double *dg = sv->m->diag;
            hipLaunchKernelGGL(pcg_update_r, dim3(vg), dim3(CGH_BLOCK), 0, s, n, dr, (const double *)dAp, dg, (const double *)dA,
                               gridA, dB, dh, iter, (const int *)dd);
            hipLaunchKernelGGL(pcg_update_xp, dim3(vg), dim3(CGH_BLOCK), 0, s, n, d_x, (const double *)dr, dp, dg,
                               (const double *)dB, vg, dh, iter, tol, dd);
        } else if (sv->strict) {
            // (few cells: the launch of the sums of p.Ap keeps its workgroups for r -= alpha Ap behind the verdict, xs_tail)
            int r_done = 0;
            rc = strict_sums(sv, dp, dAp, nullptr, 1, iter, s, pAp_presummed, device_iter || serp ? nullptr : dr, nullptr, &r_done);
            if (rc) return rc;
            if (r_done != 2) {
                // with the parallel sums, the r update also leaves their per-cell approximate sums (one sweep over r less)
                // (whole groups of the sums' hierarchy per workgroup, so that the group sums come out as well -- unless the sweep
                // direction is reversed, or the sums are so short that their launch adds up the cell sums itself)
                double *gsums = rev2 || cgh_xsum_cells(sv->xs) <= 4096 ? nullptr : cgh_xsum_groupsums(sv->xs);
                hipLaunchKernelGGL((cg_update_r<1, 3>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, dr, (const double *)dAp,
                                   (const double *)dA, gridA, dB, dh, iter, (const int *)dd, rev2, (const double *)nullptr,
                                   cgh_xsum_cellsums(sv->xs), cgh_xsum_cells(sv->xs), gsums);
                if (gsums) cgh_xsum_groups_written(sv->xs);
            }
            // ... and the launch of the sums of r may take x += alpha p along (xsum.hip, xs_fused): then K3 only updates p --
            // or, with few cells, both updates behind its verdict (xs_tail): then there is no K3
            int x_done = 0;
            rc = strict_sums(sv, dr, dr, dr, 2, iter, s, sv->xs != nullptr, d_x, dp, &x_done);
            if (rc) return rc;
            if (x_done != 2)
                hipLaunchKernelGGL((cg_update_xp<1, 3>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, x_done ? (double *)nullptr : d_x, (const double *)dr, dp,
                                   (const double *)dB, vg, dh, iter, tol, dd, rev1, (const double *)nullptr);
        } else {
            // non-temporal loads + x store (3) unless the working set is within twice the Infinity Cache (0): see cgh_fits_llc
            const int vv_env = g_cfg.vec_variant >= 0 ? g_cfg.vec_variant & 15 : -1;
            const int vv = vv_env >= 0 ? vv_env : (sv->m->layout == CGH_LAYOUT_ELL && cgh_fits_llc(sv->m, 2.0) ? 0 : 3);
#define CGH_VEC_VAR(V)                                                                                                 \
    case V:                                                                                                            \
        hipLaunchKernelGGL((cg_update_r<0, V>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, dr, (const double *)dAp,       \
                           (const double *)dA, gridA, dB, dh, iter, (const int *)dd, rev2, (const double *)nullptr);   \
        hipLaunchKernelGGL((cg_update_xp<0, V>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, d_x, (const double *)dr, dp,  \
                           (const double *)dB, vg, dh, iter, tol, dd, rev1, (const double *)nullptr);                  \
        break;
            switch (vv) {
                CGH_VEC_VAR(0)
                CGH_VEC_VAR(1)
                CGH_VEC_VAR(3)
                CGH_VEC_VAR(7)
                CGH_VEC_VAR(11)
                CGH_VEC_VAR(15)
            default:
                cgh_set_error("CGH_VEC_VARIANT=%d is not built (0,1,3,7,11,15)", vv);
                return CGH_EINVAL;
            }
#undef CGH_VEC_VAR
        }
    }
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

extern "C" {

// Queue `count` more iterations on the compute stream (no host synchronisation).  With CGH_GRAPH=1 a batch is replayed
// from a HIP graph (the kernels then take the iteration number from the device).  Measured on MI355X: foam-32k, 999
// iterations 15.4 -> 12.6 us/iteration, but instantiating the graphs costs ~1 ms each, which a 253-iteration solve of
// foam-2000 does not earn back (3.9 -> 5.9 ms) -- at these sizes an iteration is three ~3 us latency-bound kernels, not
// host launch overhead -- so it is opt-in.
int cgh_solver_iterate(cgh_solver *sv, int count, int time_kernels)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!sv || count < 0 || sv->queued + count > sv->capacity) {
        cgh_set_error("cgh_solver_iterate: count %d exceeds the solver's capacity", count);
        return CGH_EINVAL;
    }
    const int     use_graphs = g_cfg.graph;
    const int64_t graph_rows = g_cfg.graph_max_rows;
    if (!use_graphs || time_kernels || count < 4 || sv->n > graph_rows) return enqueue_iterations(sv, count, time_kernels, false);
    hipStream_t    s = g_cgh.stream;
    hipGraphExec_t exec = nullptr;
    for (auto &g : sv->graphs)
        if (g.first == count) exec = g.second;
    if (!exec) {
        hipGraph_t graph = nullptr;
        CGH_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue_iterations(sv, count, 0, true);
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc != CGH_OK || e != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            cgh_set_error("cgh_solver_iterate: graph capture failed (%s)", hipGetErrorString(e));
            return rc != CGH_OK ? rc : CGH_EHIP;
        }
        const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) {
            cgh_set_error("cgh_solver_iterate: hipGraphInstantiate -> %s", hipGetErrorString(ei));
            return CGH_EHIP;
        }
        sv->graphs.emplace_back(count, exec);
    }
    CGH_HIP(hipGraphLaunch(exec, s));
    sv->queued += count;
    return CGH_OK;
}

// 1 if this solver's next iterations run as one resident launch per batch (cg_persistent.h): after cgh_solver_start.
int cgh_solver_is_resident(const cgh_solver *sv) { return sv && persistent_eligible(sv) ? 1 : 0; }

// Tests: resident launches queued by this process so far.
long long cgh_persistent_launches(void) { return g_persistent_launches; }

// Tools: per-workgroup time stamps of the resident launch (8 long long per workgroup, 256 workgroups; NULL switches off).
int cgh_persistent_set_trace(long long *d_trace)
{
    g_persistent_trace = d_trace;
    return CGH_OK;
}

// Drain the stream and read the device's verdict: *done_iter = converging iteration (0 = not yet).
int cgh_solver_poll(cgh_solver *sv, int *done_iter, int *queued)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!sv) return CGH_EINVAL;
    int d[4] = {0, 0, 0, 0};
    CGH_HIP(hipMemcpyAsync(d, sv->done, sizeof(int) * 4, hipMemcpyDeviceToHost, g_cgh.stream));
    CGH_HIP(hipStreamSynchronize(g_cgh.stream));
    if (done_iter) *done_iter = d[0];
    if (queued) *queued = sv->queued;
    if (d[3] != 0) {
        cgh_set_error("cgh_solver_poll: a workgroup of the resident CG launch waited 2 s for another one (is the GPU shared? "
                      "CGH_PERSISTENT=0 keeps three launches per iteration); the results are NaN");
        return CGH_EHIP;
    }
    return CGH_OK;
}

// Copy history entries [first, first+count) (4 doubles each) to the host; synchronises.
int cgh_solver_history(cgh_solver *sv, int first, int count, double *h_out)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!sv || first < 0 || count < 0 || first + count > sv->capacity + 1) return CGH_EINVAL;
    return cgh_d2h(h_out, sv->hist + (size_t)first * HIST, sizeof(double) * HIST * (size_t)count);
}

// Average SpMV kernel time (ms) over the iterations queued with time_kernels=1 since start; synchronises.
int cgh_solver_spmv_ms(cgh_solver *sv, double *ms_avg)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!sv || !ms_avg) return CGH_EINVAL;
    CGH_HIP(hipStreamSynchronize(g_cgh.stream));
    double tot = 0.0;
    for (size_t k = 0; k + 1 < sv->kev.size(); k += 2) {
        float t = 0.f;
        CGH_HIP(hipEventElapsedTime(&t, sv->kev[k], sv->kev[k + 1]));
        tot += t;
    }
    *ms_avg = sv->kev.empty() ? 0.0 : tot / (double)(sv->kev.size() / 2);
    return CGH_OK;
}

const double *cgh_solver_residual(const cgh_solver *sv) { return sv ? sv->r : nullptr; }

// The device history holds `capacity`+1 entries; a solve longer than that continues in a fresh window: the last entry
// becomes entry 0 (the kernels of iteration k read entry k-1), the device-side iteration counters restart, and the
// iterations finished so far are counted in sv->base.  Only between polls, and only while not converged.
int cgh_solver_rebase(cgh_solver *sv)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!sv) return CGH_EINVAL;
    if (sv->queued == 0) return CGH_OK;
    hipStream_t s = g_cgh.stream;
    CGH_HIP(hipMemcpyAsync(sv->hist, sv->hist + (size_t)sv->queued * HIST, sizeof(double) * HIST, hipMemcpyDeviceToDevice, s));
    CGH_HIP(hipMemsetAsync(sv->done + 1, 0, sizeof(int) * 2, s));
    sv->base += sv->queued;
    sv->queued = 0;
    return CGH_OK;
}

int cgh_solver_base(const cgh_solver *sv) { return sv ? sv->base : 0; }

} // extern "C"

namespace {
struct event_pair {
    hipEvent_t a = nullptr, b = nullptr;
    ~event_pair()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};
constexpr int HISTORY_WINDOW = 4096;  // iterations of history the device keeps at a time (128 KB)

void copy_history(void *ctx, int first, int count, const double *e, int, const cgh_cg_report *)
{
    memcpy((double *)ctx + (size_t)HIST * first, e, sizeof(double) * HIST * (size_t)count);
}
} // namespace

// cgh_cg_solve with the history delivered in pieces: sink(ctx, first, count, entries, converged_iteration, report) is
// called with entries [first, first+count) (4 doubles each; first = 0 comes with report->r0 / initial_res filled in) as
// the host fetches them -- the memory of a solve does not grow with max_iter (the reference's loop is O(1) in it).
int cgh_cg_solve_sink(const cgh_matrix *m, const double *d_b, double *d_x, const cgh_cg_opts *opts, cgh_cg_report *rep,
                      cgh_history_sink sink, void *ctx)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!m || !d_b || !d_x || !opts || !rep) {
        cgh_set_error("cgh_cg_solve: NULL argument");
        return CGH_EINVAL;
    }
    const bool    fixed = opts->fixed_iters > 0;
    const int64_t max_iter = fixed ? (int64_t)opts->fixed_iters + 1 : (int64_t)opts->max_iter;
    const double  tol = fixed ? -1.0 : opts->tol;  // l1/2 >= 0 is never < -1: no convergence in benchmark mode
    const int     check = opts->check_every > 0 ? opts->check_every : 32;
    memset(rep, 0, sizeof *rep);
    rep->ret = -1;
    if (m->n <= 0 || max_iter < 1) {
        cgh_set_error("cgh_cg_solve: n=%lld max_iter=%lld", (long long)m->n, (long long)max_iter);
        return CGH_EINVAL;
    }
    const int64_t total = max_iter - 1;  // reference loop: iter = 1 .. max_iter-1 (cg.c:459)
    const int     window = (int)(total < HISTORY_WINDOW ? (total < 1 ? 1 : total) : HISTORY_WINDOW);
    cgh_solver   *sv = cgh_solver_create(m, window);
    if (!sv) return CGH_ENOMEM;
    struct guard {
        cgh_solver *p;
        ~guard() { cgh_solver_free(p); }
    } g{sv};
    hipStream_t s = g_cgh.stream;
    int         rc = cgh_solver_set_precond(sv, opts->precond);
    if (rc) return rc;
    rc = cgh_solver_start(sv, d_b, d_x, tol, opts->strict);
    if (rc) return rc;
    CGH_HIP(hipMemcpyAsync(rep->r0, sv->r, sizeof(double) * (m->n < 5 ? m->n : 5), hipMemcpyDeviceToHost, s));

    event_pair ev;
    CGH_HIP(hipEventCreate(&ev.a));
    CGH_HIP(hipEventCreate(&ev.b));
    CGH_HIP(hipEventRecord(ev.a, s));
    std::vector<double> hh((size_t)HIST * (window + 1));
    int     done_iter = 0;       // converging iteration inside the current window (0 = none yet)
    int     delivered = -1;      // last entry of the current window already handed to the sink
    double  last_res = 0.0;
    // hand entries (delivered, upto] of the current window to the sink
    auto deliver = [&](int upto, int done_local) -> int {
        if (upto <= delivered) return CGH_OK;
        const int first = delivered + 1, count = upto - delivered;
        int       rc2 = cgh_solver_history(sv, first, count, hh.data());
        if (rc2) return rc2;
        if (sv->base == 0 && first == 0) rep->initial_res = hh[0];
        last_res = hh[(size_t)HIST * (count - 1)];
        if (sink) sink(ctx, sv->base + first, count, hh.data(), done_local ? sv->base + done_local : 0, rep);
        delivered = upto;
        return CGH_OK;
    };
    while ((int64_t)sv->base + sv->queued < total && done_iter == 0) {
        if (sv->queued == window) {  // the window is full and nothing converged in it: hand it over, start the next one
            rc = deliver(sv->queued, 0);
            if (rc) return rc;
            rc = cgh_solver_rebase(sv);
            if (rc) return rc;
            delivered = 0;  // entry 0 of the new window is the old window's last entry: already delivered
        }
        int64_t chunk = total - sv->base - sv->queued;
        if (chunk > window - sv->queued) chunk = window - sv->queued;
        // (the resident launch of small systems leaves at the converging iteration by itself: no need to look that often)
        if (!fixed && chunk > check && !(opts->time_kernels == 0 && persistent_eligible(sv))) chunk = check;
        rc = cgh_solver_iterate(sv, (int)chunk, opts->time_kernels);
        if (rc) return rc;
        if (!fixed) {
            rc = cgh_solver_poll(sv, &done_iter, nullptr);
            if (rc) return rc;
        }
    }
    CGH_HIP(hipEventRecord(ev.b, s));
    rc = cgh_solver_poll(sv, &done_iter, nullptr);
    if (rc) return rc;
    float ms = 0.f;
    CGH_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
    rep->solve_ms = ms;
    if (opts->time_kernels) cgh_solver_spmv_ms(sv, &rep->spmv_ms_avg);

    const int executed_local = done_iter ? done_iter : sv->queued;
    rep->iterations = sv->base + executed_local;
    rep->ret = done_iter ? sv->base + done_iter + 1 : -1;
    rc = deliver(executed_local, done_iter);
    if (rc) return rc;
    rep->final_res = last_res;
    return CGH_OK;
}

extern "C" {

int cgh_cg_solve(const cgh_matrix *m, const double *d_b, double *d_x, const cgh_cg_opts *opts, cgh_cg_report *rep,
                 double *h_history)
{
    try {
        return cgh_cg_solve_sink(m, d_b, d_x, opts, rep, h_history ? copy_history : (cgh_history_sink) nullptr, h_history);
    } catch (const std::exception &e) {
        cgh_set_error("cgh_cg_solve: %s", e.what());
        return CGH_ENOMEM;
    }
}

} // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// The one entry point that allocates nothing (no counterpart in the reference; for callers that cannot call hipMalloc:
// code under HIP graph capture, tight serving loops that solve one small system after another, harnesses that forbid
// allocation).  It is cgh_cg_solve with the solver's workspace -- r, p, Ap and the two arrays of per-workgroup partials,
// the device memory beyond the caller's inputs and outputs -- taken from a file-scope __device__ array instead of
// cgh_alloc.  The array's NAME is a host-side stand-in, not a device address: its address on the current device comes
// from hipGetSymbolAddress, once per device (cgh_noalloc_setup, called by cgh_init with that device current), and only
// that pointer is handed to kernels and memsets.  Every call on a GPU shares that GPU's array and finds the previous
// call's contents in it: what must start at zero (the converged flag) is cleared, everything else is written before
// it is read.  Calls that overlap on one GPU (two host threads) would race on it.
constexpr int64_t NOALLOC_MAX_ROWS = 1 << 20;  // 3 vectors of 2^20 doubles + partials + 1 MiB for the order-exact sums + 2 MiB for the resident launch: 28.3 MB per GPU, held until the process ends
constexpr size_t  NOALLOC_PART_A = (size_t)3 * NOALLOC_MAX_ROWS;
constexpr size_t  NOALLOC_PART_B = NOALLOC_PART_A + CGH_MAX_PARTIALS;
constexpr size_t  NOALLOC_DONE = NOALLOC_PART_B + 2 * CGH_MAX_PARTIALS;
constexpr size_t  NOALLOC_XSUM = NOALLOC_DONE + 32;            // workspace of the order-exact sums (strict mode from 4096 rows on)
constexpr size_t  NOALLOC_XSUM_DOUBLES = (size_t)1 << 17;      // 1 MiB; 2^20 rows need 803 840 bytes (cgh_xsum_device_bytes)
// images of the resident launch of small throughput-mode solves (cg_persistent.h): r of 256 tiles, the two arrays of partials
constexpr size_t  NOALLOC_RL = NOALLOC_XSUM + NOALLOC_XSUM_DOUBLES;
constexpr size_t  NOALLOC_RL_DOUBLES = (size_t)2 * CGP_MAX_TILES * 2 * CGH_BLOCK;  // 2 MiB
constexpr size_t  NOALLOC_PA = NOALLOC_RL + NOALLOC_RL_DOUBLES, NOALLOC_PB = NOALLOC_PA + 4 * CGP_MAX_TILES;
constexpr size_t  NOALLOC_DOUBLES = NOALLOC_PB + 4 * CGP_MAX_TILES;
static_assert((NOALLOC_XSUM * sizeof(double)) % 256 == 0, "the sums' workspace starts on a 256-byte boundary");
static __device__ __align__(256) double scratch[NOALLOC_DOUBLES];
static double *g_scratch_ptr[64] = {nullptr};  // per device: each GPU holds its own copy of the array at its own address
static unsigned g_scratch_tag_period[64];      // per device: when the resident launch's images in it were last cleared

int cgh_noalloc_setup(int device)
{
    if (device < 0 || device >= 64) return CGH_OK;  // no scratch on such a device: the entry point reports it
    if (g_scratch_ptr[device] != nullptr) return CGH_OK;
    void *p = nullptr;
    CGH_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = (double *)p;
    g_scratch_tag_period[device] = ~0u;
    return CGH_OK;
}

extern "C" int cgh_cg_solve_noalloc(const cgh_matrix *m, const double *d_b, double *d_x, int max_iter, double tol, int strict,
                                    double *d_history, int *d_status)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!m || !d_b || !d_x || !d_history || !d_status || max_iter < 1) {
        cgh_set_error("cgh_cg_solve_noalloc: NULL argument or max_iter < 1");
        return CGH_EINVAL;
    }
    if (m->n <= 0) {
        cgh_set_error("cgh_cg_solve_noalloc: the matrix has no rows");
        return CGH_EINVAL;
    }
    if (m->n > NOALLOC_MAX_ROWS) {
        cgh_set_error("cgh_cg_solve_noalloc: %lld rows exceed the %lld rows the built-in workspace holds (use cgh_cg_solve)",
                      (long long)m->n, (long long)NOALLOC_MAX_ROWS);
        return CGH_EINVAL;
    }
    double *ws = g_cgh.device >= 0 && g_cgh.device < 64 ? g_scratch_ptr[g_cgh.device] : nullptr;
    if (!ws) {
        cgh_set_error("cgh_cg_solve_noalloc: no workspace on device %d", g_cgh.device);
        return CGH_ENODEV;
    }
    cgh_solver sv;  // on the stack; its vectors stay empty (no time_kernels, no graphs)
    alignas(16) unsigned char xs_obj[256];  // descriptor of the order-exact sums' workspace, also on the stack
    if (strict && m->n >= xsum_min_rows()) {
        // bit-identical sums of a long vector: their workspace (summaries, counters) is the tail of the same array
        if (cgh_xsum_object_bytes() > sizeof xs_obj || cgh_xsum_device_bytes(m->n) > NOALLOC_XSUM_DOUBLES * sizeof(double)) {
            cgh_set_error("cgh_cg_solve_noalloc: the built-in workspace cannot hold the order-exact sums of %lld rows", (long long)m->n);
            return CGH_EINVAL;
        }
        sv.xs = cgh_xsum_place(xs_obj, sizeof xs_obj, m->n, ws + NOALLOC_XSUM, NOALLOC_XSUM_DOUBLES * sizeof(double), g_cgh.stream);
        if (!sv.xs) {
            cgh_set_error("cgh_cg_solve_noalloc: could not lay out the order-exact sums' workspace");
            return CGH_EHIP;
        }
    }
    sv.m = m;
    sv.n = m->n;
    sv.capacity = max_iter;
    sv.r = ws;
    sv.p = ws + NOALLOC_MAX_ROWS;
    sv.Ap = ws + 2 * NOALLOC_MAX_ROWS;
    sv.partA = ws + NOALLOC_PART_A;
    sv.partB = ws + NOALLOC_PART_B;
    sv.done = reinterpret_cast<int *>(ws + NOALLOC_DONE);
    sv.hist = d_history;
    sv.vgrid = vec_grid(m->n);
    sv.no_alloc = true;
    sv.pl = reinterpret_cast<uint64_t *>(ws + NOALLOC_RL);
    sv.pa = reinterpret_cast<uint64_t *>(ws + NOALLOC_PA);
    sv.pb = reinterpret_cast<uint64_t *>(ws + NOALLOC_PB);
    sv.tag_period_home = &g_scratch_tag_period[g_cgh.device];
    int rc = cgh_solver_start(&sv, d_b, d_x, tol, strict);  // clears the converged flag, zeroes x
    if (rc) return rc;
    rc = enqueue_iterations(&sv, max_iter - 1, 0, false);    // cg.c:459: iter = 1 .. max_iter-1; frozen once converged
    if (rc) return rc;
    CGH_HIP(hipMemcpyAsync(d_status, sv.done, sizeof(int) * 4, hipMemcpyDeviceToDevice, g_cgh.stream));
    return CGH_OK;
}

