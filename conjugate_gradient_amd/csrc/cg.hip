// cg.hip -- fused vector kernels and the device-resident CG driver.
//
// Replaces the loop of reference src/cg.c:414-508 (and its four textual twins :21,:121,:225,:319)
// plus the BLAS-1 sweeps it calls (src/vec.c:39-94).  Per iteration the reference makes
// 1 SpMV + 2 dot + 3 axpy + 2-3 l1norm sweeps (12K+136 B/row); here an iteration is three kernels
// moving the fused minimum 12K+96 B/row:
//
//   K1  spmv(+dot)   Ap = A p, per-workgroup partial of p.Ap                 (spmv_*.hip)
//   K2  update_xr    alpha = rr/(p.Ap) from the partials (every workgroup, same order);
//                    x += alpha p; r -= alpha Ap; partials of r.r and |r|_1
//   K3  update_p     rr' and l1 from the partials; convergence test; beta = rr'/rr; p = r + beta p
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
#include <vector>

#include "cgh_internal.h"
#include "device_utils.h"

int cgh_launch_spmv(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                    const int *done, hipStream_t s);

namespace {

constexpr int HIST = 4;  // doubles per history entry: res, rr, alpha, beta

// ---------------------------------------------------------------- tree-mode kernels

// r = b - Ap0 (Ap0 = A*0 computed by the SpMV, as the reference does), p = r; partials of r.r, |r|_1
__global__ __launch_bounds__(CGH_BLOCK) void cg_init_rp(int64_t n, const double *__restrict__ b, const double *__restrict__ Ap0,
                                                       double *__restrict__ r, double *__restrict__ p,
                                                       double *__restrict__ partials)
{
    __shared__ double lds[4];
    double            rr = 0.0, l1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CGH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CGH_BLOCK) {
        const double v = b[i] - Ap0[i];
        r[i] = v;
        p[i] = v;
        rr += v * v;
        l1 += fabs(v);
    }
    const double trr = block_sum_256(rr, lds);
    const double tl1 = block_sum_256(l1, lds);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = trr;
        partials[2 * blockIdx.x + 1] = tl1;
    }
}

__global__ __launch_bounds__(CGH_BLOCK) void cg_finish_init(const double *__restrict__ partials, int count, double *hist)
{
    __shared__ double lds[4];
    const double      rr = block_sum_partials(partials, count, 2, lds);
    const double      l1 = block_sum_partials(partials + 1, count, 2, lds);
    if (threadIdx.x == 0) {
        hist[0] = l1 / 2.0;
        hist[1] = rr;
        hist[2] = NAN;
        hist[3] = NAN;
    }
}

// K2.  VEC2: n even and 16-B aligned vectors.
template <bool STRICT>
__global__ __launch_bounds__(CGH_BLOCK) void cg_update_xr(int64_t n, double *__restrict__ x, const double *__restrict__ p,
                                                         double *__restrict__ r, const double *__restrict__ Ap,
                                                         const double *__restrict__ pAp_partials, int pAp_count,
                                                         double *__restrict__ out_partials, double *hist, int iter,
                                                         const int *__restrict__ done)
{
    __shared__ double lds[4];
    if (*done != 0) return;
    double alpha;
    if constexpr (STRICT) {
        alpha = hist[(size_t)iter * HIST + 2];
    } else {
        const double pAp = block_sum_partials(pAp_partials, pAp_count, 1, lds);
        alpha = hist[(size_t)(iter - 1) * HIST + 1] / pAp;
        if (blockIdx.x == 0 && threadIdx.x == 0) hist[(size_t)iter * HIST + 2] = alpha;
    }
    const double nalpha = -alpha;
    double       rr = 0.0, l1 = 0.0;
    const int64_t pairs = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * CGH_BLOCK + threadIdx.x; k < pairs; k += (int64_t)gridDim.x * CGH_BLOCK) {
        const double2 xv = reinterpret_cast<const double2 *>(x)[k];
        const double2 pv = reinterpret_cast<const double2 *>(p)[k];
        const double2 rv = reinterpret_cast<const double2 *>(r)[k];
        const double2 av = reinterpret_cast<const double2 *>(Ap)[k];
        // reference vec_axpy: out = a + alpha*b (vec.c:88-94), called with alpha and -alpha (cg.c:467,469)
        const double2 xn = make_double2(madd_rn(xv.x, alpha, pv.x), madd_rn(xv.y, alpha, pv.y));
        const double2 rn = make_double2(madd_rn(rv.x, nalpha, av.x), madd_rn(rv.y, nalpha, av.y));
        reinterpret_cast<double2 *>(x)[k] = xn;
        reinterpret_cast<double2 *>(r)[k] = rn;
        if constexpr (!STRICT) {
            rr += rn.x * rn.x + rn.y * rn.y;
            l1 += fabs(rn.x) + fabs(rn.y);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double  xn = madd_rn(x[i], alpha, p[i]);
        const double  rn = madd_rn(r[i], nalpha, Ap[i]);
        x[i] = xn;
        r[i] = rn;
        if constexpr (!STRICT) {
            rr += rn * rn;
            l1 += fabs(rn);
        }
    }
    if constexpr (!STRICT) {
        const double trr = block_sum_256(rr, lds);
        const double tl1 = block_sum_256(l1, lds);
        if (threadIdx.x == 0) {
            out_partials[2 * blockIdx.x] = trr;
            out_partials[2 * blockIdx.x + 1] = tl1;
        }
    }
}

// K3.
template <bool STRICT>
__global__ __launch_bounds__(CGH_BLOCK) void cg_update_p(int64_t n, const double *__restrict__ r, double *__restrict__ p,
                                                        const double *__restrict__ partials, int count, double *hist,
                                                        int iter, double tol, int *done)
{
    __shared__ double lds[4];
    if (*done != 0) return;
    double beta;
    bool   conv;
    if constexpr (STRICT) {
        beta = hist[(size_t)iter * HIST + 3];
        conv = hist[(size_t)iter * HIST + 0] < tol;
    } else {
        const double rr = block_sum_partials(partials, count, 2, lds);
        const double l1 = block_sum_partials(partials + 1, count, 2, lds);
        const double res = l1 / 2.0;  // norm_factor = 2.0, cg.c:427,475
        conv = res < tol;
        beta = rr / hist[(size_t)(iter - 1) * HIST + 1];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            hist[(size_t)iter * HIST + 0] = res;
            hist[(size_t)iter * HIST + 1] = rr;
            hist[(size_t)iter * HIST + 3] = conv ? NAN : beta;
        }
    }
    if (conv) {
        // the reference returns before touching p (cg.c:475-485); later iterations see the flag and do nothing.
        // Every workgroup reaches the same verdict from the same partials; workgroup 0 publishes it, and no
        // workgroup of THIS launch reads it back.
        if (STRICT == false && blockIdx.x == 0 && threadIdx.x == 0) *done = iter;
        return;
    }
    const int64_t pairs = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * CGH_BLOCK + threadIdx.x; k < pairs; k += (int64_t)gridDim.x * CGH_BLOCK) {
        const double2 rv = reinterpret_cast<const double2 *>(r)[k];
        const double2 pv = reinterpret_cast<const double2 *>(p)[k];
        reinterpret_cast<double2 *>(p)[k] = make_double2(madd_rn(rv.x, beta, pv.x), madd_rn(rv.y, beta, pv.y));
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = madd_rn(r[n - 1], beta, p[n - 1]);
}

// ---------------------------------------------------------------- strict (sequential-order) sums
//
// One workgroup.  All 256 threads stage rounded products a[i]*b[i] (and |c[i]|) of a 2048-element
// chunk into LDS; then ONE lane adds them left to right, exactly as reference vec_dot / vec_l1norm
// do (vec.c:67-85).  mode 0: out = {dot}; mode 1 (CG step A): pAp -> alpha = rr_prev/pAp;
// mode 2 (CG step B): rr, l1 -> res, beta, done flag; mode 3 (init): rr, l1 -> hist[0].
constexpr int SEQ_CHUNK = 2048;

__global__ __launch_bounds__(CGH_BLOCK) void seq_sums(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                     const double *__restrict__ c, int mode, double *hist, int iter,
                                                     double tol, int *done, double *out)
{
    __shared__ double prod[SEQ_CHUNK];
    __shared__ double absv[SEQ_CHUNK];
    if (done != nullptr && mode != 3 && mode != 0 && *done != 0) return;
    double s_dot = 0.0, s_l1 = 0.0;
    for (int64_t base = 0; base < n; base += SEQ_CHUNK) {
        const int cnt = (int)((n - base) < SEQ_CHUNK ? (n - base) : SEQ_CHUNK);
        for (int j = threadIdx.x; j < cnt; j += CGH_BLOCK) {
            if (a != nullptr) prod[j] = mul_rn(a[base + j], b[base + j]);
            if (c != nullptr) absv[j] = fabs(c[base + j]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (a != nullptr)
                for (int j = 0; j < cnt; ++j) s_dot = add_rn(s_dot, prod[j]);
        }
        if (threadIdx.x == 64) {  // a second wave runs the second chain concurrently
            if (c != nullptr)
                for (int j = 0; j < cnt; ++j) s_l1 = add_rn(s_l1, absv[j]);
        }
        __syncthreads();
    }
    // hand s_l1 (wave 1) to thread 0
    if (threadIdx.x == 64) absv[0] = s_l1;
    __syncthreads();
    if (threadIdx.x != 0) return;
    s_l1 = absv[0];
    if (mode == 0) {
        out[0] = s_dot;
        out[1] = s_l1;
    } else if (mode == 1) {
        hist[(size_t)iter * HIST + 2] = hist[(size_t)(iter - 1) * HIST + 1] / s_dot;
    } else if (mode == 2) {
        const double res = s_l1 / 2.0;
        const bool   conv = res < tol;
        hist[(size_t)iter * HIST + 0] = res;
        hist[(size_t)iter * HIST + 1] = s_dot;
        hist[(size_t)iter * HIST + 3] = conv ? NAN : s_dot / hist[(size_t)(iter - 1) * HIST + 1];
        if (conv) *done = iter;
    } else {
        hist[0] = s_l1 / 2.0;
        hist[1] = s_dot;
        hist[2] = NAN;
        hist[3] = NAN;
    }
}

// ---------------------------------------------------------------- stand-alone BLAS-1
__global__ __launch_bounds__(CGH_BLOCK) void dot_l1_partials(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                            const double *__restrict__ c, double *__restrict__ partials)
{
    __shared__ double lds[4];
    double            d = 0.0, l = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CGH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CGH_BLOCK) {
        if (a != nullptr) d += a[i] * b[i];
        if (c != nullptr) l += fabs(c[i]);
    }
    const double td = block_sum_256(d, lds);
    const double tl = block_sum_256(l, lds);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = td;
        partials[2 * blockIdx.x + 1] = tl;
    }
}

__global__ __launch_bounds__(CGH_BLOCK) void finish_pair(const double *__restrict__ partials, int count, double *out)
{
    __shared__ double lds[4];
    const double      d = block_sum_partials(partials, count, 2, lds);
    const double      l = block_sum_partials(partials + 1, count, 2, lds);
    if (threadIdx.x == 0) {
        out[0] = d;
        out[1] = l;
    }
}

__global__ __launch_bounds__(CGH_BLOCK) void axpy_kernel(int64_t n, const double *a, const double *b, double alpha, double *out)
{
    // out may alias a or b (reference vec_axpy is called both ways): element-wise, so no __restrict__
    for (int64_t i = (int64_t)blockIdx.x * CGH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CGH_BLOCK)
        out[i] = madd_rn(a[i], alpha, b[i]);
}

struct scratch_buf {
    void *p = nullptr;
    explicit scratch_buf(size_t bytes) { p = cgh_alloc(bytes); }
    ~scratch_buf() { cgh_free(p); }
    scratch_buf(const scratch_buf &) = delete;
    scratch_buf &operator=(const scratch_buf &) = delete;
};

int vec_grid(int64_t n)
{
    return cgh_stream_grid((n + 2 * CGH_BLOCK - 1) / (2 * CGH_BLOCK), 8);
}

int reduce_pair(const double *a, const double *b, const double *c, int64_t n, int strict, double *h_out2)
{
    scratch_buf parts(sizeof(double) * 2 * CGH_MAX_PARTIALS), out(sizeof(double) * 2);
    if (!parts.p || !out.p) return CGH_ENOMEM;
    hipStream_t s = g_cgh.stream;
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

int cgh_cg_solve(const cgh_matrix *m, const double *d_b, double *d_x, const cgh_cg_opts *opts, cgh_cg_report *rep,
                 double *h_history)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!m || !d_b || !d_x || !opts || !rep) {
        cgh_set_error("cgh_cg_solve: NULL argument");
        return CGH_EINVAL;
    }
    const int64_t n = m->n;
    const bool    strict = opts->strict != 0;
    const bool    fixed = opts->fixed_iters > 0;
    const int     max_iter = fixed ? opts->fixed_iters + 1 : opts->max_iter;
    const double  tol = fixed ? -1.0 : opts->tol;  // l1/2 >= 0 is never < -1: no convergence in benchmark mode
    int           check = opts->check_every > 0 ? opts->check_every : 32;
    memset(rep, 0, sizeof *rep);
    rep->ret = -1;
    if (n <= 0 || max_iter < 1) {
        cgh_set_error("cgh_cg_solve: n=%lld max_iter=%d", (long long)n, max_iter);
        return CGH_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(d_b) | reinterpret_cast<uintptr_t>(d_x)) & 15u) {
        cgh_set_error("cgh_cg_solve: d_b and d_x must be 16-byte aligned");
        return CGH_EINVAL;
    }

    hipStream_t s = g_cgh.stream;
    const size_t vbytes = sizeof(double) * (size_t)n;
    scratch_buf  r(vbytes), p(vbytes), Ap(vbytes);
    scratch_buf  partA(sizeof(double) * CGH_MAX_PARTIALS), partB(sizeof(double) * 2 * CGH_MAX_PARTIALS);
    scratch_buf  hist(sizeof(double) * HIST * (size_t)(max_iter + 1)), done(sizeof(int) * 4);
    if (!r.p || !p.p || !Ap.p || !partA.p || !partB.p || !hist.p || !done.p) return CGH_ENOMEM;
    double *dr = (double *)r.p, *dp = (double *)p.p, *dAp = (double *)Ap.p;
    double *dA = (double *)partA.p, *dB = (double *)partB.p, *dh = (double *)hist.p;
    int    *dd = (int *)done.p;
    const int vg = vec_grid(n);

    // ---- setup: x = 0; r = b - A x; p = r; rr = r.r (cg.c:433-456)
    CGH_HIP(hipMemsetAsync(dd, 0, sizeof(int) * 4, s));
    CGH_HIP(hipMemsetAsync(d_x, 0, vbytes, s));
    int rc = cgh_launch_spmv(m, d_x, dAp, nullptr, nullptr, nullptr, s);
    if (rc) return rc;
    hipLaunchKernelGGL(cg_init_rp, dim3(vg), dim3(CGH_BLOCK), 0, s, n, d_b, (const double *)dAp, dr, dp, dB);
    if (strict)
        hipLaunchKernelGGL(seq_sums, dim3(1), dim3(CGH_BLOCK), 0, s, n, (const double *)dr, (const double *)dr,
                           (const double *)dr, 3, dh, 0, tol, dd, (double *)nullptr);
    else
        hipLaunchKernelGGL(cg_finish_init, dim3(1), dim3(CGH_BLOCK), 0, s, (const double *)dB, vg, dh);
    CGH_HIP(hipGetLastError());
    CGH_HIP(hipMemcpyAsync(rep->r0, dr, sizeof(double) * (n < 5 ? n : 5), hipMemcpyDeviceToHost, s));

    // ---- iterations
    hipEvent_t ev0, ev1;
    CGH_HIP(hipEventCreate(&ev0));
    CGH_HIP(hipEventCreate(&ev1));
    std::vector<hipEvent_t> kev;
    const bool              timek = opts->time_kernels != 0;
    CGH_HIP(hipEventRecord(ev0, s));
    int done_iter = 0, launched = 0;
    for (int iter = 1; iter < max_iter && done_iter == 0; ++iter) {
        int gridA = 0;
        if (timek) {
            hipEvent_t a, b;
            CGH_HIP(hipEventCreate(&a));
            CGH_HIP(hipEventCreate(&b));
            kev.push_back(a);
            kev.push_back(b);
            CGH_HIP(hipEventRecord(a, s));
        }
        rc = cgh_launch_spmv(m, dp, dAp, strict ? nullptr : dA, &gridA, dd, s);
        if (rc) return rc;
        if (timek) CGH_HIP(hipEventRecord(kev.back(), s));
        if (strict) {
            hipLaunchKernelGGL(seq_sums, dim3(1), dim3(CGH_BLOCK), 0, s, n, (const double *)dp, (const double *)dAp,
                               (const double *)nullptr, 1, dh, iter, tol, dd, (double *)nullptr);
            hipLaunchKernelGGL((cg_update_xr<true>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, d_x, (const double *)dp, dr,
                               (const double *)dAp, (const double *)dA, gridA, dB, dh, iter, (const int *)dd);
            hipLaunchKernelGGL(seq_sums, dim3(1), dim3(CGH_BLOCK), 0, s, n, (const double *)dr, (const double *)dr,
                               (const double *)dr, 2, dh, iter, tol, dd, (double *)nullptr);
            hipLaunchKernelGGL((cg_update_p<true>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, (const double *)dr, dp,
                               (const double *)dB, vg, dh, iter, tol, dd);
        } else {
            hipLaunchKernelGGL((cg_update_xr<false>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, d_x, (const double *)dp, dr,
                               (const double *)dAp, (const double *)dA, gridA, dB, dh, iter, (const int *)dd);
            hipLaunchKernelGGL((cg_update_p<false>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, (const double *)dr, dp,
                               (const double *)dB, vg, dh, iter, tol, dd);
        }
        launched = iter;
        if (!fixed && (iter % check == 0 || iter == max_iter - 1)) {
            CGH_HIP(hipGetLastError());
            CGH_HIP(hipMemcpyAsync(&done_iter, dd, sizeof(int), hipMemcpyDeviceToHost, s));
            CGH_HIP(hipStreamSynchronize(s));
        }
    }
    CGH_HIP(hipGetLastError());
    CGH_HIP(hipEventRecord(ev1, s));
    CGH_HIP(hipMemcpyAsync(&done_iter, dd, sizeof(int), hipMemcpyDeviceToHost, s));
    CGH_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CGH_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    rep->solve_ms = ms;
    if (timek) {
        double tot = 0.0;
        for (size_t k = 0; k + 1 < kev.size(); k += 2) {
            float t = 0.f;
            CGH_HIP(hipEventElapsedTime(&t, kev[k], kev[k + 1]));
            tot += t;
        }
        rep->spmv_ms_avg = kev.empty() ? 0.0 : tot / (double)(kev.size() / 2);
        for (hipEvent_t e : kev) (void)hipEventDestroy(e);
    }
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);

    const int executed = done_iter ? done_iter : launched;
    rep->iterations = executed;
    rep->ret = done_iter ? done_iter + 1 : -1;
    std::vector<double> hh((size_t)HIST * (executed + 1));
    CGH_HIP(hipMemcpyAsync(hh.data(), dh, hh.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    CGH_HIP(hipStreamSynchronize(s));
    rep->initial_res = hh[0];
    rep->final_res = hh[(size_t)HIST * executed];
    if (h_history) memcpy(h_history, hh.data(), hh.size() * sizeof(double));
    return CGH_OK;
}

} // extern "C"
