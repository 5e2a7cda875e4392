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

typedef double d2_t __attribute__((ext_vector_type(2)));

// Streaming accesses: every vector here is far larger than L2 and touched once per kernel, so loads and
// stores are non-temporal (measured +15 % on the SpMV's y store, profiles/).  VV bit0 = nt loads, bit1 = nt stores.
template <int VV>
__device__ __forceinline__ double2 ld2(const double *p, int64_t k)
{
    if constexpr (VV & 1) {
        const d2_t v = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p) + k);
        return make_double2(v.x, v.y);
    } else {
        return reinterpret_cast<const double2 *>(p)[k];
    }
}
template <int VV>
__device__ __forceinline__ void st2(double *p, int64_t k, double a, double b)
{
    if constexpr (VV & 2) {
        d2_t v;
        v.x = a;
        v.y = b;
        __builtin_nontemporal_store(v, reinterpret_cast<d2_t *>(p) + k);
    } else {
        reinterpret_cast<double2 *>(p)[k] = make_double2(a, b);
    }
}

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
template <bool STRICT, int VV>
__global__ __launch_bounds__(CGH_BLOCK) void cg_update_xr(int64_t n, double *__restrict__ x, const double *__restrict__ p,
                                                         double *__restrict__ r, const double *__restrict__ Ap,
                                                         const double *__restrict__ pAp_partials, int pAp_count,
                                                         double *__restrict__ out_partials, double *hist, int iter,
                                                         const int *__restrict__ done, int rev)
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
    const int64_t chunks = (pairs + CGH_BLOCK - 1) / CGH_BLOCK;
    for (int64_t cc = blockIdx.x; cc < chunks; cc += gridDim.x) {
        const int64_t k = (rev ? chunks - 1 - cc : cc) * CGH_BLOCK + threadIdx.x;
        if (k >= pairs) continue;
        const double2 xv = ld2<VV>(x, k);
        const double2 pv = ld2<VV>(p, k);
        const double2 rv = ld2<VV>(r, k);
        const double2 av = ld2<VV>(Ap, k);
        // reference vec_axpy: out = a + alpha*b (vec.c:88-94), called with alpha and -alpha (cg.c:467,469)
        const double2 xn = make_double2(madd_rn(xv.x, alpha, pv.x), madd_rn(xv.y, alpha, pv.y));
        const double2 rn = make_double2(madd_rn(rv.x, nalpha, av.x), madd_rn(rv.y, nalpha, av.y));
        st2<VV>(x, k, xn.x, xn.y);
        st2<VV>(r, k, rn.x, rn.y);
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
template <bool STRICT, int VV>
__global__ __launch_bounds__(CGH_BLOCK) void cg_update_p(int64_t n, const double *__restrict__ r, double *__restrict__ p,
                                                        const double *__restrict__ partials, int count, double *hist,
                                                        int iter, double tol, int *done, int rev)
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
    const int64_t chunks = (pairs + CGH_BLOCK - 1) / CGH_BLOCK;
    for (int64_t cc = blockIdx.x; cc < chunks; cc += gridDim.x) {
        const int64_t k = (rev ? chunks - 1 - cc : cc) * CGH_BLOCK + threadIdx.x;
        if (k >= pairs) continue;
        const double2 rv = ld2<VV>(r, k);
        const double2 pv = ld2<VV>(p, k);
        st2<VV>(p, k, madd_rn(rv.x, beta, pv.x), madd_rn(rv.y, beta, pv.y));
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

struct cgh_solver {
    const cgh_matrix *m = nullptr;
    int64_t           n = 0;
    int               capacity = 0;   // iterations the history can hold
    int               queued = 0;     // iterations queued since start
    int               vgrid = 0;
    bool              strict = false;
    double            tol = 0.0;
    double           *x = nullptr;    // caller's solution vector
    double           *r = nullptr, *p = nullptr, *Ap = nullptr;
    double           *partA = nullptr, *partB = nullptr, *hist = nullptr;
    int              *done = nullptr;
    std::vector<hipEvent_t> kev;      // SpMV kernel brackets (time_kernels)
};

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
    cgh_free(sv->r);
    cgh_free(sv->p);
    cgh_free(sv->Ap);
    cgh_free(sv->partA);
    cgh_free(sv->partB);
    cgh_free(sv->hist);
    cgh_free(sv->done);
    delete sv;
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
    hipStream_t   s = g_cgh.stream;
    const int64_t n = sv->n;
    sv->x = d_x;
    sv->tol = tol;
    sv->strict = strict != 0;
    sv->queued = 0;
    for (hipEvent_t e : sv->kev) (void)hipEventDestroy(e);
    sv->kev.clear();
    CGH_HIP(hipMemsetAsync(sv->done, 0, sizeof(int) * 4, s));
    CGH_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)n, s));
    int rc = cgh_launch_spmv(sv->m, d_x, sv->Ap, nullptr, nullptr, nullptr, s);
    if (rc) return rc;
    hipLaunchKernelGGL(cg_init_rp, dim3(sv->vgrid), dim3(CGH_BLOCK), 0, s, n, d_b, (const double *)sv->Ap, sv->r, sv->p,
                       sv->partB);
    if (sv->strict)
        hipLaunchKernelGGL(seq_sums, dim3(1), dim3(CGH_BLOCK), 0, s, n, (const double *)sv->r, (const double *)sv->r,
                           (const double *)sv->r, 3, sv->hist, 0, tol, sv->done, (double *)nullptr);
    else
        hipLaunchKernelGGL(cg_finish_init, dim3(1), dim3(CGH_BLOCK), 0, s, (const double *)sv->partB, sv->vgrid, sv->hist);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

// Queue `count` more iterations on the compute stream (no host synchronisation).
int cgh_solver_iterate(cgh_solver *sv, int count, int time_kernels)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!sv || count < 0 || sv->queued + count > sv->capacity) {
        cgh_set_error("cgh_solver_iterate: count %d exceeds the solver's capacity", count);
        return CGH_EINVAL;
    }
    hipStream_t   s = g_cgh.stream;
    const int64_t n = sv->n;
    const int     vg = sv->vgrid;
    double       *dr = sv->r, *dp = sv->p, *dAp = sv->Ap, *dA = sv->partA, *dB = sv->partB, *dh = sv->hist, *d_x = sv->x;
    int          *dd = sv->done;
    const double  tol = sv->tol;
    for (int c = 0; c < count; ++c) {
        const int iter = ++sv->queued;
        int       gridA = 0;
        // serpentine sweeps: each kernel starts where the previous one ended, so the tail of what was just
        // written/read could still be in the 256 MiB Infinity Cache.  Measured null on MI355X (0.4791 vs 0.4797
        // ms/iter, 256^3), so off by default; env CGH_SERPENTINE=1 enables it for experiments.
        static const int serp = getenv("CGH_SERPENTINE") ? atoi(getenv("CGH_SERPENTINE")) : 0;
        const int        rev1 = serp ? (iter & 1) : 0, rev2 = serp ? !rev1 : 0;
        g_cgh.sweep_reverse = rev1;
        if (time_kernels) {
            hipEvent_t a, b;
            CGH_HIP(hipEventCreate(&a));
            CGH_HIP(hipEventCreate(&b));
            sv->kev.push_back(a);
            sv->kev.push_back(b);
            CGH_HIP(hipEventRecord(a, s));
        }
        int rc = cgh_launch_spmv(sv->m, dp, dAp, sv->strict ? nullptr : dA, &gridA, dd, s);
        if (rc) return rc;
        g_cgh.sweep_reverse = 0;
        if (time_kernels) CGH_HIP(hipEventRecord(sv->kev.back(), s));
        if (sv->strict) {
            hipLaunchKernelGGL(seq_sums, dim3(1), dim3(CGH_BLOCK), 0, s, n, (const double *)dp, (const double *)dAp,
                               (const double *)nullptr, 1, dh, iter, tol, dd, (double *)nullptr);
            hipLaunchKernelGGL((cg_update_xr<true, 3>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, d_x, (const double *)dp, dr,
                               (const double *)dAp, (const double *)dA, gridA, dB, dh, iter, (const int *)dd, rev2);
            hipLaunchKernelGGL(seq_sums, dim3(1), dim3(CGH_BLOCK), 0, s, n, (const double *)dr, (const double *)dr,
                               (const double *)dr, 2, dh, iter, tol, dd, (double *)nullptr);
            hipLaunchKernelGGL((cg_update_p<true, 3>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, (const double *)dr, dp,
                               (const double *)dB, vg, dh, iter, tol, dd, rev1);
        } else {
            static const int vv = getenv("CGH_VEC_VARIANT") ? atoi(getenv("CGH_VEC_VARIANT")) & 3 : 3;
#define CGH_VEC_VAR(V)                                                                                                 \
    case V:                                                                                                            \
        hipLaunchKernelGGL((cg_update_xr<false, V>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, d_x, (const double *)dp, dr,  \
                           (const double *)dAp, (const double *)dA, gridA, dB, dh, iter, (const int *)dd, rev2);             \
        hipLaunchKernelGGL((cg_update_p<false, V>), dim3(vg), dim3(CGH_BLOCK), 0, s, n, (const double *)dr, dp,        \
                           (const double *)dB, vg, dh, iter, tol, dd, rev1);                                                 \
        break;
            switch (vv) {
                CGH_VEC_VAR(0)
                CGH_VEC_VAR(1)
                CGH_VEC_VAR(2)
                CGH_VEC_VAR(3)
            }
#undef CGH_VEC_VAR
        }
    }
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

// Drain the stream and read the device's verdict: *done_iter = converging iteration (0 = not yet).
int cgh_solver_poll(cgh_solver *sv, int *done_iter, int *queued)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!sv) return CGH_EINVAL;
    int d = 0;
    CGH_HIP(hipMemcpyAsync(&d, sv->done, sizeof(int), hipMemcpyDeviceToHost, g_cgh.stream));
    CGH_HIP(hipStreamSynchronize(g_cgh.stream));
    if (done_iter) *done_iter = d;
    if (queued) *queued = sv->queued;
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

int cgh_cg_solve(const cgh_matrix *m, const double *d_b, double *d_x, const cgh_cg_opts *opts, cgh_cg_report *rep,
                 double *h_history)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!m || !d_b || !d_x || !opts || !rep) {
        cgh_set_error("cgh_cg_solve: NULL argument");
        return CGH_EINVAL;
    }
    const bool   fixed = opts->fixed_iters > 0;
    const int    max_iter = fixed ? opts->fixed_iters + 1 : opts->max_iter;
    const double tol = fixed ? -1.0 : opts->tol;  // l1/2 >= 0 is never < -1: no convergence in benchmark mode
    const int    check = opts->check_every > 0 ? opts->check_every : 32;
    memset(rep, 0, sizeof *rep);
    rep->ret = -1;
    if (m->n <= 0 || max_iter < 1) {
        cgh_set_error("cgh_cg_solve: n=%lld max_iter=%d", (long long)m->n, max_iter);
        return CGH_EINVAL;
    }
    cgh_solver *sv = cgh_solver_create(m, max_iter);
    if (!sv) return CGH_ENOMEM;
    struct guard {
        cgh_solver *p;
        ~guard() { cgh_solver_free(p); }
    } g{sv};
    hipStream_t s = g_cgh.stream;
    int         rc = cgh_solver_start(sv, d_b, d_x, tol, opts->strict);
    if (rc) return rc;
    CGH_HIP(hipMemcpyAsync(rep->r0, sv->r, sizeof(double) * (m->n < 5 ? m->n : 5), hipMemcpyDeviceToHost, s));

    hipEvent_t ev0, ev1;
    CGH_HIP(hipEventCreate(&ev0));
    CGH_HIP(hipEventCreate(&ev1));
    CGH_HIP(hipEventRecord(ev0, s));
    int done_iter = 0;
    const int total = max_iter - 1;  // reference loop: iter = 1 .. max_iter-1 (cg.c:459)
    while (sv->queued < total && done_iter == 0) {
        int chunk = total - sv->queued;
        if (!fixed && chunk > check) chunk = check;
        rc = cgh_solver_iterate(sv, chunk, opts->time_kernels);
        if (rc) return rc;
        if (!fixed) {
            rc = cgh_solver_poll(sv, &done_iter, nullptr);
            if (rc) return rc;
        }
    }
    CGH_HIP(hipEventRecord(ev1, s));
    rc = cgh_solver_poll(sv, &done_iter, nullptr);
    if (rc) return rc;
    float ms = 0.f;
    CGH_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    rep->solve_ms = ms;
    if (opts->time_kernels) cgh_solver_spmv_ms(sv, &rep->spmv_ms_avg);

    const int executed = done_iter ? done_iter : sv->queued;
    rep->iterations = executed;
    rep->ret = done_iter ? done_iter + 1 : -1;
    std::vector<double> hh((size_t)HIST * (executed + 1));
    rc = cgh_solver_history(sv, 0, executed + 1, hh.data());
    if (rc) return rc;
    rep->initial_res = hh[0];
    rep->final_res = hh[(size_t)HIST * executed];
    if (h_history) memcpy(h_history, hh.data(), hh.size() * sizeof(double));
    return CGH_OK;
}

} // extern "C"
