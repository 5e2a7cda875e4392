// cg_kernels.h -- device kernels of the CG iteration (shared by cg.hip and dist.hip; see cg.hip for the design).
#pragma once
#include <cmath>

#include "cgh_internal.h"
#include "device_utils.h"

namespace {

constexpr int HIST = CGH_HIST;  // doubles per history entry: res, rr, alpha, beta

typedef double d2_t __attribute__((ext_vector_type(2)));

// Streaming accesses: every vector here is far larger than L2 and touched once per kernel, so loads and
// stores are non-temporal (measured +15 % on the SpMV's y store, profiles/).  VV bit0 = nt loads, bit1 = nt stores
// of x, bit2 = nt stores of r, bit3 = nt stores of p.  Shipped: 3 -- x is not touched again until the next iteration,
// while r (re-read by cg_update_xp) and p (re-read by the SpMV) are left to the caches: 0.4713 vs 0.4787 ms/iteration
// with everything non-temporal (256^3, MI355X).
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
template <bool NT>
__device__ __forceinline__ void st2(double *p, int64_t k, double a, double b)
{
    if constexpr (NT) {
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

// K2.  r -= alpha Ap; partials of r.r and |r|_1.  (x += alpha p is done by K3, which reads p anyway: one sweep over p
// less per iteration, same arithmetic, same bits.)
// MODE 0: alpha from this GPU's p.Ap partials; 1: alpha already in the history (strict sums); 2: p.Ap reduced
// over all ranks waits in gs[0] (multi-GPU).
template <int MODE, int VV>
__global__ __launch_bounds__(CGH_BLOCK) void cg_update_r(int64_t n, double *__restrict__ r, const double *__restrict__ Ap,
                                                        const double *__restrict__ pAp_partials, int pAp_count,
                                                        double *__restrict__ out_partials, double *hist, int iter,
                                                        const int *__restrict__ done, int rev, const double *__restrict__ gs,
                                                        double *__restrict__ cellsum = nullptr, int64_t ncells = 0,
                                                        double *__restrict__ groupsum = nullptr)
{
    constexpr bool STRICT = MODE == 1;
    __shared__ double lds[8];
    if (*done != 0) {
        // frozen.  A replayed (graph) launch tells the cg_update_xp behind it that there is no iteration in flight.
        if (iter == 0 && blockIdx.x == 0 && threadIdx.x == 0) const_cast<int *>(done)[1] = 0;
        return;
    }
    // iter == 0: take the iteration number from the device (done[2] = last iteration finished by cg_update_xp), so that
    // identical launches can be replayed from a HIP graph; done[1] tells cg_update_xp which iteration this is
    if (iter == 0) iter = done[2] + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<int *>(done)[1] = iter;
    double alpha;
    if constexpr (MODE == 1) {
        alpha = hist[(size_t)iter * HIST + 2];
    } else if constexpr (MODE == 2) {
        alpha = hist[(size_t)(iter - 1) * HIST + 1] / gs[0];
        if (blockIdx.x == 0 && threadIdx.x == 0) hist[(size_t)iter * HIST + 2] = alpha;
    } else {
        const double pAp = block_sum_partials(pAp_partials, pAp_count, 1, lds);
        alpha = hist[(size_t)(iter - 1) * HIST + 1] / pAp;
        if (blockIdx.x == 0 && threadIdx.x == 0) hist[(size_t)iter * HIST + 2] = alpha;
    }
    const double nalpha = -alpha;
    double       rr = 0.0, l1 = 0.0;
    const int64_t pairs = n >> 1;
    const int64_t chunks = (pairs + CGH_BLOCK - 1) / CGH_BLOCK;
    if constexpr (STRICT) {
        if (cellsum != nullptr && groupsum != nullptr) {
            // Strict mode with the order-exact sums (xsum.hip): a workgroup takes whole GROUPS of their hierarchy (64 cells =
            // 8192 rows = 16 chunks, one after the other) and leaves, beside every cell's approximate r.r and |r|_1, the
            // group's -- the sums' launch then needs no pass of its own over the cell sums (xs_group_sums).
            const int64_t ngroups = (ncells + 63) / 64;
            const int     wave = threadIdx.x >> 6;
            for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
                double gq = 0.0, ga = 0.0;
#pragma unroll 4
                for (int c16 = 0; c16 < 16; ++c16) {
                    const int64_t k = (g * 16 + c16) * CGH_BLOCK + threadIdx.x;
                    double2       rn = make_double2(0.0, 0.0);
                    if (k < pairs) {
                        const double2 rv = ld2<VV>(r, k);
                        const double2 av = ld2<VV>(Ap, k);
                        rn = make_double2(madd_rn(rv.x, nalpha, av.x), madd_rn(rv.y, nalpha, av.y));
                        st2<(VV & 4) != 0>(r, k, rn.x, rn.y);
                    }
                    const double  q = wave_sum_lane63(rn.x * rn.x + rn.y * rn.y), a = wave_sum_lane63(fabs(rn.x) + fabs(rn.y));
                    const int64_t cell = k >> 6;
                    if ((threadIdx.x & 63) == 63) {
                        if (cell < ncells) {
                            cellsum[cell] = q;
                            cellsum[ncells + cell] = a;
                        }
                        gq += q;
                        ga += a;
                    }
                }
                if ((threadIdx.x & 63) == 63) {
                    lds[wave] = gq;
                    lds[4 + wave] = ga;
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    groupsum[g] = ((lds[0] + lds[1]) + lds[2]) + lds[3];
                    groupsum[ngroups + g] = ((lds[4] + lds[5]) + lds[6]) + lds[7];
                }
                __syncthreads();
            }
            if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) r[n - 1] = madd_rn(r[n - 1], nalpha, Ap[n - 1]);
            return;
        }
    }
    for (int64_t cc = blockIdx.x; cc < chunks; cc += gridDim.x) {
        const int64_t k = (rev ? chunks - 1 - cc : cc) * CGH_BLOCK + threadIdx.x;
        double2       rn = make_double2(0.0, 0.0);
        if (k < pairs) {
            const double2 rv = ld2<VV>(r, k);
            const double2 av = ld2<VV>(Ap, k);
            // reference vec_axpy: out = a + alpha*b (vec.c:88-94), called with -alpha (cg.c:469)
            rn = make_double2(madd_rn(rv.x, nalpha, av.x), madd_rn(rv.y, nalpha, av.y));
            st2<(VV & 4) != 0>(r, k, rn.x, rn.y);
        }
        if constexpr (!STRICT) {
            rr += rn.x * rn.x + rn.y * rn.y;
            l1 += fabs(rn.x) + fabs(rn.y);
        } else if (cellsum != nullptr) {
            // A wave's 64 pairs are the 128 elements of one cell of the order-exact sums (xsum.hip): leave the cell's
            // approximate r.r and |r|_1 for their binade prediction, which saves them a sweep over r.  (An odd n's last
            // element is missing from the last cell's sum: the prediction is only a guess, the sums stay exact.)
            const double  q = wave_sum(rn.x * rn.x + rn.y * rn.y), a = wave_sum(fabs(rn.x) + fabs(rn.y));
            const int64_t cell = k >> 6;
            if ((threadIdx.x & 63) == 0 && cell < ncells) {
                cellsum[cell] = q;
                cellsum[ncells + cell] = a;
            }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double  rn = madd_rn(r[i], nalpha, Ap[i]);
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

// K3.  x += alpha p (alpha of this iteration, from the history); then the convergence test, beta, p = r + beta p.
// x == NULL: x has been updated already (strict mode: by the launch of the sums of r, xsum.hip).
template <int MODE, int VV>
__global__ __launch_bounds__(CGH_BLOCK) void cg_update_xp(int64_t n, double *__restrict__ x, const double *__restrict__ r,
                                                         double *__restrict__ p, const double *__restrict__ partials, int count,
                                                         double *hist, int iter, double tol, int *done, int rev,
                                                         const double *__restrict__ gs)
{
    constexpr bool STRICT = MODE == 1;
    __shared__ double lds[4];
    if (iter == 0) {
        iter = done[1];
        if (iter == 0) return;  // replayed launch behind a frozen cg_update_r
    }
    // The flag may already name THIS iteration: set by the strict sums just before this launch, or by workgroup 0 of
    // this very launch (below) -- either way the converging iteration still owes its x update.
    const int frozen_at = *done;
    if (frozen_at != 0 && frozen_at != iter) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) done[2] = iter;
    const double alpha = hist[(size_t)iter * HIST + 2];
    double beta;
    bool   conv;
    if constexpr (MODE == 1) {
        beta = hist[(size_t)iter * HIST + 3];
        conv = hist[(size_t)iter * HIST + 0] < tol;
    } else {
        double rr, l1;
        if constexpr (MODE == 2) {
            rr = gs[1];
            l1 = gs[2];
        } else {
            rr = block_sum_partials(partials, count, 2, lds);
            l1 = block_sum_partials(partials + 1, count, 2, lds);
        }
        const double res = l1 / 2.0;  // norm_factor = 2.0, cg.c:427,475
        conv = res < tol;
        beta = rr / hist[(size_t)(iter - 1) * HIST + 1];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            hist[(size_t)iter * HIST + 0] = res;
            hist[(size_t)iter * HIST + 1] = rr;
            hist[(size_t)iter * HIST + 3] = conv ? NAN : beta;
        }
    }
    const int64_t pairs = n >> 1;
    const int64_t chunks = (pairs + CGH_BLOCK - 1) / CGH_BLOCK;
    if (conv) {
        // the reference updates x, then returns before touching p (cg.c:467-485); later iterations see the flag and do
        // nothing.  Every workgroup reaches the same verdict from the same partials; workgroup 0 publishes it, and no
        // workgroup of THIS launch reads it back.
        if (STRICT == false && blockIdx.x == 0 && threadIdx.x == 0) *done = iter;
        if (x == nullptr) return;
        for (int64_t cc = blockIdx.x; cc < chunks; cc += gridDim.x) {
            const int64_t k = cc * CGH_BLOCK + threadIdx.x;
            if (k >= pairs) continue;
            const double2 xv = ld2<VV>(x, k), pv = ld2<VV>(p, k);
            st2<(VV & 2) != 0>(x, k, madd_rn(xv.x, alpha, pv.x), madd_rn(xv.y, alpha, pv.y));
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[n - 1] = madd_rn(x[n - 1], alpha, p[n - 1]);
        return;
    }
    for (int64_t cc = blockIdx.x; cc < chunks; cc += gridDim.x) {
        const int64_t k = (rev ? chunks - 1 - cc : cc) * CGH_BLOCK + threadIdx.x;
        if (k >= pairs) continue;
        const double2 rv = ld2<VV>(r, k);
        const double2 pv = ld2<VV>(p, k);
        // reference vec_axpy: out = a + alpha*b (vec.c:88-94): x += alpha p (cg.c:467), p = r + beta p (cg.c:489)
        if (x != nullptr) {
            const double2 xv = ld2<VV>(x, k);
            st2<(VV & 2) != 0>(x, k, madd_rn(xv.x, alpha, pv.x), madd_rn(xv.y, alpha, pv.y));
        }
        st2<(VV & 8) != 0>(p, k, madd_rn(rv.x, beta, pv.x), madd_rn(rv.y, beta, pv.y));
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        if (x != nullptr) x[n - 1] = madd_rn(x[n - 1], alpha, p[n - 1]);
        p[n - 1] = madd_rn(r[n - 1], beta, p[n - 1]);
    }
}

// ---------------------------------------------------------------- Jacobi-preconditioned variant
//
// No reference counterpart (the reference is "nonePCG"); off by default.  z = r / diag is never stored: the three
// kernels divide on the fly.  Same structure, history layout and stop test (l1norm(r)/2 < tol) as the plain loop;
// history[.][1] holds r.z instead of r.r.  Throughput-mode reductions only.
__global__ __launch_bounds__(CGH_BLOCK) void pcg_init_rp(int64_t n, const double *__restrict__ b, const double *__restrict__ Ap0,
                                                        const double *__restrict__ dg, double *__restrict__ r,
                                                        double *__restrict__ p, double *__restrict__ partials)
{
    __shared__ double lds[4];
    double            rz = 0.0, l1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CGH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CGH_BLOCK) {
        const double v = b[i] - Ap0[i];
        const double z = v / dg[i];
        r[i] = v;
        p[i] = z;
        rz += v * z;
        l1 += fabs(v);
    }
    const double trz = block_sum_256(rz, lds);
    const double tl1 = block_sum_256(l1, lds);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = trz;
        partials[2 * blockIdx.x + 1] = tl1;
    }
}

__global__ __launch_bounds__(CGH_BLOCK) void pcg_update_r(int64_t n, double *__restrict__ r, const double *__restrict__ Ap,
                                                         const double *__restrict__ dg, const double *__restrict__ pAp_partials,
                                                         int pAp_count, double *__restrict__ out_partials, double *hist, int iter,
                                                         const int *__restrict__ done)
{
    __shared__ double lds[4];
    if (*done != 0) {
        if (iter == 0 && blockIdx.x == 0 && threadIdx.x == 0) const_cast<int *>(done)[1] = 0;
        return;
    }
    if (iter == 0) iter = done[2] + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<int *>(done)[1] = iter;
    const double pAp = block_sum_partials(pAp_partials, pAp_count, 1, lds);
    const double alpha = hist[(size_t)(iter - 1) * HIST + 1] / pAp;
    if (blockIdx.x == 0 && threadIdx.x == 0) hist[(size_t)iter * HIST + 2] = alpha;
    const double nalpha = -alpha;
    double       rz = 0.0, l1 = 0.0;
    const int64_t pairs = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * CGH_BLOCK + threadIdx.x; k < pairs; k += (int64_t)gridDim.x * CGH_BLOCK) {
        const double2 rv = ld2<1>(r, k), av = ld2<1>(Ap, k), dv = ld2<1>(dg, k);
        const double2 rn = make_double2(madd_rn(rv.x, nalpha, av.x), madd_rn(rv.y, nalpha, av.y));
        st2<false>(r, k, rn.x, rn.y);
        rz += rn.x * (rn.x / dv.x) + rn.y * (rn.y / dv.y);
        l1 += fabs(rn.x) + fabs(rn.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double  rn = madd_rn(r[i], nalpha, Ap[i]);
        r[i] = rn;
        rz += rn * (rn / dg[i]);
        l1 += fabs(rn);
    }
    const double trz = block_sum_256(rz, lds);
    const double tl1 = block_sum_256(l1, lds);
    if (threadIdx.x == 0) {
        out_partials[2 * blockIdx.x] = trz;
        out_partials[2 * blockIdx.x + 1] = tl1;
    }
}

__global__ __launch_bounds__(CGH_BLOCK) void pcg_update_xp(int64_t n, double *__restrict__ x, const double *__restrict__ r,
                                                          double *__restrict__ p, const double *__restrict__ dg,
                                                          const double *__restrict__ partials, int count, double *hist, int iter,
                                                          double tol, int *done)
{
    __shared__ double lds[4];
    if (iter == 0) {
        iter = done[1];
        if (iter == 0) return;
    }
    const int frozen_at = *done;
    if (frozen_at != 0 && frozen_at != iter) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) done[2] = iter;
    const double alpha = hist[(size_t)iter * HIST + 2];
    const double rz = block_sum_partials(partials, count, 2, lds);
    const double l1 = block_sum_partials(partials + 1, count, 2, lds);
    const double res = l1 / 2.0;
    const bool   conv = res < tol;
    const double beta = rz / hist[(size_t)(iter - 1) * HIST + 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hist[(size_t)iter * HIST + 0] = res;
        hist[(size_t)iter * HIST + 1] = rz;
        hist[(size_t)iter * HIST + 3] = conv ? NAN : beta;
        if (conv) *done = iter;
    }
    const int64_t pairs = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * CGH_BLOCK + threadIdx.x; k < pairs; k += (int64_t)gridDim.x * CGH_BLOCK) {
        const double2 xv = ld2<1>(x, k), pv = ld2<1>(p, k);
        st2<true>(x, k, madd_rn(xv.x, alpha, pv.x), madd_rn(xv.y, alpha, pv.y));
        if (!conv) {
            const double2 rv = ld2<1>(r, k), dv = ld2<1>(dg, k);
            st2<false>(p, k, madd_rn(rv.x / dv.x, beta, pv.x), madd_rn(rv.y / dv.y, beta, pv.y));
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        x[i] = madd_rn(x[i], alpha, p[i]);
        if (!conv) p[i] = madd_rn(r[i] / dg[i], beta, p[i]);
    }
}

// ---------------------------------------------------------------- strict (sequential-order) sums
//
// One workgroup.  All 256 threads stage rounded products a[i]*b[i] (and |c[i]|) of a 2048-element
// chunk into LDS; then ONE lane adds them left to right, exactly as reference vec_dot / vec_l1norm
// do (vec.c:67-85).  mode 0: out = {dot}; mode 1 (CG step A): pAp -> alpha = rr_prev/pAp;
// mode 2 (CG step B): rr, l1 -> res, beta, done flag; mode 3 (init): rr, l1 -> hist[0].
constexpr int SEQ_CHUNK = 2048;

// s = (((s + v[0]) + v[1]) + ...) + v[cnt-1], one lane.  The adds form one dependent chain; the LDS reads do not
// depend on it, so they are issued 16 at a time (ds_read_b128) ahead of the adds that consume them.
__device__ __forceinline__ double chain_add(double s, const double *v, int cnt)
{
    int j = 0;
    for (; j + 16 <= cnt; j += 16) {
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            const double2 w = *reinterpret_cast<const double2 *>(v + j + k);
            t[k] = w.x;
            t[k + 1] = w.y;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s = add_rn(s, t[k]);
    }
    for (; j < cnt; ++j) s = add_rn(s, v[j]);
    return s;
}

__global__ __launch_bounds__(CGH_BLOCK) void seq_sums(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                     const double *__restrict__ c, int mode, double *hist, int iter,
                                                     double tol, int *done, double *out)
{
    __shared__ __attribute__((aligned(16))) double prod[SEQ_CHUNK];
    __shared__ __attribute__((aligned(16))) double absv[SEQ_CHUNK];
    if (done != nullptr && mode != 3 && mode != 0 && *done != 0) return;
    if (iter == 0 && mode == 1) iter = done[2] + 1;   // before cg_update_r of this iteration
    if (iter == 0 && mode == 2) iter = done[1];       // after it
    double s_dot = 0.0, s_l1 = 0.0;
    for (int64_t base = 0; base < n; base += SEQ_CHUNK) {
        const int cnt = (int)((n - base) < SEQ_CHUNK ? (n - base) : SEQ_CHUNK);
        for (int j = threadIdx.x; j < cnt; j += CGH_BLOCK) {
            if (a != nullptr) prod[j] = mul_rn(a[base + j], b[base + j]);
            if (c != nullptr) absv[j] = fabs(c[base + j]);
        }
        __syncthreads();
        if (threadIdx.x == 0 && a != nullptr) s_dot = chain_add(s_dot, prod, cnt);
        if (threadIdx.x == 64 && c != nullptr) s_l1 = chain_add(s_l1, absv, cnt);  // a second wave runs the second chain
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

} // namespace
