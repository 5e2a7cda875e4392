// xsum.hip -- order-exact parallel sums on the device: the reference's left-to-right vec_dot / vec_l1norm
// (src/vec.c:67-85 there) reproduced bit for bit at streaming speed.  Arithmetic and proof sketch: xsum.h;
// host model of the same algorithm: oracle/xsum_model.cpp (tests/test_xsum_model.py).
//
// Four launches per call, for one or two sums over the same index range (slot 0 = dot, slot 1 = l1):
//   xs_cell_sums   one workgroup per group (64 cells x 128 elements): approximate (tree) sum of every cell and group
//   xs_group_scan  one workgroup: exclusive prefix of the group sums
//   xs_summarize   one workgroup per group: predicted running sum at each cell start -> assumed binade ->
//                  integer summary {D, lo, hi} of every cell, and their join per group
//   xs_drive       one wave per sum: walks the group summaries 64 at a time (one per lane, validated and applied
//                  with a wave prefix sum), drops to the cells of the first group that fails and to the elements
//                  of the first cell that fails, then finalises like seq_sums (alpha / beta / residual / done flag)
// Everything the driver skips was validated by integer arithmetic that is exact, so the result does not depend on
// the prediction: a wrong prediction only sends a cell down the element-by-element path.
#include "cgh_internal.h"
#include "device_utils.h"
#include "xsum.h"

struct cgh_xsum {
    int64_t     n = 0, ncells = 0, ngroups = 0;
    double     *cellsum = nullptr;   // [2][ncells]
    double     *groupsum = nullptr;  // [2][ngroups]
    double     *gprefix = nullptr;   // [2][ngroups]
    xs_summary *cells = nullptr;     // [2][ncells]
    xs_summary *groups = nullptr;    // [2][ngroups]
};

namespace {

enum { XS_DOT = 0, XS_L1 = 1, XS_BOTH_SAME = 2 };  // (a,b) | (c) | a == b == c

// terms of elements i, i+1 (i even); elements at or beyond n contribute +0.0, which leaves every sum unchanged
template <int KIND>
__device__ __forceinline__ void xs_terms(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                         const double *__restrict__ c, int64_t i, double (&dot)[2], double (&l1)[2])
{
    dot[0] = dot[1] = l1[0] = l1[1] = 0.0;
    if (i + 1 < n) {
        if constexpr (KIND == XS_DOT) {
            const double2 av = *reinterpret_cast<const double2 *>(a + i), bv = *reinterpret_cast<const double2 *>(b + i);
            dot[0] = mul_rn(av.x, bv.x);
            dot[1] = mul_rn(av.y, bv.y);
        } else if constexpr (KIND == XS_L1) {
            const double2 cv = *reinterpret_cast<const double2 *>(c + i);
            l1[0] = fabs(cv.x);
            l1[1] = fabs(cv.y);
        } else {
            const double2 av = *reinterpret_cast<const double2 *>(a + i);
            dot[0] = mul_rn(av.x, av.x);
            dot[1] = mul_rn(av.y, av.y);
            l1[0] = fabs(av.x);
            l1[1] = fabs(av.y);
        }
    } else if (i < n) {
        if constexpr (KIND == XS_DOT) dot[0] = mul_rn(a[i], b[i]);
        if constexpr (KIND == XS_L1) l1[0] = fabs(c[i]);
        if constexpr (KIND == XS_BOTH_SAME) {
            dot[0] = mul_rn(a[i], a[i]);
            l1[0] = fabs(a[i]);
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void xs_cell_sums(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                   const double *__restrict__ c, int64_t ncells, int64_t ngroups,
                                                   double *__restrict__ cellsum, double *__restrict__ groupsum)
{
    __shared__ double lds[2][XS_GROUP];
    const int     lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g = blockIdx.x;
#pragma unroll 4
    for (int cc = 0; cc < 16; ++cc) {
        const int     ci = wave * 16 + cc;
        const int64_t cell = g * XS_GROUP + ci;
        double        dot[2], l1[2];
        xs_terms<KIND>(cell < ncells ? n : 0, a, b, c, cell * XS_CELL + 2 * lane, dot, l1);
        const double sd = wave_sum(dot[0] + dot[1]);
        const double sl = wave_sum(l1[0] + l1[1]);
        if (lane == 0) {
            lds[0][ci] = sd;
            lds[1][ci] = sl;
            if (cell < ncells) {
                cellsum[cell] = sd;
                cellsum[ncells + cell] = sl;
            }
        }
    }
    __syncthreads();
    if (wave < 2) {
        const double v = wave_sum(lds[wave][lane]);
        if (lane == 0) groupsum[wave * ngroups + g] = v;
    }
}

// gprefix[slot][g] = groupsum[slot][0] + ... + groupsum[slot][g-1] (approximate: any order)
__global__ __launch_bounds__(256) void xs_group_scan(int64_t ngroups, const double *__restrict__ groupsum, double *__restrict__ gprefix)
{
    __shared__ double lds[256];
    const int64_t chunk = (ngroups + 255) / 256;
    const int64_t lo = chunk * threadIdx.x, hi = lo + chunk < ngroups ? lo + chunk : ngroups;
    for (int slot = 0; slot < 2; ++slot) {
        const double *in = groupsum + slot * ngroups;
        double       *out = gprefix + slot * ngroups;
        double        v = 0.0;
        for (int64_t k = lo; k < hi; ++k) v += in[k];
        lds[threadIdx.x] = v;
        __syncthreads();
        double run = 0.0;
        for (int t = 0; t < (int)threadIdx.x; ++t) run += lds[t];
        __syncthreads();
        for (int64_t k = lo; k < hi; ++k) {
            out[k] = run;
            run += in[k];
        }
    }
}

__device__ __forceinline__ int64_t shfl_down_i64(int64_t v, int off) { return (int64_t)__shfl_down((long long)v, off, 64); }
__device__ __forceinline__ int64_t shfl_up_i64(int64_t v, int off) { return (int64_t)__shfl_up((long long)v, off, 64); }
__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) { return (int64_t)__shfl((long long)v, src, 64); }

// join over the 64 lanes (lane order = run order); result valid in lane 0
__device__ __forceinline__ xs_summary wave_join(xs_summary q)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        xs_summary o;
        o.D = shfl_down_i64(q.D, off);
        o.lo = shfl_down_i64(q.lo, off);
        o.hi = shfl_down_i64(q.hi, off);
        o.E = __shfl_down(q.E, off, 64);
        o.meta = __shfl_down(q.meta, off, 64);
        // lanes whose partner is beyond the wave keep their own (their value is not used by lane 0's tree)
        if ((int)(threadIdx.x & 63) + off < 64) q = xs_join(q, o);
    }
    return q;
}

template <int KIND>
__global__ __launch_bounds__(256) void xs_summarize(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                   const double *__restrict__ c, int64_t ncells, int64_t ngroups,
                                                   const double *__restrict__ cellsum, const double *__restrict__ gprefix,
                                                   xs_summary *__restrict__ cells, xs_summary *__restrict__ groups)
{
    constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1;  // slots in use: S0..S1
    __shared__ double     start[2][XS_GROUP];  // predicted running sum at the start of each cell
    __shared__ xs_summary sm[2][XS_GROUP];
    const int     lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g = blockIdx.x;
    if (wave >= S0 && wave <= S1) {
        const int64_t cell = g * XS_GROUP + lane;
        const double  v = cell < ncells ? cellsum[wave * ncells + cell] : 0.0;
        double        incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        start[wave][lane] = gprefix[wave * ngroups + g] + (incl - v);
    }
    __syncthreads();
#pragma unroll 2
    for (int cc = 0; cc < 16; ++cc) {
        const int     ci = wave * 16 + cc;
        const int64_t cell = g * XS_GROUP + ci;
        double        t[2][2];
        xs_terms<KIND>(cell < ncells ? n : 0, a, b, c, cell * XS_CELL + 2 * lane, t[0], t[1]);
#pragma unroll
        for (int slot = S0; slot <= S1; ++slot) {
            xs_summary q = xs_identity();
            if (cell < ncells) {
                int32_t E, neg;
                if (xs_binade_of(start[slot][ci], &E, &neg)) {
                    bool          bad0 = false, bad1 = false;
                    const int64_t d0 = xs_element(t[slot][0], E, neg, &bad0), d1 = xs_element(t[slot][1], E, neg, &bad1);
                    int64_t       D = d0 + d1;  // |d| < 2^52 each
                    int64_t       lo = d0 < 0 ? d0 : 0, hi = d0 > 0 ? d0 : 0;
                    lo = D < lo ? D : lo;
                    hi = D > hi ? D : hi;
                    // all lanes assume the same binade: only D / lo / hi travel through the tree
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int64_t oD = shfl_down_i64(D, off), olo = shfl_down_i64(lo, off), ohi = shfl_down_i64(hi, off);
                        if (lane + off < 64) {
                            const int64_t blo = xs_sat_add(D, olo), bhi = xs_sat_add(D, ohi);
                            lo = lo < blo ? lo : blo;
                            hi = hi > bhi ? hi : bhi;
                            D = xs_sat_add(D, oD);
                        }
                    }
                    const bool any_bad = __ballot(bad0 || bad1) != 0ull;
                    q.D = D;
                    q.lo = lo;
                    q.hi = hi;
                    q.E = E;
                    q.meta = (neg ? 2 : 0) | (any_bad ? XS_IRREGULAR : 0);
                } else {
                    q.meta = XS_IRREGULAR;  // running sum predicted 0 / subnormal / non-finite: element by element
                }
            }
            if (lane == 0) {
                sm[slot][ci] = q;
                if (cell < ncells) cells[slot * ncells + cell] = q;
            }
        }
    }
    __syncthreads();
    if (wave >= S0 && wave <= S1) {
        const xs_summary q = wave_join(sm[wave][lane]);
        if (lane == 0) groups[wave * ngroups + g] = q;
    }
}

// ---------------------------------------------------------------------------------------------- the driver
__device__ __forceinline__ double readlane_f64(double v, int k)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

// Items [start, cnt) of a batch (one summary per lane): apply the longest valid prefix to the running sum s
// (wave-uniform) and return the index of the first item that does not apply (cnt if all did).
__device__ __forceinline__ int xs_batch(const xs_summary &q, int start, int cnt, double &s)
{
    const int lane = threadIdx.x & 63;
    int32_t   E0, neg0;
    if (!xs_binade_of(s, &E0, &neg0)) {
        // no binade (0, subnormal, inf, nan): only empty runs may be skipped
        const bool               active = lane >= start && lane < cnt;
        const unsigned long long fail = __ballot(active && !(q.E < 0 && !(q.meta & XS_IRREGULAR)));
        return fail ? (int)__builtin_ctzll(fail) : cnt;
    }
    const int64_t M = xs_mantissa(s);
    const bool    active = lane >= start && lane < cnt;
    const int64_t D = active ? q.D : 0;
    int64_t       incl = D;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t o = shfl_up_i64(incl, off);
        if (lane >= off) incl = xs_sat_add(incl, o);
    }
    int64_t excl = shfl_up_i64(incl, 1);
    if (lane == 0) excl = 0;
    const bool               ok = !active || xs_applies(q, E0, neg0, xs_sat_add(M, excl));
    const unsigned long long fail = __ballot(!ok);
    const int                f = fail ? (int)__builtin_ctzll(fail) : cnt;
    if (f > start) s = xs_compose(E0, neg0, M + shfl_i64(incl, f - 1));
    return f;
}

// s + (the 128 elements of one cell), left to right.  Every lane carries the same s.
template <int KIND>
__device__ __forceinline__ double xs_cell_sequential(double s, int slot, int64_t n, const double *__restrict__ a,
                                                     const double *__restrict__ b, const double *__restrict__ c, int64_t cell)
{
    const int lane = threadIdx.x & 63;
    double    t[2][2];
    xs_terms<KIND>(n, a, b, c, cell * XS_CELL + 2 * lane, t[0], t[1]);
    const double t0 = slot ? t[1][0] : t[0][0], t1 = slot ? t[1][1] : t[0][1];
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
        s = add_rn(s, readlane_f64(t0, k));
        s = add_rn(s, readlane_f64(t1, k));
    }
    return s;
}

template <int KIND>
__device__ double xs_walk(int slot, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                          const double *__restrict__ c, int64_t ncells, int64_t ngroups, const xs_summary *__restrict__ cells,
                          const xs_summary *__restrict__ groups)
{
    const int lane = threadIdx.x & 63;
    cells += slot * ncells;
    groups += slot * ngroups;
    double s = 0.0;
    for (int64_t g0 = 0; g0 < ngroups; g0 += 64) {
        const int  cnt = (int)(ngroups - g0 < 64 ? ngroups - g0 : 64);
        xs_summary Q = xs_identity();
        if (lane < cnt) Q = groups[g0 + lane];
        int start = 0;
        while (start < cnt) {
            const int f = xs_batch(Q, start, cnt, s);
            if (f >= cnt) break;
            const int64_t c0 = (g0 + f) * XS_GROUP;
            const int     ccnt = (int)(ncells - c0 < XS_GROUP ? ncells - c0 : XS_GROUP);
            xs_summary    C = xs_identity();
            if (lane < ccnt) C = cells[c0 + lane];
            int cstart = 0;
            while (cstart < ccnt) {
                const int cf = xs_batch(C, cstart, ccnt, s);
                if (cf >= ccnt) break;
                s = xs_cell_sequential<KIND>(s, slot, n, a, b, c, c0 + cf);
                cstart = cf + 1;
            }
            start = f + 1;
        }
    }
    return s;
}

template <int KIND>
__global__ __launch_bounds__(128) void xs_drive(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                               const double *__restrict__ c, int64_t ncells, int64_t ngroups,
                                               const xs_summary *__restrict__ cells, const xs_summary *__restrict__ groups,
                                               int mode, double *hist, int iter, double tol, int *done, double *out)
{
    constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1;
    __shared__ double res[2];
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) res[threadIdx.x] = 0.0;
    __syncthreads();
    const bool frozen = done != nullptr && mode != 3 && mode != 0 && *done != 0;
    if (!frozen && wave >= S0 && wave <= S1) {
        const double s = xs_walk<KIND>(wave, n, a, b, c, ncells, ngroups, cells, groups);
        if ((threadIdx.x & 63) == 0) res[wave] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0 || frozen) return;
    const double s_dot = res[0], s_l1 = res[1];
    if (iter == 0 && mode == 1) iter = done[2] + 1;  // as seq_sums: iteration number from the device (graph replay)
    if (iter == 0 && mode == 2) iter = done[1];
    if (mode == 0) {
        out[0] = s_dot;
        out[1] = s_l1;
    } else if (mode == 1) {
        hist[(size_t)iter * CGH_HIST + 2] = hist[(size_t)(iter - 1) * CGH_HIST + 1] / s_dot;
    } else if (mode == 2) {
        const double r = s_l1 / 2.0;
        const bool   conv = r < tol;
        hist[(size_t)iter * CGH_HIST + 0] = r;
        hist[(size_t)iter * CGH_HIST + 1] = s_dot;
        hist[(size_t)iter * CGH_HIST + 3] = conv ? NAN : s_dot / hist[(size_t)(iter - 1) * CGH_HIST + 1];
        if (conv) *done = iter;
    } else {
        hist[0] = s_l1 / 2.0;
        hist[1] = s_dot;
        hist[2] = NAN;
        hist[3] = NAN;
    }
}

template <int KIND>
int launch_all(cgh_xsum *w, int64_t n, const double *a, const double *b, const double *c, int mode, double *hist, int iter,
               double tol, int *done, double *out, hipStream_t s)
{
    const unsigned ng = (unsigned)w->ngroups;
    hipLaunchKernelGGL((xs_cell_sums<KIND>), dim3(ng), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups, w->cellsum, w->groupsum);
    hipLaunchKernelGGL(xs_group_scan, dim3(1), dim3(256), 0, s, w->ngroups, (const double *)w->groupsum, w->gprefix);
    hipLaunchKernelGGL((xs_summarize<KIND>), dim3(ng), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups,
                       (const double *)w->cellsum, (const double *)w->gprefix, w->cells, w->groups);
    hipLaunchKernelGGL((xs_drive<KIND>), dim3(1), dim3(128), 0, s, n, a, b, c, w->ncells, w->ngroups,
                       (const xs_summary *)w->cells, (const xs_summary *)w->groups, mode, hist, iter, tol, done, out);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

} // namespace

cgh_xsum *cgh_xsum_create(int64_t n)
{
    if (n <= 0) return nullptr;
    cgh_xsum *w = new cgh_xsum();
    w->n = n;
    w->ncells = (n + XS_CELL - 1) / XS_CELL;
    w->ngroups = (w->ncells + XS_GROUP - 1) / XS_GROUP;
    w->cellsum = (double *)cgh_alloc(sizeof(double) * 2 * (size_t)w->ncells);
    w->groupsum = (double *)cgh_alloc(sizeof(double) * 2 * (size_t)w->ngroups);
    w->gprefix = (double *)cgh_alloc(sizeof(double) * 2 * (size_t)w->ngroups);
    w->cells = (xs_summary *)cgh_alloc(sizeof(xs_summary) * 2 * (size_t)w->ncells);
    w->groups = (xs_summary *)cgh_alloc(sizeof(xs_summary) * 2 * (size_t)w->ngroups);
    if (!w->cellsum || !w->groupsum || !w->gprefix || !w->cells || !w->groups) {
        cgh_xsum_free(w);
        return nullptr;
    }
    return w;
}

void cgh_xsum_free(cgh_xsum *w)
{
    if (!w) return;
    cgh_free(w->cellsum);
    cgh_free(w->groupsum);
    cgh_free(w->gprefix);
    cgh_free(w->cells);
    cgh_free(w->groups);
    delete w;
}

int cgh_xsum_launch(cgh_xsum *w, int64_t n, const double *a, const double *b, const double *c, int mode, double *hist,
                    int iter, double tol, int *done, double *out, hipStream_t s)
{
    if (!w || n != w->n) {
        cgh_set_error("cgh_xsum_launch: workspace was created for another length");
        return CGH_EINVAL;
    }
    // the 16-byte loads need aligned vectors (every vector of this library is: cgh_alloc, and the solver checks b and x)
    const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
    if (al & 15u) {
        cgh_set_error("cgh_xsum_launch: vectors must be 16-byte aligned");
        return CGH_EINVAL;
    }
    if (a && b && !c) return launch_all<XS_DOT>(w, n, a, b, c, mode, hist, iter, tol, done, out, s);
    if (!a && !b && c) return launch_all<XS_L1>(w, n, a, b, c, mode, hist, iter, tol, done, out, s);
    if (a && a == b && a == c) return launch_all<XS_BOTH_SAME>(w, n, a, b, c, mode, hist, iter, tol, done, out, s);
    cgh_set_error("cgh_xsum_launch: unsupported operand combination");
    return CGH_EINVAL;
}
