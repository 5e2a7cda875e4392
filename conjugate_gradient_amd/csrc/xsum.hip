// xsum.hip -- order-exact parallel sums on the device: the reference's left-to-right vec_dot / vec_l1norm
// (src/vec.c:67-85 there) reproduced bit for bit at streaming speed.  Arithmetic and proof sketch: xsum.h;
// host model of the same algorithm: oracle/xsum_model.cpp (tests/test_xsum_model.py).
//
// Four launches per call, for one or two sums over the same index range (slot 0 = dot, slot 1 = l1):
//   xs_cell_sums   one workgroup per group (64 cells x 128 elements): approximate (tree) sum of every cell and group
//   xs_group_scan  one workgroup: exclusive prefix of the group sums (only beyond 4096 groups; below that every
//                  workgroup of xs_summarize adds up the sums of the groups before it)
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

// groupsum[slot][g] = sum of the group's 64 cell sums, when the cell sums came from the kernel that produced the
// vector (cg_update_r in strict mode writes them as a by-product: one sweep over r less).  One wave per (slot, group).
__global__ __launch_bounds__(256) void xs_group_sums(int64_t ncells, int64_t ngroups, const double *__restrict__ cellsum,
                                                    double *__restrict__ groupsum, int slot_lo, int slot_hi)
{
    const int     lane = threadIdx.x & 63;
    const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t per = ngroups;
    const int     slot = slot_lo + (int)(wv / per);
    const int64_t g = wv % per;
    if (slot > slot_hi) return;
    const int64_t cell = g * XS_GROUP + lane;
    const double  v = wave_sum(cell < ncells ? cellsum[slot * ncells + cell] : 0.0);
    if (lane == 0) groupsum[slot * ngroups + g] = v;
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

// terms of the 8 consecutive elements i .. i+7 (i a multiple of 8) for the slots in use
template <int KIND>
__device__ __forceinline__ void xs_terms8(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                          const double *__restrict__ c, int64_t i, double (&t)[2][8])
{
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        double d[2], l[2];
        xs_terms<KIND>(n, a, b, c, i + k, d, l);
        t[0][k] = d[0];
        t[0][k + 1] = d[1];
        t[1][k] = l[0];
        t[1][k + 1] = l[1];
    }
}

// One workgroup per group.  16 lanes share a cell (8 consecutive elements each: a short serial prefix in registers,
// then a 4-step tree inside the 16 lanes), so a wave summarises 4 cells at a time and the workgroup 16.  Inside a
// cell no saturation is needed: 128 elements of |d| < 2^52 stay below 2^59.
template <int KIND>
__global__ __launch_bounds__(256) void xs_summarize(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                   const double *__restrict__ c, int64_t ncells, int64_t ngroups,
                                                   const double *__restrict__ cellsum, const double *__restrict__ groupsum,
                                                   const double *__restrict__ gprefix, xs_summary *__restrict__ cells,
                                                   xs_summary *__restrict__ groups)
{
    constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1;  // slots in use: S0..S1
    __shared__ double     start[2][XS_GROUP];  // predicted running sum at the start of each cell
    __shared__ xs_summary sm[2][XS_GROUP];
    __shared__ double     red[4];
    const int     lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // groups == NULL ("flat", few cells): four workgroups per group, one pass of 16 cells each, no group summary --
    // the whole problem is then a handful of workgroups, and their serial depth is what counts
    const bool    flat = groups == nullptr;
    const int64_t g = flat ? blockIdx.x >> 2 : blockIdx.x;
    // approximate sum of everything before this group: from the scan kernel, or (few groups) summed here
    double before[2] = {0.0, 0.0};
#pragma unroll
    for (int slot = S0; slot <= S1; ++slot)
        before[slot] = gprefix    ? gprefix[slot * ngroups + g]
                       : groupsum ? block_sum_partials(groupsum + slot * ngroups, (int)g, 1, red)
                                  : block_sum_partials(cellsum + slot * ncells, (int)(g * XS_GROUP), 1, red);  // few cells
    if (wave >= S0 && wave <= S1) {
        const int64_t cell = g * XS_GROUP + lane;
        const double  v = cell < ncells ? cellsum[wave * ncells + cell] : 0.0;
        double        incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        start[wave][lane] = (wave ? before[1] : before[0]) + (incl - v);
    }
    __syncthreads();
    const int sub = threadIdx.x & 15;  // position inside the cell's 16 lanes
    const int pass_lo = flat ? (int)(blockIdx.x & 3) : 0, pass_hi = flat ? pass_lo + 1 : 4;
#pragma unroll 1
    for (int pass = pass_lo; pass < pass_hi; ++pass) {
        const int     ci = pass * 16 + (threadIdx.x >> 4);
        const int64_t cell = g * XS_GROUP + ci;
        double        t[2][8];
        xs_terms8<KIND>(cell < ncells ? n : 0, a, b, c, cell * XS_CELL + 8 * sub, t);
#pragma unroll
        for (int slot = S0; slot <= S1; ++slot) {
            int32_t    E, neg;
            const bool usable = xs_binade_of(start[slot][ci], &E, &neg);  // else: predicted 0 / subnormal / non-finite
            bool       bad = !usable;
            int64_t    D = 0, lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t d = xs_element(t[slot][k], usable ? E : 1075, neg, &bad);
                D += d;
                lo = D < lo ? D : lo;
                hi = D > hi ? D : hi;
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const int64_t oD = shfl_down_i64(D, off), olo = shfl_down_i64(lo, off), ohi = shfl_down_i64(hi, off);
                if (sub + off < 16) {
                    const int64_t blo = D + olo, bhi = D + ohi;
                    lo = lo < blo ? lo : blo;
                    hi = hi > bhi ? hi : bhi;
                    D += oD;
                }
            }
            const unsigned long long bad_lanes = __ballot(bad);
            // A cell of zeros only leaves any running sum as it is (s + (+-0) = s; the walk starts at +0 and never holds
            // -0), whatever binade was predicted -- and none can be predicted while everything before it is zero too.
            // CG on a right-hand side with small support (the OpenFOAM systems) keeps such prefixes for hundreds of
            // iterations: without this every cell of them went element by element (foam-32k: 158 of 250 cells).
            bool zeros = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) zeros = zeros && t[slot][k] == 0.0;
            const unsigned long long zero_lanes = __ballot(zeros);
            const bool               cell_is_zero = ((zero_lanes >> (lane & 48)) & 0xffffull) == 0xffffull;
            if (sub == 0) {
                xs_summary q = xs_identity();
                if (cell < ncells) {
                    if (!cell_is_zero) {
                        q.D = D;
                        q.lo = lo;
                        q.hi = hi;
                        q.E = usable ? E : 0;
                        q.meta = (neg ? 2 : 0) | (((bad_lanes >> (lane & 48)) & 0xffffull) ? XS_IRREGULAR : 0);
                    }
                    cells[slot * ncells + cell] = q;
                }
                sm[slot][ci] = q;
            }
        }
    }
    if (flat) return;
    __syncthreads();
    if (wave >= S0 && wave <= S1) {
        const xs_summary q = wave_join(sm[wave][lane]);
        if (lane == 0) groups[wave * ngroups + g] = q;
    }
}

// ---------------------------------------------------------------------------------------------- the driver
// Optional profile of the serial walk (tools/prof_drive.py): lane 0 of the first wave in use adds event counts and
// 100 MHz ticks into 16 words.  [0] batches of groups, [1] groups walked cell by cell, [2] cells added element by
// element, [3] ticks in total, [4] ticks inside groups walked cell by cell, [5] ticks inside element chains (incl.
// waiting for their elements), [6] launches.
#define XS_PROF_ADD(k, v)                                                   \
    do {                                                                    \
        if (prof != nullptr && (threadIdx.x & 63) == 0) prof[k] += (v);     \
    } while (0)
__device__ __forceinline__ double readlane_f64(double v, int k)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

// v of the lane selected by a DPP control (0 where the control has no source lane or the row is masked out)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int64_t dpp_i64(int64_t v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__device__ __forceinline__ int64_t readlane_i64(int64_t v, int k)
{
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, k), hi = __builtin_amdgcn_readlane((int)(v >> 32), k);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
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
    // Inclusive prefix over the lanes with DPP moves (no LDS crossbar round trips on this serial path).  Plain wrapping
    // adds: the prefix of lane k involves lanes <= k only, so it is exact for every lane up to and including the first
    // one that fails (all runs before it are valid, hence small), and nothing beyond that lane is used.
    int64_t incl = D;
    incl += dpp_i64<0x111, 0xf>(incl);  // row_shr:1
    incl += dpp_i64<0x112, 0xf>(incl);  // row_shr:2
    incl += dpp_i64<0x114, 0xf>(incl);  // row_shr:4
    incl += dpp_i64<0x118, 0xf>(incl);  // row_shr:8   -> prefix inside each row of 16 lanes
    incl += dpp_i64<0x142, 0xa>(incl);  // row_bcast:15 into rows 1 and 3
    incl += dpp_i64<0x143, 0xc>(incl);  // row_bcast:31 into rows 2 and 3
    const int64_t excl = incl - D;
    const bool               ok = !active || xs_applies(q, E0, neg0, xs_sat_add(M, excl));
    const unsigned long long fail = __ballot(!ok);
    const int                f = fail ? (int)__builtin_ctzll(fail) : cnt;
    if (f > start) s = xs_compose(E0, neg0, M + readlane_i64(incl, f - 1));
    return f;
}

// s + (the 128 elements of one cell: lane l holds elements 2l, 2l+1 in t0, t1), left to right.  Every lane carries the
// same s.  The elements go through this wave's 1 KiB of LDS: the reads are broadcasts that do not depend on the
// additions, so they run ahead of the one dependent chain (as chain_add of seq_sums, cg_kernels.h).
__device__ __forceinline__ double xs_chain(double s, double t0, double t1, double *stage)
{
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the previous cell's reads are over
    *reinterpret_cast<double2 *>(stage + 2 * lane) = make_double2(t0, t1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
    for (int j = 0; j < XS_CELL; j += 16) {
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            const double2 w = *reinterpret_cast<const double2 *>(stage + j + k);
            t[k] = w.x;
            t[k + 1] = w.y;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s = add_rn(s, t[k]);
    }
    return s;
}

template <int KIND>
__device__ __forceinline__ void xs_cell_terms(int slot, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                              const double *__restrict__ c, int64_t cell, double &t0, double &t1)
{
    double t[2][2];
    xs_terms<KIND>(n, a, b, c, cell * XS_CELL + 2 * (threadIdx.x & 63), t[0], t[1]);
    t0 = slot ? t[1][0] : t[0][0];
    t1 = slot ? t[1][1] : t[0][1];
}

constexpr int XS_PF_CELLS = 8;  // cells of a batch whose elements are fetched ahead (those flagged irregular)

// Elements of the cells of a batch that are KNOWN to need the element-by-element path (irregular flag: tie, binade
// change, ...), requested all at once: the serial chain of the driver is memory latency.
struct xs_prefetch {
    int    idx[XS_PF_CELLS];
    double t0[XS_PF_CELLS], t1[XS_PF_CELLS];
};

template <int KIND>
__device__ __forceinline__ void xs_prefetch_cells(xs_prefetch &f, int slot, int64_t n, const double *__restrict__ a,
                                                  const double *__restrict__ b, const double *__restrict__ c, int64_t c0,
                                                  int ccnt, const xs_summary &C)
{
    const int          lane = threadIdx.x & 63;
    unsigned long long flagged = __ballot(lane < ccnt && (C.meta & XS_IRREGULAR));
#pragma unroll
    for (int j = 0; j < XS_PF_CELLS; ++j) {
        f.idx[j] = flagged ? (int)__builtin_ctzll(flagged) : -1;
        flagged &= flagged - 1;
        f.t0[j] = f.t1[j] = 0.0;
        if (f.idx[j] >= 0) xs_cell_terms<KIND>(slot, n, a, b, c, c0 + f.idx[j], f.t0[j], f.t1[j]);
    }
}

// The cells [c0, c0+ccnt) of one batch (summary of cell c0+lane in C), applied to s.  A cell that fails only the range
// check against the actual running sum (not prefetched) is fetched when it is met.
template <int KIND>
__device__ __forceinline__ double xs_walk_cells(double s, int slot, int64_t n, const double *__restrict__ a,
                                                const double *__restrict__ b, const double *__restrict__ c, int64_t c0,
                                                int ccnt, const xs_summary &C, double *stage, const xs_prefetch &f,
                                                long long *prof = nullptr)
{
    int cstart = 0;
    while (cstart < ccnt) {
        const int cf = xs_batch(C, cstart, ccnt, s);
        if (cf >= ccnt) break;
        const long long tc0 = prof ? wall_clock64() : 0;
        double t0 = 0.0, t1 = 0.0;
        bool   have = false;
#pragma unroll
        for (int j = 0; j < XS_PF_CELLS; ++j)
            if (f.idx[j] == cf) {
                t0 = f.t0[j];
                t1 = f.t1[j];
                have = true;
            }
        if (!have) xs_cell_terms<KIND>(slot, n, a, b, c, c0 + cf, t0, t1);
        s = xs_chain(s, t0, t1, stage);
        if (prof) {
            XS_PROF_ADD(2, 1);
            XS_PROF_ADD(5, wall_clock64() - tc0 + (long long)(s != s));  // (depends on s: the clock is read after the chain)
        }
        cstart = cf + 1;
    }
    return s;
}

// Few cells (<= 4096): no group level.  The cell summaries are walked 64 at a time, two batches ahead in flight: when
// batch k is walked, its summaries and the elements of its flagged cells arrived while batch k-1 was walked.
template <int KIND>
__device__ double xs_walk_flat(int slot, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                               const double *__restrict__ c, int64_t ncells, const xs_summary *__restrict__ cells, double *stage,
                               long long *prof = nullptr)
{
    const int lane = threadIdx.x & 63;
    cells += slot * ncells;
    auto load = [&](int64_t c0) {
        xs_summary q = xs_identity();
        if (c0 + lane < ncells) q = cells[c0 + lane];
        return q;
    };
    auto count = [&](int64_t c0) { return (int)(ncells - c0 < 64 ? (ncells - c0 > 0 ? ncells - c0 : 0) : 64); };
    double      s = 0.0;
    xs_summary  C0 = load(0), C1 = load(64);
    xs_prefetch f0, f1;
    xs_prefetch_cells<KIND>(f0, slot, n, a, b, c, 0, count(0), C0);
    for (int64_t c0 = 0; c0 < ncells; c0 += 64) {
        const xs_summary C2 = load(c0 + 128);
        xs_prefetch_cells<KIND>(f1, slot, n, a, b, c, c0 + 64, count(c0 + 64), C1);
        XS_PROF_ADD(0, 1);
        s = xs_walk_cells<KIND>(s, slot, n, a, b, c, c0, count(c0), C0, stage, f0, prof);
        C0 = C1;
        C1 = C2;
        f0 = f1;
    }
    return s;
}

template <int KIND>
__device__ double xs_walk(int slot, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                          const double *__restrict__ c, int64_t ncells, int64_t ngroups, const xs_summary *__restrict__ cells,
                          const xs_summary *__restrict__ groups, double *stage, long long *prof = nullptr)
{
    const int lane = threadIdx.x & 63;
    cells += slot * ncells;
    groups += slot * ngroups;
    double s = 0.0;
    // group summaries: three batches ahead of the one being walked (a batch is ~0.5 us of work, a miss in L2 ~2 us)
    auto load_groups = [&](int64_t g0) {
        xs_summary q = xs_identity();
        if (g0 + lane < ngroups) q = groups[g0 + lane];
        return q;
    };
    xs_summary Q0 = load_groups(0), Q1 = load_groups(64), Q2 = load_groups(128);
    for (int64_t g0 = 0; g0 < ngroups; g0 += 64) {
        const int        cnt = (int)(ngroups - g0 < 64 ? ngroups - g0 : 64);
        const xs_summary Q = Q0;
        Q0 = Q1;
        Q1 = Q2;
        Q2 = load_groups(g0 + 192);
        // Cell summaries of the groups already known to need their cells (irregular: a binade change is predicted inside,
        // a tie, ...): two of them are always in flight ahead of the walk, refilled as the walk passes them.
        unsigned long long remaining = __ballot(lane < cnt && (Q.meta & XS_IRREGULAR));
        int                pf_g[2] = {-1, -1};
        xs_summary         pf_C[2] = {xs_identity(), xs_identity()};
        auto refill = [&](int j) {
            pf_g[j] = remaining ? (int)__builtin_ctzll(remaining) : -1;
            remaining &= remaining - 1;
            pf_C[j] = xs_identity();
            if (pf_g[j] >= 0 && (g0 + pf_g[j]) * XS_GROUP + lane < ncells) pf_C[j] = cells[(g0 + pf_g[j]) * XS_GROUP + lane];
        };
        refill(0);
        refill(1);
        XS_PROF_ADD(0, 1);
        int start = 0;
        while (start < cnt) {
            const int f = xs_batch(Q, start, cnt, s);
            if (f >= cnt) break;
            const long long tg0 = prof ? wall_clock64() : 0;
            const int64_t c0 = (g0 + f) * XS_GROUP;
            const int     ccnt = (int)(ncells - c0 < XS_GROUP ? ncells - c0 : XS_GROUP);
            xs_summary    C = xs_identity();
            bool          have = false;
            remaining &= f >= 63 ? 0ull : ~((2ull << f) - 1ull);  // nothing at or before f is worth fetching any more
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (pf_g[j] == f) {
                    C = pf_C[j];
                    have = true;
                }
                if (pf_g[j] >= 0 && pf_g[j] <= f) refill(j);  // used, or passed without being needed
            }
            if (!have && lane < ccnt) C = cells[c0 + lane];
            xs_prefetch pf;
            xs_prefetch_cells<KIND>(pf, slot, n, a, b, c, c0, ccnt, C);
            s = xs_walk_cells<KIND>(s, slot, n, a, b, c, c0, ccnt, C, stage, pf, prof);
            if (prof) {
                XS_PROF_ADD(1, 1);
                XS_PROF_ADD(4, wall_clock64() - tg0 + (long long)(s != s));
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
                                               int mode, double *hist, int iter, double tol, int *done, double *out,
                                               long long *prof_buf)
{
    constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1;
    __shared__ double res[2];
    __shared__ __attribute__((aligned(16))) double stage[2][XS_CELL];
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) res[threadIdx.x] = 0.0;
    __syncthreads();
    const bool frozen = done != nullptr && mode != 3 && mode != 0 && *done != 0;
    if (!frozen && wave >= S0 && wave <= S1) {
        long long      *prof = wave == S0 ? prof_buf : nullptr;
        const long long t0 = prof ? wall_clock64() : 0;
        const double s = groups != nullptr ? xs_walk<KIND>(wave, n, a, b, c, ncells, ngroups, cells, groups, stage[wave], prof)
                                           : xs_walk_flat<KIND>(wave, n, a, b, c, ncells, cells, stage[wave], prof);
        if ((threadIdx.x & 63) == 0) res[wave] = s;
        if (prof) {
            XS_PROF_ADD(3, wall_clock64() - t0 + (long long)(s != s));
            XS_PROF_ADD(6, 1);
        }
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

long long *g_xs_prof = nullptr;  // device buffer of 16 words, or NULL (cgh_xsum_set_prof)

template <int KIND>
int launch_all(cgh_xsum *w, bool presummed, int64_t n, const double *a, const double *b, const double *c, int mode, double *hist, int iter,
               double tol, int *done, double *out, hipStream_t s)
{
    const unsigned ng = (unsigned)w->ngroups;
    const double *gsum = w->groupsum;
    if (presummed && w->ncells <= 4096) {
        gsum = nullptr;  // every workgroup of xs_summarize adds up the cell sums before its group itself: no launch here
    } else if (presummed) {
        constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1;
        const int64_t waves = (int64_t)(S1 - S0 + 1) * w->ngroups;
        hipLaunchKernelGGL(xs_group_sums, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, w->ncells, w->ngroups,
                           (const double *)w->cellsum, w->groupsum, S0, S1);
    } else {
        hipLaunchKernelGGL((xs_cell_sums<KIND>), dim3(ng), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups, w->cellsum,
                           w->groupsum);
    }
    // few groups: every workgroup of xs_summarize adds the group sums before it itself (one launch less)
    const double *prefix = nullptr;
    if (w->ngroups > 4096) {
        hipLaunchKernelGGL(xs_group_scan, dim3(1), dim3(256), 0, s, w->ngroups, (const double *)w->groupsum, w->gprefix);
        prefix = w->gprefix;
    }
    const bool  flat = w->ncells <= 4096;  // few cells: no group level (xs_walk_flat)
    xs_summary *grp = flat ? nullptr : w->groups;
    hipLaunchKernelGGL((xs_summarize<KIND>), dim3(flat ? 4 * ng : ng), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups,
                       (const double *)w->cellsum, gsum, prefix, w->cells, grp);
    hipLaunchKernelGGL((xs_drive<KIND>), dim3(1), dim3(128), 0, s, n, a, b, c, w->ncells, w->ngroups,
                       (const xs_summary *)w->cells, (const xs_summary *)grp, mode, hist, iter, tol, done, out, g_xs_prof);
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

// Profile of the serial walks that follow (see XS_PROF_ADD): 16 zeroed 64-bit words of device memory, or NULL to stop.
extern "C" void cgh_xsum_set_prof(void *dev_16_words) { g_xs_prof = (long long *)dev_16_words; }

double *cgh_xsum_cellsums(cgh_xsum *w) { return w ? w->cellsum : nullptr; }
int64_t cgh_xsum_cells(const cgh_xsum *w) { return w ? w->ncells : 0; }

int cgh_xsum_launch(cgh_xsum *w, int64_t n, const double *a, const double *b, const double *c, int mode, double *hist,
                    int iter, double tol, int *done, double *out, hipStream_t s, bool presummed)
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
    if (a && b && !c) return launch_all<XS_DOT>(w, presummed, n, a, b, c, mode, hist, iter, tol, done, out, s);
    if (!a && !b && c) return launch_all<XS_L1>(w, presummed, n, a, b, c, mode, hist, iter, tol, done, out, s);
    if (a && a == b && a == c) return launch_all<XS_BOTH_SAME>(w, presummed, n, a, b, c, mode, hist, iter, tol, done, out, s);
    cgh_set_error("cgh_xsum_launch: unsupported operand combination");
    return CGH_EINVAL;
}
