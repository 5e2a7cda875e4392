// xsum.hip -- order-exact parallel sums on the device: the reference's left-to-right vec_dot / vec_l1norm
// (src/vec.c:67-85 there) reproduced bit for bit at streaming speed.  Arithmetic and proof sketch: xsum.h;
// host model of the same algorithm: oracle/xsum_model.cpp (tests/test_xsum_model.py).
//
// Launches per call, for one or two sums over the same index range (slot 0 = dot, slot 1 = l1); from 2049 cells on,
// xs_summarize and xs_drive run as ONE launch, xs_fused (further down), whose driver walks while the summaries are built:
//   xs_cell_sums   one workgroup per group (64 cells x 128 elements): approximate (tree) sum of every cell and group
//   xs_group_scan  one workgroup: exclusive prefix of the group sums (only beyond 4096 groups; below that every
//                  workgroup of xs_summarize adds up the sums of the groups before it)
//   xs_summarize   one workgroup per group: predicted running sum at each cell start -> assumed binade ->
//                  integer summary {D, lo, hi} of every cell, and their join per group
//   xs_drive       one workgroup per sum: 16 waves stage the summaries in LDS and join them 64 at a time, one wave
//                  walks the joined summaries (validated and applied with a wave prefix sum), drops into the batch,
//                  the group and the cell that fails, then finalises like seq_sums (alpha / beta / residual / done)
// Everything the driver skips was validated by integer arithmetic that is exact, so the result does not depend on
// the prediction: a wrong prediction only sends a cell down the element-by-element path.
#include <new>

#include "cgh_internal.h"
#include "device_utils.h"
#include "xsum.h"
#include "halo_ll.h"

struct cgh_xsum {
    int64_t     n = 0, ncells = 0, ngroups = 0;
    double     *cellsum = nullptr;   // [2][ncells]
    double     *groupsum = nullptr;  // [2][ngroups]
    double     *gprefix = nullptr;   // [2][ngroups]
    xs_summary *cells = nullptr;     // [2][ncells]
    xs_summary *groups = nullptr;    // [2][ngroups]
    double     *xfer = nullptr;      // [4]: the two sums on their way to the finalising workgroup, its ticket
    unsigned   *ready = nullptr;     // [2][batches of 64 groups]: groups summarised so far in the running fused launch (zero between launches)
    xs_summary *batches = nullptr;   // [2][batches]: the join of each batch's group summaries (fused launch)
    unsigned   *head = nullptr;      // [2]: workgroups of the head group counted so far (fused launch, group level; zero between launches)
    unsigned long long *tail = nullptr;  // [8] tagged words: the verdict of a fused launch of few cells for the workgroups that stay on for the vector update (xs_tail)
    unsigned    tail_seq = 0;        // launches with such a tail so far: their tag
    bool        owns = true;         // false: object and arrays live in storage of the caller's (cgh_xsum_place)
    bool        groups_presummed = false;  // the producer of the vector also left the group sums (consumed by the next launch)
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

// The same in two steps, so that the loads of the next pass can be in flight while this one is worked on:
// the operands of elements i, i+1 (zeros at or beyond n), then their terms.
template <int KIND>
__device__ __forceinline__ void xs_raw(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                       const double *__restrict__ c, int64_t i, double2 &u, double2 &v)
{
    const double *p = KIND == XS_L1 ? c : a;
    u = v = make_double2(0.0, 0.0);
    if (i + 1 < n) {
        u = *reinterpret_cast<const double2 *>(p + i);
        if constexpr (KIND == XS_DOT) v = *reinterpret_cast<const double2 *>(b + i);
    } else if (i < n) {
        u.x = p[i];
        if constexpr (KIND == XS_DOT) v.x = b[i];
    }
}
template <int KIND>
__device__ __forceinline__ void xs_raw_terms(const double2 &u, const double2 &v, double (&dot)[2], double (&l1)[2])
{
    dot[0] = dot[1] = l1[0] = l1[1] = 0.0;
    if constexpr (KIND == XS_DOT) {
        dot[0] = mul_rn(u.x, v.x);
        dot[1] = mul_rn(u.y, v.y);
    } else if constexpr (KIND == XS_L1) {
        l1[0] = fabs(u.x);
        l1[1] = fabs(u.y);
    } else {
        dot[0] = mul_rn(u.x, u.x);
        dot[1] = mul_rn(u.y, u.y);
        l1[0] = fabs(u.x);
        l1[1] = fabs(u.y);
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

// v of the lane selected by a DPP control inside the row of 16 lanes (unspecified where the control has no source lane)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int64_t shfl_down_i64(int64_t v, int off) { return (int64_t)__shfl_down((long long)v, off, 64); }
__device__ __forceinline__ int64_t shfl_up_i64(int64_t v, int off) { return (int64_t)__shfl_up((long long)v, off, 64); }
__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) { return (int64_t)__shfl((long long)v, src, 64); }

// 64-bit / 32-bit DPP moves inside a row of 16 lanes (no LDS crossbar round trips)
template <int CTRL>
__device__ __forceinline__ int64_t row_dpp_i64(int64_t v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, 0xf, 0xf, false);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
template <int CTRL>
__device__ __forceinline__ int row_dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int64_t readlane_i64_early(int64_t v, int k)
{
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, k), hi = __builtin_amdgcn_readlane((int)(v >> 32), k);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// join over the 64 lanes (lane order = run order); result valid in lane 0.  Inside each row of 16 lanes a tree of DPP
// moves (lane i takes lane i + off of its row), then the four row summaries are read out of lanes 16, 32, 48 and joined
// left to right: no LDS round trip anywhere (the shuffle tree over 64 lanes cost 72 ds_bpermutes per join).
__device__ __forceinline__ xs_summary wave_join(xs_summary q)
{
    const int sub = threadIdx.x & 15;
#define XS_JOIN_STEP(CTRL, OFF)                       \
    {                                                 \
        xs_summary o;                                 \
        o.A = row_dpp_i64<CTRL>(q.A);                 \
        o.C = row_dpp_i64<CTRL>(q.C);                 \
        o.Mlo = row_dpp_i64<CTRL>(q.Mlo);             \
        o.Mhi = row_dpp_i64<CTRL>(q.Mhi);             \
        o.E = row_dpp_i32<CTRL>(q.E);                 \
        o.meta = row_dpp_i32<CTRL>(q.meta);           \
        if (sub + OFF < 16) q = xs_join(q, o);        \
    }
    XS_JOIN_STEP(0x101, 1)  // row_shl:1
    XS_JOIN_STEP(0x102, 2)
    XS_JOIN_STEP(0x104, 4)
    XS_JOIN_STEP(0x108, 8)
#undef XS_JOIN_STEP
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        xs_summary o;
        o.A = readlane_i64_early(q.A, 16 * row);
        o.C = readlane_i64_early(q.C, 16 * row);
        o.Mlo = readlane_i64_early(q.Mlo, 16 * row);
        o.Mhi = readlane_i64_early(q.Mhi, 16 * row);
        o.E = __builtin_amdgcn_readlane(q.E, 16 * row);
        o.meta = __builtin_amdgcn_readlane(q.meta, 16 * row);
        if ((threadIdx.x & 63) == 0) q = xs_join(q, o);  // (only lane 0 carries the running join: lanes 16, 32, 48 keep their rows')
    }
    return q;
}

// ---- cells with one special element (an exact tie, or the element at which the sum enters the next binade): xsum.h.
// The 16 lanes of a cell (one DPP row; lane `sub` holds elements 8 sub .. 8 sub + 7) build what xs_cell_summary states
// sequentially.  Called by whole waves; rows that do not need it compute values nobody uses.
__device__ __forceinline__ int64_t row_excl_scan(int64_t v, int sub)
{
    int64_t incl = v;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const int64_t o = (int64_t)__shfl_up((long long)incl, off, 16);
        if (sub >= off) incl += o;
    }
    return incl - v;
}
__device__ __forceinline__ int64_t row_min(int64_t v)
{
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const int64_t o = (int64_t)__shfl_xor((long long)v, m, 16);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int64_t row_max(int64_t v)
{
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const int64_t o = (int64_t)__shfl_xor((long long)v, m, 16);
        v = o > v ? o : v;
    }
    return v;
}

// The exact summary of a cell the fast path of xs_summarize has flagged (an element of 2^45 ulps or more, an exact tie,
// or the predicted mantissa leaves the binade): everything in 64-bit integers, the special element spelt out if there
// is exactly one.  Two builds: rolled loops and out of line for the streaming kernel (the rare path must not set the
// register budget of the kernel that holds it), unrolled and inline for sums of a few hundred cells, where one workgroup's
// latency is the kernel's and a good part of the cells is flagged (the head of a sum, one crossing per doubling).
// 64-bit DPP moves inside a row of 16 lanes (no LDS crossbar round trips: the exact routine used to spend most of its
// ~13 us in 75 ds_bpermutes and in classifying every element five times)
// exclusive prefix over the row's lanes (row_shr: lanes without a source lane receive 0)
__device__ __forceinline__ int64_t row_excl_scan_dpp(int64_t v)
{
    int64_t incl = v;
    incl += row_dpp_i64<0x111>(incl);
    incl += row_dpp_i64<0x112>(incl);
    incl += row_dpp_i64<0x114>(incl);
    incl += row_dpp_i64<0x118>(incl);
    return incl - v;
}
// min / max over the row, in every lane (row_ror: every lane has a source lane)
__device__ __forceinline__ int64_t row_min_dpp(int64_t v)
{
    int64_t o;
    o = row_dpp_i64<0x121>(v); v = o < v ? o : v;
    o = row_dpp_i64<0x122>(v); v = o < v ? o : v;
    o = row_dpp_i64<0x124>(v); v = o < v ? o : v;
    o = row_dpp_i64<0x128>(v); v = o < v ? o : v;
    return v;
}
__device__ __forceinline__ int64_t row_max_dpp(int64_t v)
{
    int64_t o;
    o = row_dpp_i64<0x121>(v); v = o > v ? o : v;
    o = row_dpp_i64<0x122>(v); v = o > v ? o : v;
    o = row_dpp_i64<0x124>(v); v = o > v ? o : v;
    o = row_dpp_i64<0x128>(v); v = o > v ? o : v;
    return v;
}

// op: what the LDS rows hold -- 0 the terms themselves, 1 the operand whose SQUARE is the term (r.r), 2 the operand whose
// absolute value is the term (|r|_1): the sums of r keep ONE copy of r in LDS for both.
__device__ __forceinline__ double xs_term_of(double w, int op) { return op == 1 ? mul_rn(w, w) : op == 2 ? fabs(w) : w; }

// Same result as xs_cell_summary (xsum.h) on the cell's 128 elements.  Every element is classified ONCE under the
// assumed binade E (movement and class kept in registers) and the elements behind the special one once more under the
// binade it leaves the sum in; the row scans are DPP moves.
__device__ __forceinline__ xs_summary xs_cell16_exact_impl(const double *tl, int32_t E, int32_t neg, int64_t Mt, int op)
{
    double  t[8];
    int64_t dk[8];
    int     cls8 = 0;  // 2 bits per element
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = xs_term_of(tl[k], op);
    const int lane = threadIdx.x & 63, sub = lane & 15, row0 = lane & 48;
    int64_t   T = 0, lo_l = 0, hi_l = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int cls = xs_classify(t[k], E, neg, &dk[k]);
        cls8 |= cls << (2 * k);
        if (cls) dk[k] = 0;  // (a tie's floor is recomputed where it is needed)
        T += dk[k];
        lo_l = T < lo_l ? T : lo_l;
        hi_l = T > hi_l ? T : hi_l;
    }
    const bool    hard = (cls8 & 0xaaaa) != 0, tie = (cls8 & 0x5555) != 0;
    const int64_t base = row_excl_scan_dpp(T);
    int           kx = 8, ty = 0;
    int64_t       run = base;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int cls = (cls8 >> (2 * k)) & 3;
        if (kx == 8 && cls == 1) {
            kx = k;
            ty = 1;
        } else if (kx == 8 && cls == 0 && Mt + run + dk[k] > XS_TWO53 - 2) {
            kx = k;
            ty = 2;
        }
        run += dk[k];
    }
    const unsigned hard_row = (unsigned)(__ballot(hard) >> row0) & 0xffffu, tie_row = (unsigned)(__ballot(tie) >> row0) & 0xffffu;
    const unsigned spec_row = (unsigned)(__ballot(kx < 8) >> row0) & 0xffffu;
    bool           valid = !hard_row && spec_row;
    // the plain summary: what the cell is if it has no (single) special element
    const xs_summary plain = xs_regular(shfl_i64(base + T, row0 + 15), row_min_dpp(base + lo_l), row_max_dpp(base + hi_l), E, neg,
                                        hard_row || tie_row);
    const int      lstar = spec_row ? (int)__builtin_ctz(spec_row) : 0, src = row0 + lstar;
    const int      kstar = __shfl(kx, src, 64), type = __shfl(ty, src, 64);
    // part 1: everything before the special element, in binade E (nothing before the first special element is one)
    int64_t lo1 = 0, hi1 = 0;
    run = base;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (sub < lstar || (sub == lstar && k < kstar)) {
            run += dk[k];
            lo1 = run < lo1 ? run : lo1;
            hi1 = run > hi1 ? run : hi1;
        }
    const int64_t P = shfl_i64(run, src);
    lo1 = row_min_dpp(lo1);
    hi1 = row_max_dpp(hi1);
    // the special element
    double ts = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) ts = k == kstar ? t[k] : ts;
    const double  tau = ldexp(neg ? -ts : ts, 1075 - E), cfl = floor(tau);
    const int64_t c = shfl_i64((int64_t)cfl, src);
    const int     whole = __shfl((int)(tau == cfl), src, 64);
    if (type == 2 && whole) valid = false;  // integer tau at a crossing: a tie in the new binade when W + c is odd
    // part 2: everything after it, in the binade it leaves the sum in
    const int32_t E2 = type == 2 ? E + 1 : E;
    int64_t       T2 = 0, lo2 = 0, hi2 = 0;
    bool          bad2 = false;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (sub > lstar || (sub == lstar && k > kstar)) {
            int64_t d2;
            if (xs_classify(t[k], E2, neg, &d2)) bad2 = true, d2 = 0;
            T2 += d2;
            lo2 = T2 < lo2 ? T2 : lo2;   // (relative to the lane's own start: base2 is added below)
            hi2 = T2 > hi2 ? T2 : hi2;
        }
    const int64_t base2 = row_excl_scan_dpp(T2);
    const int64_t D2 = shfl_i64(base2 + T2, row0 + 15);
    // prefixes inside a lane start at base2: min over (base2 + prefix) with the empty prefix 0 of the whole part included
    lo2 = row_min_dpp(sub >= lstar ? base2 + lo2 : 0);
    hi2 = row_max_dpp(sub >= lstar ? base2 + hi2 : 0);
    lo2 = lo2 < 0 ? lo2 : 0;
    hi2 = hi2 > 0 ? hi2 : 0;
    if ((unsigned)(__ballot(bad2) >> row0) & 0xffffu) valid = false;  // a second special element
    if (!valid) return plain;
    return type == 2 ? xs_crossing(P, lo1, hi1, c, D2, lo2, hi2, E, neg, false) : xs_tie(P, lo1, hi1, c, D2, lo2, hi2, E, neg, false);
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

// Summaries that a workgroup on another XCD reads in the SAME launch (the fused kernel, below): agent-scope stores write
// through the XCD's L2, agent-scope loads do not trust it; no fence (an agent-scope release per workgroup writes back the
// whole L2 and doubled the kernel's time).
__device__ __forceinline__ void xs_store_words_agent(unsigned long long *dst, const unsigned long long *src, int words)
{
    for (int i = threadIdx.x; i < words; i += blockDim.x) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Every vector-memory store this wave has issued so far has been acknowledged -- for the write-through (agent-scope,
// sc1) stores above: by the memory side, not by the XCD's L2.  A workgroup-scope release emits no such wait on gfx950, and
// an agent-scope release fence would add a write-back of the whole L2 (see above); the count that follows is only ever
// added after every wave of the workgroup has passed this wait and a barrier.
__device__ __forceinline__ void xs_stores_acknowledged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The driver's side: a flag / count was read with an agent-scope load; the agent-scope loads of what it guards must not
// be moved above it by the compiler (the hardware issues them in order, after the value they depend on has arrived).
__device__ __forceinline__ void xs_compiler_acquire() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ xs_summary xs_load_agent(const xs_summary *src)
{
    const unsigned long long *d = reinterpret_cast<const unsigned long long *>(src);
    xs_summary                q;
    q.A = (int64_t)__hip_atomic_load(d + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    q.C = (int64_t)__hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    q.Mlo = (int64_t)__hip_atomic_load(d + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    q.Mhi = (int64_t)__hip_atomic_load(d + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long em = __hip_atomic_load(d + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    q.E = (int32_t)(uint32_t)em;
    q.meta = (int32_t)(uint32_t)(em >> 32);
    return q;
}
static_assert(sizeof(xs_summary) == 40, "xs_summary is read and written as five 64-bit words");
constexpr unsigned XS_BATCH_READY = 1u << 16;
constexpr int      XS_DIRECT_BATCHES = 6;  // few cells, up to this many batches: nobody joins them, the walk takes their cells as they are counted  // added to a batch's count of summarised groups once its own summary is in memory
// word `k` (0..4, one per lane) of lane 0's summary
__device__ __forceinline__ unsigned long long readlane_words(const xs_summary &q, int k)
{
    const unsigned long long w[5] = {(unsigned long long)q.A, (unsigned long long)q.C, (unsigned long long)q.Mlo, (unsigned long long)q.Mhi,
                                     (unsigned long long)(uint32_t)q.E | ((unsigned long long)(uint32_t)q.meta << 32)};
    unsigned long long       r = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)w[i]);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w[i] >> 32));
        if (k == i) r = ((unsigned long long)hi << 32) | lo;
    }
    return r;
}

__device__ __noinline__ xs_summary xs_cell16_exact(const double *t, int32_t E, int32_t neg, int64_t Mt, int op)
{
    return xs_cell16_exact_impl(t, E, neg, Mt, op);
}

// One workgroup per group.  16 lanes share a cell (8 consecutive elements each: a short serial prefix in registers,
// then a 4-step tree inside the 16 lanes), so a wave summarises 4 cells at a time and the workgroup 16.  Inside a
// cell no saturation is needed: 128 elements of |d| < 2^52 stay below 2^59.
template <int NS>
struct xsm_shared {
    double     start[2][XS_GROUP];  // predicted running sum at the start of each cell
    xs_summary sm[2][XS_GROUP];
    double     red[4];
    __attribute__((aligned(16))) double tr[NS][4][64 * 10];  // per slot: 4 waves x 64 rows of 8 terms (+2 of padding)
};

// `bid` = the workgroup's number among those that summarise.  FUSED: the driver runs in the same launch (xs_fused): the
// summaries are written through to memory and the group is counted in ready[slot][batch of the group].
template <int KIND, bool FLAT, bool FUSED>
__device__ __forceinline__ void xs_summarize_body(xsm_shared<1> &W, const unsigned bid, int64_t n,
                                                  const double *__restrict__ a, const double *__restrict__ b,
                                                  const double *__restrict__ c, int64_t ncells, int64_t ngroups,
                                                  const double *__restrict__ cellsum, const double *__restrict__ groupsum,
                                                  const double *__restrict__ gprefix, xs_summary *__restrict__ cells,
                                                  xs_summary *__restrict__ groups, const double *__restrict__ offsets,
                                                  unsigned *__restrict__ ready, xs_summary *__restrict__ batches, int64_t nbatch,
                                                  const int head_pass = -1, unsigned *__restrict__ head_count = nullptr)
{
    constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1;  // slots in use: S0..S1
    auto &start = W.start;
    auto &sm = W.sm;
    auto &red = W.red;
    auto &tr = W.tr;
    // the sums of r (r.r and |r|_1 at once) transpose r ITSELF through LDS, once, and form both terms after the transposition
    constexpr bool RAW = KIND == XS_BOTH_SAME;
    const int     lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // groups == NULL ("flat", few cells): four workgroups per group, one pass of 16 cells each, no group summary --
    // the whole problem is then a handful of workgroups, and their serial depth is what counts
    constexpr bool flat = FLAT;  // (groups == NULL)
    // head_pass >= 0 (fused launch, group level): this is one of the FOUR workgroups that share group 0, one pass each.
    // The head of a sum is where the running sum doubles every few elements: several of group 0's cells take the exact
    // integer routine, a workgroup that runs the four passes one after the other needs 35 us for them where the others
    // need 15 (tools/trace_wgs.py) -- and the walk can only start once group 0 is in.
    const bool    head = !flat && head_pass >= 0;
    const int64_t g = flat ? bid >> 2 : (head ? 0 : bid);
    const int sub = threadIdx.x & 15;  // position inside the cell's 16 lanes
    const int pass_lo = flat ? (int)(bid & 3) : (head ? head_pass : 0), pass_hi = flat || head ? pass_lo + 1 : 4;
    // Loads are coalesced -- load j of a lane fetches elements 32 j + 2 sub, + 1 of its cell, so 16 lanes read 256
    // contiguous bytes (8 consecutive elements per lane, the layout the scan wants, made every load touch 64 different
    // lines: 3.8 TB/s at best) -- and the terms go through LDS to reach the lane that scans them: lane r of the cell owns
    // elements 8 r .. 8 r + 7, rows of 10 doubles keep the 16-byte accesses conflict-free.  The next pass's loads are
    // issued before this pass is worked on.
    double *const mine = tr[0][wave] + (threadIdx.x & 48) * 10;  // this cell's 16 rows (slot 0)
    double2       u[4], v[4];
    auto fetch = [&](int pass) {
        const int64_t cell = g * XS_GROUP + pass * 16 + (threadIdx.x >> 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) xs_raw<KIND>(cell < ncells ? n : 0, a, b, c, cell * XS_CELL + 32 * j + 2 * sub, u[j], v[j]);
    };
    fetch(pass_lo);  // (first: these loads do not depend on the prediction below, whose own loads they would otherwise wait behind)
    // approximate sum of everything before this group: from the scan kernel, or (few groups) summed here
    double before[2] = {0.0, 0.0};
#pragma unroll
    for (int slot = S0; slot <= S1; ++slot)
        before[slot] = gprefix    ? gprefix[slot * ngroups + g]
                       : groupsum ? block_sum_partials(groupsum + slot * ngroups, (int)g, 1, red)
                                  : block_sum_partials(cellsum + slot * ncells, (int)(g * XS_GROUP), 1, red);  // few cells
    // a sum that continues another (the part of one rank of a distributed vector): what came before it, for the prediction
    if (offsets != nullptr) {
#pragma unroll
        for (int slot = S0; slot <= S1; ++slot) before[slot] += offsets[slot];
    }
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
#pragma unroll 1
    for (int pass = pass_lo; pass < pass_hi; ++pass) {
        const int     ci = pass * 16 + (threadIdx.x >> 4);
        const int64_t cell = g * XS_GROUP + ci;
        double        t[2][8];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the previous pass's reads of the rows are over
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double *dst = mine + (4 * j + (sub >> 2)) * 10 + 2 * (sub & 3);
            if constexpr (RAW) {
                *reinterpret_cast<double2 *>(dst) = u[j];
            } else {
                double d[2], l[2];
                xs_raw_terms<KIND>(u[j], v[j], d, l);
                *reinterpret_cast<double2 *>(dst) = S0 == 0 ? make_double2(d[0], d[1]) : make_double2(l[0], l[1]);
            }
        }
        if (pass + 1 < pass_hi) fetch(pass + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const double2 w = *reinterpret_cast<const double2 *>(mine + sub * 10 + k);
            if constexpr (RAW) {
                t[0][k] = mul_rn(w.x, w.x);
                t[0][k + 1] = mul_rn(w.y, w.y);
                t[1][k] = fabs(w.x);
                t[1][k + 1] = fabs(w.y);
            } else {
                t[S0][k] = w.x;
                t[S0][k + 1] = w.y;
            }
        }
#pragma unroll
        for (int slot = S0; slot <= S1; ++slot) {
            int32_t    E, neg;
            const bool usable = xs_binade_of(start[slot][ci], &E, &neg);  // else: predicted 0 / subnormal / non-finite
            // Fast path: the movements d = rint(tau) are added up as doubles.  While every |tau| < 2^45 all partial sums
            // of the cell's 128 elements are integers below 2^52, hence exact, and a double add / min / max is one
            // instruction where the 64-bit integer ones are several.  A cell with a larger element, an exact tie, or whose
            // predicted mantissa leaves the binade is redone exactly, in integers, by xs_cell16_exact.
            int             flag = 0, zeros = 1;
            double          Dd = 0.0, lod = 0.0, hid = 0.0;
            const int       sh = 1075 - (usable ? E : 1075);
            const long long flip = (long long)(neg & 1) << 63;  // tau = sigma * t * 2^sh: the sign goes in with one xor
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double x = t[slot][k];
                zeros &= (int)(x == 0.0);
                const double tau = ldexp(__longlong_as_double(__double_as_longlong(x) ^ flip), sh);
                const double d = rint(tau);
                // 2^45 (also NaN, inf); exact tie.  (Bitwise on purpose: no branches around single instructions.)
                flag |= (int)!(fabs(tau) < 35184372088832.0) | (int)(fabs(tau - d) == 0.5);
                Dd += d;
                lod = fmin(lod, Dd);
                hid = fmax(hid, Dd);
            }
            // join over the cell's 16 lanes (one DPP row): lane i takes lane i + off, valid where that lane exists
#define XS_ROW_STEP(CTRL, OFF)                                                                                   \
    {                                                                                                            \
        const double oD = dpp_f64<CTRL>(Dd), olo = dpp_f64<CTRL>(lod), ohi = dpp_f64<CTRL>(hid);                 \
        if (sub + OFF < 16) {                                                                                    \
            lod = fmin(lod, Dd + olo);                                                                           \
            hid = fmax(hid, Dd + ohi);                                                                           \
            Dd += oD;                                                                                            \
        }                                                                                                        \
    }
            XS_ROW_STEP(0x101, 1)  // row_shl:1
            XS_ROW_STEP(0x102, 2)
            XS_ROW_STEP(0x104, 4)
            XS_ROW_STEP(0x108, 8)
#undef XS_ROW_STEP
            const bool cell_flag = ((__ballot(flag != 0) >> (lane & 48)) & 0xffffull) != 0;
            // A cell of zeros only leaves any running sum as it is (s + (+-0) = s; the walk starts at +0 and never holds
            // -0), whatever binade was predicted -- and none can be predicted while everything before it is zero too.
            // CG on a right-hand side with small support (the OpenFOAM systems) keeps such prefixes for hundreds of
            // iterations: without this every cell of them went element by element (foam-32k: 158 of 250 cells).
            const bool cell_is_zero = ((__ballot(zeros != 0) >> (lane & 48)) & 0xffffull) == 0xffffull;
            xs_summary q = xs_identity();
            if (cell < ncells && !cell_is_zero) {
                if (!usable)
                    q = xs_regular(0, 0, 0, 0, 0, true);  // predicted 0 / subnormal / non-finite: element by element
                else if (!cell_flag)
                    q = xs_regular((int64_t)Dd, (int64_t)lod, (int64_t)hid, E, neg, false);  // (exact; sub == 0 has them)
            }
            const int64_t Mt = xs_mantissa(start[slot][ci]);
            bool          want = cell < ncells && !cell_is_zero && usable && (cell_flag || Mt > q.Mhi);  // (sub == 0 knows)
            const unsigned long long want_rows = __ballot(want && sub == 0);
            want = (want_rows >> (lane & 48)) & 1;
            if (want_rows) {  // whole waves enter; ~ln(n) ties and ~log2(n) crossings per sum, so almost none do
                const double    *row = mine + sub * 10;
                const int        op = RAW ? (slot == 0 ? 1 : 2) : 0;
                const xs_summary ex = FLAT ? xs_cell16_exact_impl(row, E, neg, Mt, op) : xs_cell16_exact(row, E, neg, Mt, op);
                if (want) q = ex;
            }
            if (sub == 0) {
                if (!FUSED && cell < ncells) cells[slot * ncells + cell] = q;
                sm[slot][ci] = q;
            }
        }
    }
    if constexpr (FLAT && FUSED) {
        // few cells, one launch: a batch is the 64 cells of one group, summarised by four workgroups of 16 cells each
        __syncthreads();
        const int     pass = (int)(bid & 3);
        const int64_t c0 = g * XS_GROUP + pass * 16;
        const int     ccnt = (int)(ncells - c0 < 16 ? (ncells - c0 < 0 ? 0 : ncells - c0) : 16);
#pragma unroll
        for (int slot = S0; slot <= S1; ++slot)
            xs_store_words_agent(reinterpret_cast<unsigned long long *>(cells + slot * ncells + c0),
                                 reinterpret_cast<const unsigned long long *>(&sm[slot][pass * 16]), ccnt * 5);
        xs_stores_acknowledged();  // every wave's write-through stores are in memory ...
        __syncthreads();            // ... before the workgroup is counted
        int *last = reinterpret_cast<int *>(&red[0]);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int slot = S0; slot <= S1; ++slot) {
                const unsigned old = __hip_atomic_fetch_add(ready + (slot - S0) * nbatch + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last[slot] = old == 3u;
            }
        }
        __syncthreads();
        if (nbatch > XS_DIRECT_BATCHES && wave >= S0 && wave <= S1 && last[wave]) {
            const int  gc = (int)(ncells - g * XS_GROUP < XS_GROUP ? ncells - g * XS_GROUP : XS_GROUP);
            xs_summary q = xs_identity();
            if (lane < gc) q = xs_load_agent(cells + wave * ncells + g * XS_GROUP + lane);
            const xs_summary j = wave_join(q);
            if (lane < 5)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(batches + (wave - S0) * nbatch + g) + lane, readlane_words(j, lane),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xs_stores_acknowledged();  // (one wave: the batch's summary is in memory before its flag goes up)
            if (lane == 0)
                __hip_atomic_fetch_add(ready + (wave - S0) * nbatch + g, XS_BATCH_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (flat) return;
    __syncthreads();
    if constexpr (!FUSED) {
        if (wave >= S0 && wave <= S1) {
            const xs_summary q = wave_join(sm[wave][lane]);
            if (lane == 0) groups[wave * ngroups + g] = q;
        }
    } else {
        // the group's summaries go into the start[] rows (dead since the passes), then the 64 cell summaries and the
        // group's are written through in one coalesced sweep
        xs_summary *gq = reinterpret_cast<xs_summary *>(&start[0][0]);  // [2] summaries in 1 KiB
        int        *last = reinterpret_cast<int *>(&red[0]);            // (red: 4 doubles, dead by now) [0..1] head, [2..3] batch
        if (!head) {
            if (wave >= S0 && wave <= S1) {
                const xs_summary q = wave_join(sm[wave][lane]);
                if (lane == 0) gq[wave] = q;
            }
            __syncthreads();
            const int ccnt = (int)(ncells - g * XS_GROUP < XS_GROUP ? ncells - g * XS_GROUP : XS_GROUP);
#pragma unroll
            for (int slot = S0; slot <= S1; ++slot) {
                xs_store_words_agent(reinterpret_cast<unsigned long long *>(cells + slot * ncells + g * XS_GROUP),
                                     reinterpret_cast<const unsigned long long *>(&sm[slot][0]), ccnt * 5);
                if (threadIdx.x < 5)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(groups + slot * ngroups + g) + threadIdx.x,
                                       reinterpret_cast<const unsigned long long *>(gq + slot)[threadIdx.x], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            xs_stores_acknowledged();  // every wave's write-through stores have been acknowledged by memory ...
            __syncthreads();
        } else {
            // one of the four workgroups of group 0 (a full group: this path has more than 2048 cells): its 16 cell
            // summaries go out; the last of the four reads all 64 back, joins them into the group's summary and goes on
            // as the workgroup of a whole group does
#pragma unroll
            for (int slot = S0; slot <= S1; ++slot)
                xs_store_words_agent(reinterpret_cast<unsigned long long *>(cells + slot * ncells + head_pass * 16),
                                     reinterpret_cast<const unsigned long long *>(&sm[slot][head_pass * 16]), 16 * 5);
            xs_stores_acknowledged();
            __syncthreads();
            if (threadIdx.x == 0) {
#pragma unroll
                for (int slot = S0; slot <= S1; ++slot) {
                    const unsigned old = __hip_atomic_fetch_add(head_count + (slot - S0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last[slot] = old == 3u;
                    if (old == 3u) __hip_atomic_store(head_count + (slot - S0), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
                }
            }
            __syncthreads();
            if (wave >= S0 && wave <= S1 && last[wave]) {
                const xs_summary q = xs_load_agent(cells + wave * ncells + lane);
                const xs_summary j = wave_join(q);
                if (lane < 5)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(groups + wave * ngroups) + lane, readlane_words(j, lane),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                xs_stores_acknowledged();  // (one wave: group 0's summary is in memory before the group is counted)
            }
            __syncthreads();
        }
        const int64_t bb = g >> 6;
        const int     gcnt = (int)(ngroups - bb * 64 < 64 ? ngroups - bb * 64 : 64);
        if (threadIdx.x == 0) {                                  // ... before the group is counted
#pragma unroll
            for (int slot = S0; slot <= S1; ++slot) {
                last[2 + slot] = 0;
                if (head && !last[slot]) continue;               // (group 0 is counted by the last of its four workgroups)
                const unsigned old = __hip_atomic_fetch_add(ready + (slot - S0) * nbatch + bb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last[2 + slot] = old == (unsigned)(gcnt - 1);
            }
        }
        __syncthreads();
        // The workgroup that completes a batch of 64 groups joins their summaries into the batch's (they are all in memory:
        // every group was written through before it was counted) and raises the batch's flag for the driver.
        if (wave >= S0 && wave <= S1 && last[2 + wave]) {
            xs_summary q = xs_identity();
            if (lane < gcnt) q = xs_load_agent(groups + wave * ngroups + bb * 64 + lane);
            const xs_summary j = wave_join(q);
            if (lane < 5)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(batches + (wave - S0) * nbatch + bb) + lane,
                                   readlane_words(j, lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xs_stores_acknowledged();  // (one wave: the batch's summary is in memory before its flag goes up)
            if (lane == 0)
                __hip_atomic_fetch_add(ready + (wave - S0) * nbatch + bb, XS_BATCH_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int KIND, bool FLAT>
__global__ __launch_bounds__(256) void xs_summarize(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                   const double *__restrict__ c, int64_t ncells, int64_t ngroups,
                                                   const double *__restrict__ cellsum, const double *__restrict__ groupsum,
                                                   const double *__restrict__ gprefix, xs_summary *__restrict__ cells,
                                                   xs_summary *__restrict__ groups, const double *__restrict__ offsets)
{
    __shared__ xsm_shared<1> W;
    xs_summarize_body<KIND, FLAT, false>(W, blockIdx.x, n, a, b, c, ncells, ngroups, cellsum, groupsum, gprefix, cells, groups, offsets,
                                         nullptr, nullptr, 0);
}

// ---------------------------------------------------------------------------------------------- the driver
// Optional profile of the driver (tools/prof_drive.py): lane 0 of the first sum's waves adds event counts and 100 MHz
// ticks into 16 words.  [0] batches the walk descended into, [1] groups walked cell by cell, [2] cells added element by
// element, [3] ticks in total, [4] ticks in phase 1 (staging, batch joins), [5] ticks in phase 2 (the walk), [6] launches,
// [7] runs with a crossing or a tie applied whole, [8] prefix scans, [9] calls of xs_batch, [10] ticks before phase 1.
#define XS_PROF_ADD(k, v)                                                                                   \
    do {                                                                                                    \
        if (prof != nullptr && (threadIdx.x & 63) == 0)                                                     \
            __hip_atomic_fetch_add(&prof[k], (long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   \
    } while (0) /* no-return atomics: nothing waits for them, so counting does not stretch what it counts */
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

// Inclusive prefix over the lanes of the movements of a batch's plain runs (K = 0, no tie, not irregular; the others
// count 0), with DPP moves (no LDS crossbar round trips on the serial path).  It does not depend on the running sum, so
// one scan serves every stop of the walk inside the batch.  Plain wrapping adds: differences over a stretch of valid
// runs are exact, which is all that is ever used.
__device__ __forceinline__ bool xs_plain(const xs_summary &q) { return (q.meta & (XS_IRREGULAR | XS_TIE | 0xff00)) == 0; }
__device__ __forceinline__ int64_t xs_scan(const xs_summary &q, int cnt)
{
    int64_t incl = (int)(threadIdx.x & 63) < cnt && xs_plain(q) ? q.A : 0;
    incl += dpp_i64<0x111, 0xf>(incl);  // row_shr:1
    incl += dpp_i64<0x112, 0xf>(incl);  // row_shr:2
    incl += dpp_i64<0x114, 0xf>(incl);  // row_shr:4
    incl += dpp_i64<0x118, 0xf>(incl);  // row_shr:8   -> prefix inside each row of 16 lanes
    incl += dpp_i64<0x142, 0xa>(incl);  // row_bcast:15 into rows 1 and 3
    incl += dpp_i64<0x143, 0xc>(incl);  // row_bcast:31 into rows 2 and 3
    return incl;
}

// Items [start, cnt) of a batch (one summary per lane, incl = xs_scan of the batch): apply the longest valid prefix
// to the running sum s (wave-uniform) and return the index of the first item that does not apply (cnt if all did).
// Plain runs are validated and applied together; a run that holds binade crossings or a tie stops the stretch, is
// applied by itself, and the next stretch starts behind it.
__device__ __forceinline__ int xs_batch(const xs_summary &q, int64_t incl, int start, int cnt, double &s, long long *prof = nullptr)
{
    const int     lane = threadIdx.x & 63;
    const bool    plain = xs_plain(q);
    const int64_t excl = incl - (plain ? q.A : 0);
    XS_PROF_ADD(9, 1);
    for (;;) {
        XS_PROF_ADD(8, 1);
        int32_t    E0, neg0;
        const bool active = lane >= start && lane < cnt;
        if (!xs_binade_of(s, &E0, &neg0)) {
            // no binade (0, subnormal, inf, nan): only empty runs may be skipped
            const unsigned long long fail = __ballot(active && !xs_is_identity(q));
            return fail ? (int)__builtin_ctzll(fail) : cnt;
        }
        const int64_t            M = xs_mantissa(s);
        const int64_t            base = start > 0 ? readlane_i64(incl, start - 1) : 0;  // (lanes before `start` hold 0 or plain runs)
        const bool               ok = !active || (plain && xs_applies(q, E0, neg0, M + (excl - base)));
        const unsigned long long fail = __ballot(!ok);
        const int                f = fail ? (int)__builtin_ctzll(fail) : cnt;
        int64_t                  Mf = M;  // the mantissa item f meets
        if (f > start) Mf = M + (readlane_i64(incl, f - 1) - base);
        if (f > start) s = xs_compose(E0, neg0, Mf);
        if (f >= cnt) return cnt;
        const int metaf = __builtin_amdgcn_readlane(q.meta, f);
        if ((metaf & XS_IRREGULAR) || !(metaf & (XS_TIE | 0xff00))) return f;
        xs_summary qf;
        qf.meta = metaf;
        qf.E = __builtin_amdgcn_readlane(q.E, f);
        qf.Mlo = readlane_i64(q.Mlo, f);
        qf.Mhi = readlane_i64(q.Mhi, f);
        if (!xs_applies(qf, E0, neg0, Mf)) return f;
        qf.A = readlane_i64(q.A, f);
        qf.C = readlane_i64(q.C, f);
        s = xs_apply(qf, E0, neg0, Mf);
        XS_PROF_ADD(7, 1);
        start = f + 1;
        if (start >= cnt) return cnt;
    }
}

// s + (the 128 elements of one cell, staged in LDS), left to right.  Every lane carries the same s.  The reads are
// broadcasts that do not depend on the additions, so they run ahead of the one dependent chain (as chain_add of
// seq_sums, cg_kernels.h).
__device__ __forceinline__ double xs_chain(double s, const double *stage)
{
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

// trace of one walk (profile buffer words 16..): (event code, 100 MHz tick) pairs, plain stores nobody waits for
#define XS_TRACE(code)                                                                   \
    do {                                                                                 \
        if (prof != nullptr && (threadIdx.x & 63) == 0 && ntrace < 240) {                \
            prof[16 + 2 * ntrace] = (code);                                              \
            prof[17 + 2 * ntrace] = wall_clock64() - tk0;                                \
        }                                                                                \
        ++ntrace;                                                                        \
    } while (0)

// ---- The driver: one workgroup of 16 waves per sum.
// Round = up to XD_ITEMS items (group summaries; cell summaries when there is no group level).
//   Phase 1, all waves: a wave takes every 16th batch of 64 items -- copies them into LDS, joins them into ONE summary
//   of the batch, and for the items flagged irregular fetches what the walk will need when it gets there: the 64 cell
//   summaries of such a group and the 128 elements of such a cell (three dependent trips to memory, taken by 16 waves
//   at once instead of by the walk, one after the other).
//   Phase 2, wave 0: the walk.  Batch summaries (<= 32, one per lane) are validated and applied with one prefix sum;
//   the walk descends into a batch that does not apply, from there into a group, from there into the elements of a
//   cell -- all of it in LDS unless a pool overflowed.
constexpr int XD_THREADS = 1024, XD_WAVES = XD_THREADS / 64;
constexpr int XD_ITEMS = 2048;  // items per round
constexpr int XD_POOL = 12;     // irregular groups whose cells are staged ahead
constexpr int XD_CHAINS = 16;   // irregular cells whose elements are staged ahead
constexpr int XD_DIRECT = 6;    // up to this many batches the walk takes them directly (no batch summaries)

struct xd_shared {
    xs_summary G[XD_ITEMS];
    xs_summary B[XD_ITEMS / 64];
    xs_summary pool[XD_POOL][XS_GROUP];
    double     chain[XD_CHAINS][XS_CELL];
    double     stage[XS_CELL];
    int64_t    pool_gid[XD_POOL];
    int64_t    chain_cell[XD_CHAINS];
    int        pool_n, chain_n;
};

// elements of a cell the walk will add one by one: into the pool, if there is room
template <int KIND>
__device__ __forceinline__ void xd_stage_chain(xd_shared &S, int slot, int64_t n, const double *__restrict__ a,
                                               const double *__restrict__ b, const double *__restrict__ c, int64_t cell)
{
    const int lane = threadIdx.x & 63;
    int       id = 0;
    if (lane == 0) id = atomicAdd(&S.chain_n, 1);
    id = __builtin_amdgcn_readfirstlane(id);
    if (id >= XD_CHAINS) return;
    double t0, t1;
    xs_cell_terms<KIND>(slot, n, a, b, c, cell, t0, t1);
    if (lane == 0) S.chain_cell[id] = cell;
    *reinterpret_cast<double2 *>(&S.chain[id][2 * lane]) = make_double2(t0, t1);
}

// s + cell, element by element (wave 0)
template <int KIND>
__device__ __forceinline__ double xd_chain(xd_shared &S, double s, int slot, int64_t n, const double *__restrict__ a,
                                           const double *__restrict__ b, const double *__restrict__ c, int64_t cell)
{
    const int                lane = threadIdx.x & 63;
    const int                have = S.chain_n < XD_CHAINS ? S.chain_n : XD_CHAINS;
    const unsigned long long hit = __ballot(lane < have && S.chain_cell[lane] == cell);
    if (hit) return xs_chain(s, S.chain[__builtin_ctzll(hit)]);
    double t0, t1;
    xs_cell_terms<KIND>(slot, n, a, b, c, cell, t0, t1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the previous cell's reads are over
    *reinterpret_cast<double2 *>(S.stage + 2 * lane) = make_double2(t0, t1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return xs_chain(s, S.stage);
}

// the sum of slot `slot` (its result is wave 0's; call with the whole workgroup)
template <int KIND>
__device__ double xd_sum(xd_shared &S, int slot, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                         const double *__restrict__ c, int64_t ncells, int64_t ngroups, const xs_summary *__restrict__ cells,
                         const xs_summary *__restrict__ groups, double s0, long long *prof)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    cells += slot * ncells;
    if (groups) groups += slot * ngroups;
    const xs_summary *items = groups ? groups : cells;
    const int64_t     nitems = groups ? ngroups : ncells;
    double            s = s0;
    const long long   tk0 = prof ? wall_clock64() : 0;
    int               ntrace = 0;
    for (int64_t base = 0; base < nitems; base += XD_ITEMS) {
        const int cnt = (int)(nitems - base < XD_ITEMS ? nitems - base : XD_ITEMS), nb = (cnt + 63) / 64;
        if (threadIdx.x == 0) S.pool_n = S.chain_n = 0;
        __syncthreads();
        const long long tp0 = prof ? wall_clock64() : 0;
        // ---- phase 1
        const bool top = nb > XD_DIRECT;  // few batches: the walk takes them one by one, nobody joins them
        for (int bb = wave; bb < nb; bb += (int)(blockDim.x >> 6)) {
            const int64_t i = base + bb * 64 + lane;
            xs_summary    q = xs_identity();
            if (i < nitems) q = items[i];
            S.G[bb * 64 + lane] = q;
            unsigned long long irr = __ballot((q.meta & XS_IRREGULAR) != 0);
            while (irr) {
                const int64_t it = base + bb * 64 + __builtin_ctzll(irr);
                irr &= irr - 1;
                if (!groups) {
                    xd_stage_chain<KIND>(S, slot, n, a, b, c, it);
                    continue;
                }
                int id = 0;
                if (lane == 0) id = atomicAdd(&S.pool_n, 1);
                id = __builtin_amdgcn_readfirstlane(id);
                if (id >= XD_POOL) continue;
                xs_summary cq = xs_identity();
                if (it * XS_GROUP + lane < ncells) cq = cells[it * XS_GROUP + lane];
                if (lane == 0) S.pool_gid[id] = it;
                S.pool[id][lane] = cq;
                unsigned long long cirr = __ballot((cq.meta & XS_IRREGULAR) != 0);
                while (cirr) {
                    xd_stage_chain<KIND>(S, slot, n, a, b, c, it * XS_GROUP + __builtin_ctzll(cirr));
                    cirr &= cirr - 1;
                }
            }
            if (top) {
                const xs_summary j = wave_join(q);
                if (lane == 0) S.B[bb] = j;
            }
        }
        __syncthreads();
        const long long tp1 = prof ? wall_clock64() : 0;
        if (wave == 0) XS_PROF_ADD(4, tp1 - tp0);
        XS_TRACE(100);
        // ---- phase 2
        if (wave == 0) {
            xs_summary Bq = xs_identity();
            if (top && lane < nb) Bq = S.B[lane];
            const int64_t Binc = top ? xs_scan(Bq, nb) : 0;
            int           tstart = 0;
            while (tstart < nb) {
                const int fb = top ? xs_batch(Bq, Binc, tstart, nb, s, prof) : tstart;
                XS_TRACE(1000 + fb);
                if (fb >= nb) break;
                XS_PROF_ADD(0, 1);
                const xs_summary Gq = S.G[fb * 64 + lane];
                const int        gcnt = cnt - fb * 64 < 64 ? cnt - fb * 64 : 64;
                const int64_t    Ginc = xs_scan(Gq, gcnt);
                int              gstart = 0;
                while (gstart < gcnt) {
                    const int f = xs_batch(Gq, Ginc, gstart, gcnt, s, prof);
                    XS_TRACE(2000 + f);
                    if (f >= gcnt) break;
                    const int64_t it = base + fb * 64 + f;
                    if (!groups) {
                        s = xd_chain<KIND>(S, s, slot, n, a, b, c, it);
                        XS_PROF_ADD(2, 1);
                    } else {
                        XS_PROF_ADD(1, 1);
                        const int                have = S.pool_n < XD_POOL ? S.pool_n : XD_POOL;
                        const unsigned long long hit = __ballot(lane < have && S.pool_gid[lane] == it);
                        const int64_t            c0 = it * XS_GROUP;
                        const int                ccnt = (int)(ncells - c0 < XS_GROUP ? ncells - c0 : XS_GROUP);
                        xs_summary               Cq = xs_identity();
                        XS_TRACE(hit ? 5001 : 5000);
                        if (hit)
                            Cq = S.pool[__builtin_ctzll(hit)][lane];
                        else if (lane < ccnt)
                            Cq = cells[c0 + lane];
                        const int64_t Cinc = xs_scan(Cq, ccnt);
                        int           cstart = 0;
                        while (cstart < ccnt) {
                            const int cf = xs_batch(Cq, Cinc, cstart, ccnt, s, prof);
                            XS_TRACE(3000 + cf);
                            if (cf >= ccnt) break;
                            s = xd_chain<KIND>(S, s, slot, n, a, b, c, c0 + cf);
                            XS_PROF_ADD(2, 1);
                            XS_TRACE(4000 + (int)(s != s));
                            cstart = cf + 1;
                        }
                    }
                    gstart = f + 1;
                }
                tstart = fb + 1;
            }
            XS_TRACE(9000 + (int)(s != s));
            if (prof) XS_PROF_ADD(5, wall_clock64() - tp1 + (long long)(s != s));
            if (prof) XS_PROF_ADD(10, tp0 - tk0);
        }
        __syncthreads();  // the next round refills what the walk read
    }
    return s;
}

// What the sums are for (as seq_sums): mode 0 -> out[0..1]; 1 -> alpha of iteration `iter` from p.Ap; 2 -> residual, r.r,
// beta, convergence flag; 3 -> the initial residual.
__device__ __forceinline__ void xs_finalize(double s_dot, double s_l1, int mode, double *hist, int iter, double tol, int *done,
                                            double *out)
{
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

// Grid = one workgroup per sum in use (KIND 2: r.r and |r|_1 at once); the workgroup that finishes last finalises like
// seq_sums (alpha / beta / residual / done flag).  xfer: [0..1] the sums, [2] tickets (zero between launches).
template <int KIND>
__global__ __launch_bounds__(XD_THREADS) void xs_drive(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                      const double *__restrict__ c, int64_t ncells, int64_t ngroups,
                                                      const xs_summary *__restrict__ cells, const xs_summary *__restrict__ groups,
                                                      int mode, double *hist, int iter, double tol, int *done, double *out,
                                                      double *xfer, const double *s_in, double *s_out, long long *prof_buf)
{
    constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1;
    __shared__ xd_shared S;
    const int  slot = S0 + (int)blockIdx.x;
    // s_out: this is one rank's part of a longer sum -- start from s_in, leave the running sums, do not finalise
    const bool frozen = done != nullptr && (s_out != nullptr || (mode != 3 && mode != 0)) && *done != 0;
    if (frozen) {
        if (s_out != nullptr && threadIdx.x == 0) s_out[slot] = 0.0;
        return;
    }
    long long      *prof = slot == S0 ? prof_buf : nullptr;
    const long long t0 = prof ? wall_clock64() : 0;
    const double    s = xd_sum<KIND>(S, slot, n, a, b, c, ncells, ngroups, cells, groups, s_in ? s_in[slot] : 0.0, prof);
    if (threadIdx.x != 0) return;
    if (prof) {
        XS_PROF_ADD(3, wall_clock64() - t0 + (long long)(s != s));
        XS_PROF_ADD(6, 1);
    }
    if (s_out != nullptr) {
        s_out[slot] = s;
        if (S0 == S1) s_out[1 - slot] = s_in ? s_in[1 - slot] : 0.0;  // the sum not taken here passes through
        return;
    }
    double s_dot = s, s_l1 = s;
    if constexpr (S1 > S0) {
        // two workgroups: the second to arrive has both sums (stores and loads at agent scope around the ticket)
        unsigned long long *tk = reinterpret_cast<unsigned long long *>(xfer + 2);
        __hip_atomic_store(xfer + slot, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long old = __hip_atomic_fetch_add(tk, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) return;
        __hip_atomic_store(tk, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_dot = __hip_atomic_load(xfer + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_l1 = __hip_atomic_load(xfer + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    xs_finalize(KIND == XS_L1 ? 0.0 : s_dot, KIND == XS_DOT ? 0.0 : s_l1, mode, hist, iter, tol, done, out);
}


// ---------------------------------------------------------------------------------------------- summaries and walk in ONE launch
// xs_summarize followed by xs_drive leaves the chip to one workgroup for as long as the summaries took to build (256^3:
// 52 + 55 us for p.Ap).  xs_fused runs both at once: workgroups 0 (and 1, for the second sum) are DRIVERS, the others
// summarise one group each, in launch order, and count themselves per batch of 64 groups; a driver takes the batches in
// order as they fill up, so the walk ends a few microseconds after the last group's summary.
//   driver = 4 waves: waves 1..3 are STAGERS (stager k owns ring slot k and the batches k, k+3, ...: waits until the
//   walk has left the slot and the batch's groups are all counted, copies the 64 group summaries into LDS, joins them into
//   the batch's summary, and fetches ahead what a descent would need: the cell summaries of a group flagged irregular and
//   the elements of its irregular cells); wave 0 WALKS: per batch one summary to validate and apply, else down into the
//   groups, the cells, the elements, as xd_sum does.
// Nothing waits for a driver, and a driver only waits for workgroups that wait for nothing, so the launch always drains;
// should a count never fill up (it cannot, short of a fault elsewhere) every wait gives up after ~2 s and the sums are NaN.
constexpr long long XR_PATIENCE = 200000000;  // ticks of the 100 MHz clock

constexpr int XR_CHAINS = 20;  // cells whose elements are fetched ahead of the walk (few cells only)

struct xr_shared {
    double stage[XS_CELL];
    double chain[XR_CHAINS][XS_CELL];
    int    chain_cell[XR_CHAINS];  // the cell whose elements chain[k] holds, -1 until they are in
    int    chain_n, walked, abort;
};

__device__ __forceinline__ int xr_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void xr_post(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int KIND>
__device__ __forceinline__ double xr_chain(xr_shared &S, double s, int slot, int64_t n, const double *__restrict__ a,
                                           const double *__restrict__ b, const double *__restrict__ c, int64_t cell)
{
    const int lane = threadIdx.x & 63;
    {   // fetched ahead?
        const int                have = xr_peek(&S.chain_n) < XR_CHAINS ? xr_peek(&S.chain_n) : XR_CHAINS;
        const unsigned long long hit = __ballot(lane < have && xr_peek(&S.chain_cell[lane < XR_CHAINS ? lane : 0]) == (int)cell);
        if (hit) return xs_chain(s, S.chain[__builtin_ctzll(hit)]);
    }
    double t0, t1;
    xs_cell_terms<KIND>(slot, n, a, b, c, cell, t0, t1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the previous cell's reads are over
    *reinterpret_cast<double2 *>(S.stage + 2 * lane) = make_double2(t0, t1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return xs_chain(s, S.stage);
}

// Few cells: the launch is short and the walk's descents are its critical path, each a round trip to memory.  Waves 1..3 of
// the driver run ahead of the walk: wave w takes the batches w-1, w+2, ... as their flags go up, looks for cells flagged
// irregular (to be added element by element) and fetches their elements into LDS.  Best effort: the walk fetches what it
// does not find.
template <int KIND>
__device__ void xr_fetch_ahead(xr_shared &S, int slot, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                               const double *__restrict__ c, int64_t ncells, int nb, const xs_summary *__restrict__ cells,
                               const unsigned *__restrict__ ready)
{
    const int      lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ready_at = nb <= XS_DIRECT_BATCHES ? 4u : XS_BATCH_READY;
    cells += slot * ncells;
    for (int bb = w - 1; bb < nb; bb += 3) {
        const long long t0 = wall_clock64();
        for (;;) {
            if (xr_peek(&S.walked) > bb || xr_peek(&S.abort) || wall_clock64() - t0 > XR_PATIENCE) break;
            unsigned f = 0;
            if (lane == 0) f = __hip_atomic_load(ready + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)f) >= ready_at) break;
            __builtin_amdgcn_s_sleep(4);
        }
        xs_compiler_acquire();
        if (xr_peek(&S.walked) > bb) continue;
        if (xr_peek(&S.abort) || wall_clock64() - t0 > XR_PATIENCE) return;
        const int64_t c0 = (int64_t)bb * XS_GROUP;
        const int     ccnt = (int)(ncells - c0 < XS_GROUP ? ncells - c0 : XS_GROUP);
        xs_summary    Cq = xs_identity();
        if (lane < ccnt) Cq = xs_load_agent(cells + c0 + lane);
        unsigned long long irr = __ballot((Cq.meta & XS_IRREGULAR) != 0);
        while (irr) {
            const int64_t cell = c0 + __builtin_ctzll(irr);
            irr &= irr - 1;
            int id = 0;
            if (lane == 0) id = atomicAdd(&S.chain_n, 1);
            id = __builtin_amdgcn_readfirstlane(id);
            if (id >= XR_CHAINS) return;
            double t0e, t1e;
            xs_cell_terms<KIND>(slot, n, a, b, c, cell, t0e, t1e);
            *reinterpret_cast<double2 *>(&S.chain[id][2 * lane]) = make_double2(t0e, t1e);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) xr_post(&S.chain_cell[id], (int)cell);
        }
    }
}

// the sum of slot `slot` while the summaries are still being built (one wave).  FLAT: no group level, a batch is 64 cells.
template <int KIND, bool FLAT>
__device__ double xr_sum(xr_shared &S, int slot, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                         const double *__restrict__ c, int64_t ncells, int64_t ngroups, const xs_summary *__restrict__ cells,
                         const xs_summary *__restrict__ groups, const xs_summary *__restrict__ batches, unsigned *__restrict__ ready,
                         long long *trace)
{
    const int lane = threadIdx.x & 63;
    // optional time stamps (tools/trace_fused.py; 100 MHz ticks): [0] start, [8 + b] batch b seen ready, [152 + b] walked
    const long long tk0 = trace ? wall_clock64() : 0;
    if (trace && lane == 0) trace[0] = tk0;
#define XR_STAMP(base, b)                                                                          \
    do {                                                                                           \
        if (trace != nullptr && lane == 0 && (b) < 72) trace[(base) + (b)] = wall_clock64() - tk0; \
    } while (0)
    cells += slot * ncells;
    if (!FLAT) groups += slot * ngroups;
    const int nb = FLAT ? (int)ngroups : (int)((ngroups + 63) / 64);
    // few cells and few batches: a batch is ready as soon as its four workgroups are counted, and is walked cell by cell
    const bool     direct = FLAT && nb <= XS_DIRECT_BATCHES;
    const unsigned ready_at = direct ? 4u : XS_BATCH_READY;
    double         s = 0.0;
    int            cur = 0;
    int            head = 0;  // cells at the front of the sum that the walk has already added element by element
    {
        // Until the first batch's summaries are in memory (6.6 us after the start at 32 768 elements, 21 us at 2.1M) the
        // walk has nothing to do, while the front of a sum is where it is slowest (the running sum doubles every few
        // elements: the first cells are the ones that end up being added element by element anyway) -- and up to a few
        // million elements the walk is the launch's critical path.  So the walking wave starts on the first group's cells
        // at once, one cell per 0.7 us with the next cell's elements in flight, and hands over to the summaries as soon
        // as the first batch is ready; the summaries of the cells it got through are then simply not used (xs_batch starts
        // behind them).
        const int ccnt0 = (int)(ncells < XS_GROUP ? ncells : XS_GROUP);
        double    e0, e1;
        xs_cell_terms<KIND>(slot, n, a, b, c, 0, e0, e1);
        while (head < ccnt0) {
            unsigned f = 0;
            if (lane == 0) f = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // examined behind the chain
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the previous cell's reads are over
            *reinterpret_cast<double2 *>(S.stage + 2 * lane) = make_double2(e0, e1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double n0 = 0.0, n1 = 0.0;
            if (head + 1 < ccnt0) xs_cell_terms<KIND>(slot, n, a, b, c, head + 1, n0, n1);
            s = xs_chain(s, S.stage);
            ++head;
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)f) >= ready_at) break;
            e0 = n0;
            e1 = n1;
        }
        xs_compiler_acquire();
    }
    long long t0 = wall_clock64();
    while (cur < nb) {
        // the batches from `cur` on whose summaries are in memory: a prefix of up to 64, taken together
        const int cnt = nb - cur < 64 ? nb - cur : 64;
        unsigned  f = 0;
        if (lane < cnt) f = __hip_atomic_load(ready + cur + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long rdy = __ballot(f >= ready_at);
        xs_compiler_acquire();  // what the flags guard is loaded (agent scope) after them
        const int                m = ~rdy ? (int)__builtin_ctzll(~rdy) : 64;
        if (m == 0) {
            if (wall_clock64() - t0 > XR_PATIENCE) {
                if (lane == 0) xr_post(&S.abort, 1);
                return NAN;
            }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        xs_summary Bq = xs_identity();
        if (lane < m) {
            if (direct || (cur == 0 && lane == 0 && head > 0))
                Bq = xs_regular(0, 0, 0, 0, 0, true);  // "does not apply": every batch is entered / the walk is inside batch 0 already
            else
                Bq = xs_load_agent(batches + cur + lane);
            __hip_atomic_store(ready + cur + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
        }
        for (int q = 0; q < m; ++q) XR_STAMP(8, cur + q);
        const int64_t Binc = xs_scan(Bq, m);
        int           tstart = 0;
        while (tstart < m) {
            const int fb = xs_batch(Bq, Binc, tstart, m, s);
            if (fb >= m) break;
            const int bb = cur + fb;
            if constexpr (FLAT) {
                const int64_t c0 = (int64_t)bb * XS_GROUP;
                const int     ccnt = (int)(ncells - c0 < XS_GROUP ? ncells - c0 : XS_GROUP);
                xs_summary    Cq = xs_identity();
                if (lane < ccnt) Cq = xs_load_agent(cells + c0 + lane);
                const int64_t Cinc = xs_scan(Cq, ccnt);
                int           cstart = bb == 0 ? head : 0;  // (the first `head` cells of the sum are in s already)
                while (cstart < ccnt) {
                    const int cf = xs_batch(Cq, Cinc, cstart, ccnt, s);
                    if (cf >= ccnt) break;
                    s = xr_chain<KIND>(S, s, slot, n, a, b, c, c0 + cf);
                    cstart = cf + 1;
                }
            } else {
                const int     gcnt = (int)(ngroups - (int64_t)bb * 64 < 64 ? ngroups - (int64_t)bb * 64 : 64);
                xs_summary    Gq = xs_identity();
                if (lane < gcnt) Gq = xs_load_agent(groups + (int64_t)bb * 64 + lane);
                if (bb == 0 && lane == 0 && head > 0) Gq = xs_regular(0, 0, 0, 0, 0, true);  // the walk is inside group 0 already
                const int64_t Ginc = xs_scan(Gq, gcnt);
                int           gstart = 0;
                while (gstart < gcnt) {
                    const int f2 = xs_batch(Gq, Ginc, gstart, gcnt, s);
                    if (f2 >= gcnt) break;
                    const int64_t it = (int64_t)bb * 64 + f2, c0 = it * XS_GROUP;
                    const int     ccnt = (int)(ncells - c0 < XS_GROUP ? ncells - c0 : XS_GROUP);
                    xs_summary    Cq = xs_identity();
                    if (lane < ccnt) Cq = xs_load_agent(cells + c0 + lane);
                    const int64_t Cinc = xs_scan(Cq, ccnt);
                    int           cstart = it == 0 ? head : 0;  // (the first `head` cells of the sum are in s already)
                    while (cstart < ccnt) {
                        const int cf = xs_batch(Cq, Cinc, cstart, ccnt, s);
                        if (cf >= ccnt) break;
                        s = xr_chain<KIND>(S, s, slot, n, a, b, c, c0 + cf);
                        cstart = cf + 1;
                    }
                    gstart = f2 + 1;
                }
            }
            XR_STAMP(152, bb);
            tstart = fb + 1;
        }
        cur += m;
        if (lane == 0) xr_post(&S.walked, cur);
        XR_STAMP(152, cur - 1);
        t0 = wall_clock64();
    }
    return s;
}

// x[i] += alpha p[i] for the rows of four groups (4 x 8192), alpha of the iteration in flight from the history: the
// arithmetic of cg_update_xp (reference vec_axpy, vec.c:88-94: rounded product, rounded sum)
typedef double xd2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void xs_update_x(int64_t n, double *__restrict__ x, const double *__restrict__ p, const double *hist, int iter,
                                            const int *done, unsigned k)
{
    if (iter == 0) iter = done[1];  // replayed launches: the iteration number comes from the device (as xs_finalize)
    const double  alpha = hist[(size_t)iter * CGH_HIST + 2];
    const int64_t e0 = (int64_t)k * 4 * XS_GROUP * XS_CELL, e1 = e0 + 4 * XS_GROUP * XS_CELL < n ? e0 + 4 * XS_GROUP * XS_CELL : n;
    const int64_t j1 = e1 >> 1;
    for (int64_t j = (e0 >> 1) + threadIdx.x; j < j1; j += 4 * 256) {
        xd2_t xv[4], pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (j + q * 256 < j1) {
                xv[q] = __builtin_nontemporal_load(reinterpret_cast<const xd2_t *>(x) + j + q * 256);
                pv[q] = reinterpret_cast<const xd2_t *>(p)[j + q * 256];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (j + q * 256 < j1) {
                xd2_t o;
                o.x = madd_rn(xv[q].x, alpha, pv[q].x);
                o.y = madd_rn(xv[q].y, alpha, pv[q].y);
                __builtin_nontemporal_store(o, reinterpret_cast<xd2_t *>(x) + j + q * 256);
            }
    }
    if ((n & 1) && e1 == n && e0 < n && threadIdx.x == 0) x[n - 1] = madd_rn(x[n - 1], alpha, p[n - 1]);
}

// ---- Few cells: the vector update that consumes the sums' verdict stays in the launch.
// At these sizes (<= 2048 cells) a strict CG iteration is five short launches, and the two vector updates between the sums
// (r -= alpha Ap behind p.Ap; x += alpha p, p = r + beta p behind the sums of r) are ~4.7 us each although they move
// almost nothing.  Every summarising workgroup is resident anyway, so it stays on after its 16 cells: it loads its rows
// of the operands, waits for the driver's verdict -- alpha, or (beta, converged), as tagged words, two 64-bit words per
// double as in halo_ll.h -- and updates its 2048 rows; behind p.Ap it also leaves the approximate per-cell sums of the
// new r that the next launch predicts its binades from (as cg_update_r does).  Same arithmetic as cg_update_r /
// cg_update_xp (madd_rn), so the same bits.  Nothing waits for these workgroups; their wait gives up after 2 s (NaN).
__device__ __forceinline__ void xs_tail_publish(unsigned long long *box, unsigned tag, double v0, unsigned flag)
{
    const unsigned long long b0 = (unsigned long long)__double_as_longlong(v0), t = (unsigned long long)tag << 32;
    __hip_atomic_store(box + 0, t | (b0 & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(box + 1, t | (b0 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(box + 2, t | flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every lane of the calling wave gets (v0, flag); false after the timeout
__device__ __forceinline__ bool xs_tail_await(const unsigned long long *box, unsigned tag, double &v0, unsigned &flag)
{
    unsigned long long w0 = 0, w1 = 0, w2 = 0;
    const long long    t0 = wall_clock64();
    for (unsigned spin = 0;; ++spin) {
        w0 = __hip_atomic_load(box + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w1 = __hip_atomic_load(box + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w2 = __hip_atomic_load(box + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag && (unsigned)(w2 >> 32) == tag) break;
        if ((spin & 255u) == 255u && wall_clock64() - t0 > XR_PATIENCE) {
            v0 = __builtin_nan("");
            flag = 0;
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    v0 = __longlong_as_double((long long)((w1 << 32) | (w0 & 0xffffffffull)));
    flag = (unsigned)w2;
    return true;
}

// the rows of summarising workgroup `sb` of a launch of few cells: 16 cells, four per wave, one row pair per lane and cell
template <int KIND>
__device__ __forceinline__ void xs_tail(unsigned sb, int64_t n, const double *__restrict__ a, const double *__restrict__ b, int64_t ncells,
                                        double *__restrict__ cellsum, const unsigned long long *box, unsigned tag, double *upd,
                                        double *upd_p, const double *hist, int iter)
{
    const int     lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c0 = (int64_t)(sb >> 2) * XS_GROUP + (int64_t)(sb & 3) * 16 + wave * 4;
    double2       u[4], v[4], w[4];
    bool          whole[4], any[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = (c0 + q) * XS_CELL + 2 * lane;
        whole[q] = i + 1 < n;
        any[q] = i < n;
        u[q] = v[q] = w[q] = make_double2(0.0, 0.0);
        if constexpr (KIND == XS_DOT) {  // behind p.Ap (a = p, b = Ap): r -= alpha Ap
            if (whole[q]) {
                u[q] = *reinterpret_cast<const double2 *>(upd + i);
                v[q] = *reinterpret_cast<const double2 *>(b + i);
            } else if (any[q]) {
                u[q].x = upd[i];
                v[q].x = b[i];
            }
        } else {  // behind the sums of r (a = r): x += alpha p; p = r + beta p
            if (whole[q]) {
                u[q] = *reinterpret_cast<const double2 *>(upd + i);
                v[q] = *reinterpret_cast<const double2 *>(upd_p + i);
                w[q] = *reinterpret_cast<const double2 *>(a + i);
            } else if (any[q]) {
                u[q].x = upd[i];
                v[q].x = upd_p[i];
                w[q].x = a[i];
            }
        }
    }
    double   verdict;
    unsigned flag;
    xs_tail_await(box, tag, verdict, flag);
    if constexpr (KIND == XS_DOT) {
        const double nalpha = -verdict;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t i = (c0 + q) * XS_CELL + 2 * lane;
            double2       rn = make_double2(0.0, 0.0);
            if (any[q]) rn.x = madd_rn(u[q].x, nalpha, v[q].x);  // reference vec_axpy with -alpha (cg.c:469), as cg_update_r
            if (whole[q]) {
                rn.y = madd_rn(u[q].y, nalpha, v[q].y);
                *reinterpret_cast<double2 *>(upd + i) = rn;
            } else if (any[q]) {
                upd[i] = rn.x;
            }
            const double qs = wave_sum_lane63(rn.x * rn.x + rn.y * rn.y), as = wave_sum_lane63(fabs(rn.x) + fabs(rn.y));
            if (lane == 63 && c0 + q < ncells) {
                cellsum[c0 + q] = qs;
                cellsum[ncells + c0 + q] = as;
            }
        }
    } else {
        const double alpha = hist[(size_t)iter * CGH_HIST + 2], beta = verdict;
        const bool   conv = flag != 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t i = (c0 + q) * XS_CELL + 2 * lane;
            // reference vec_axpy (vec.c:88-94): x += alpha p (cg.c:467), then -- unless converged -- p = r + beta p (cg.c:489)
            if (whole[q]) {
                *reinterpret_cast<double2 *>(upd + i) = make_double2(madd_rn(u[q].x, alpha, v[q].x), madd_rn(u[q].y, alpha, v[q].y));
                if (!conv) *reinterpret_cast<double2 *>(upd_p + i) = make_double2(madd_rn(w[q].x, beta, v[q].x), madd_rn(w[q].y, beta, v[q].y));
            } else if (any[q]) {
                upd[i] = madd_rn(u[q].x, alpha, v[q].x);
                if (!conv) upd_p[i] = madd_rn(w[q].x, beta, v[q].x);
            }
        }
    }
}

template <int KIND, bool FLAT>
__global__ __launch_bounds__(256) void xs_fused(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                               const double *__restrict__ c, int64_t ncells, int64_t ngroups,
                                               const double *__restrict__ cellsum, const double *__restrict__ groupsum,
                                               const double *__restrict__ gprefix, xs_summary *__restrict__ cells,
                                               xs_summary *__restrict__ groups, xs_summary *__restrict__ batches,
                                               unsigned *__restrict__ ready, int mode, double *hist, int iter, double tol, int *done,
                                               double *out, double *xfer, long long *trace, double *xu_x, const double *xu_p,
                                               long long *wgtrace, unsigned *__restrict__ head_count, unsigned long long *tail_box,
                                               unsigned tail_tag)
{
    constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1, ND = S1 - S0 + 1;
    typedef xsm_shared<1> sum_shared;
    // optional stamps of every workgroup (tools/trace_wgs.py): [4 b] start, [4 b + 1] end, 100 MHz ticks; [4 b + 2] its role
    if (wgtrace != nullptr && threadIdx.x == 0) wgtrace[4 * (size_t)blockIdx.x] = wall_clock64();
    struct stamp_end {
        long long *p;
        __device__ ~stamp_end()
        {
            if (p != nullptr && threadIdx.x == 0) p[1] = wall_clock64();
        }
    } stamp{wgtrace != nullptr ? wgtrace + 4 * (size_t)blockIdx.x : nullptr};
    constexpr size_t       LDS = sizeof(sum_shared) > sizeof(xr_shared) ? sizeof(sum_shared) : sizeof(xr_shared);
    __shared__ __attribute__((aligned(16))) unsigned char raw[LDS];
    // Drivers and summarising workgroups of one launch all take the same side: the flag only changes when a driver
    // finalises, after the last summary of the launch (the workgroups that only carry x += alpha p: see below).
    const int frozen_at = done != nullptr && mode != 3 && mode != 0 ? *done : 0;
    const int64_t nbatch = FLAT ? ngroups : (ngroups + 63) / 64;
    if (blockIdx.x >= (unsigned)ND) {
        unsigned sb = blockIdx.x - ND;  // the workgroup's number among those that are no drivers
        if constexpr (KIND == XS_BOTH_SAME && !FLAT) {
            // The summaries of r.r and |r|_1 are bound by their arithmetic, so the launch also carries the x += alpha p of the
            // same CG iteration (cg.c:467 of the reference), which is pure bandwidth: every fifth workgroup updates the x of
            // four groups' rows instead of summarising (cg_update_xp then leaves x alone).  Tried in round 3 and dropped: one
            // x workgroup per group (129 us for the launch instead of 104: twice the workgroup slots), and the x update at
            // the tail of every summarising workgroup (109 us).
            if (xu_x != nullptr) {
                const unsigned k = sb / 5, r = sb % 5;
                if (r == 4) {
                    if (wgtrace != nullptr && threadIdx.x == 0) wgtrace[4 * (size_t)blockIdx.x + 2] = 2;  // x update
                    // nobody waits for these workgroups: one that starts after the driver has finalised a CONVERGING
                    // iteration finds the flag naming this very iteration and still owes its rows their update
                    const int it = iter ? iter : done[1];  // (replayed launches: the iteration number comes from the device)
                    if (frozen_at == 0 || frozen_at == it) xs_update_x(n, xu_x, xu_p, hist, it, done, k);
                    return;
                }
                sb = k * 4 + r;
            }
        }
        if (frozen_at != 0) return;
        if constexpr (FLAT) {
            xs_summarize_body<KIND, FLAT, true>(*reinterpret_cast<sum_shared *>(raw), sb, n, a, b, c, ncells, ngroups, cellsum, groupsum,
                                                 gprefix, cells, groups, nullptr, ready, batches, nbatch);
            if constexpr (KIND != XS_L1) {
                if (tail_box != nullptr)
                    xs_tail<KIND>(sb, n, a, b, ncells, const_cast<double *>(cellsum), tail_box, tail_tag, xu_x, const_cast<double *>(xu_p), hist, iter);
            }
        } else {
            // group level: summarising workgroups 0..3 share group 0 (one pass each), workgroup s >= 4 takes group s - 3
            if ((int64_t)sb >= ngroups + 3) return;
            xs_summarize_body<KIND, FLAT, true>(*reinterpret_cast<sum_shared *>(raw), sb < 4 ? 0u : sb - 3u, n, a, b, c, ncells, ngroups, cellsum,
                                                 groupsum, gprefix, cells, groups, nullptr, ready, batches, nbatch, sb < 4 ? (int)sb : -1,
                                                 head_count);
        }
        return;
    }
    if (frozen_at != 0) return;
    const int  slot = S0 + (int)blockIdx.x;
    xr_shared &S = *reinterpret_cast<xr_shared *>(raw);
    if (threadIdx.x < XR_CHAINS) S.chain_cell[threadIdx.x] = -1;
    if (threadIdx.x == 0) S.chain_n = S.walked = S.abort = 0;
    __syncthreads();
    if (threadIdx.x >= 64) {  // the walk is one wave; with few cells the others fetch ahead of it
        if constexpr (FLAT)
            xr_fetch_ahead<KIND>(S, slot, n, a, b, c, ncells, (int)nbatch, cells, ready + (slot - S0) * nbatch);
        return;
    }
    const double s = xr_sum<KIND, FLAT>(S, slot, n, a, b, c, ncells, ngroups, cells, groups,
                                  batches + (slot - S0) * nbatch, ready + (slot - S0) * nbatch, slot == S0 ? trace : nullptr);
    if (threadIdx.x != 0) return;
    double s_dot = s, s_l1 = s;
    if constexpr (S1 > S0) {
        // two drivers: the second to arrive has both sums (as xs_drive)
        unsigned long long *tk = reinterpret_cast<unsigned long long *>(xfer + 2);
        __hip_atomic_store(xfer + slot, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long old = __hip_atomic_fetch_add(tk, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) return;
        __hip_atomic_store(tk, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_dot = __hip_atomic_load(xfer + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_l1 = __hip_atomic_load(xfer + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    xs_finalize(KIND == XS_L1 ? 0.0 : s_dot, KIND == XS_DOT ? 0.0 : s_l1, mode, hist, iter, tol, done, out);
    if constexpr (FLAT && KIND != XS_L1) {
        if (tail_box != nullptr) {
            // the verdict for the workgroups that stayed on (xs_tail): alpha behind p.Ap; beta and "converged" behind the sums of r
            // (this thread wrote them into the history a moment ago); cg_update_xp's bookkeeping of the last finished iteration
            if (mode == 1) {
                xs_tail_publish(tail_box, tail_tag, hist[(size_t)iter * CGH_HIST + 2], 0u);
            } else {
                const bool conv = hist[(size_t)iter * CGH_HIST + 0] < tol;
                done[2] = iter;
                xs_tail_publish(tail_box, tail_tag, hist[(size_t)iter * CGH_HIST + 3], conv ? 1u : 0u);
            }
        }
    }
}

// The sums of a vector spread over several ranks, once the last rank's part is in (cgh_xsum_finalize): one thread.
__global__ void xs_finalize_kernel(const double *__restrict__ sums, int mode, double *hist, int iter, double tol, int *done)
{
    const bool frozen = done != nullptr && mode != 3 && mode != 0 && *done != 0;
    if (!frozen) xs_finalize(sums[0], sums[1], mode, hist, iter, tol, done, nullptr);
}


// ---------------------------------------------------------------------------------------------- sums over several ranks, composed
// The bit-identical multi-GPU mode (dist.hip): the reference's left-to-right sum over a vector whose rows are spread over N
// ranks, rank 0's first.  Round 2 chained the ranks (rank q summarised and walked its part only once rank q-1's exact sum
// had arrived: N rounds per sum).  Here every rank summarises its part AT ONCE, against an approximate prefix of the
// ranks before it (xc_offsets: the ranks' tree totals, exchanged through the mailboxes), joins its groups into at most 32
// batch summaries (xc_reduce_batches) and hands those 40-byte maps to every other rank (xc_walk).  Every rank then walks
// ALL ranks' batch summaries in rank order -- integer maps applied to the one exact running sum, a few instructions each
// -- and holds the same total bits as every other rank at the end.  A batch whose summary does not apply to the sum that
// really arrives (the head of the sum, a crossing and a tie in one batch, a misprediction) is walked by its OWNER, who
// has the groups, cells and elements, and who sends the exact sum behind it to the others, which wait for it there.
// Messages are tagged 64-bit words in mailbox region 3 (halo_ll.h); the slot comes from the count of exchanges executed.
__device__ __forceinline__ uint64_t *xc_slot_of(uint64_t *box, unsigned seq, int sender)
{
    return box + CGH_XC_REGION0 + ((size_t)(seq & (CGH_LINK_SLOTS - 1)) * CGH_LINK_MAX_RANKS + (size_t)sender) * CGH_XC_WORDS;
}

// offsets[slot] = sum over the ranks before this one of their approximate totals (any order: a prediction)
__global__ __launch_bounds__(256) void xc_offsets(cgh_link L, int64_t ngroups, const double *__restrict__ groupsum, int s0, int s1,
                                                  double *__restrict__ offsets, const int *__restrict__ done, int check_done,
                                                  unsigned epoch, unsigned idx)
{
    __shared__ double lds[4];
    __shared__ double ex[2 * CGH_LINK_MAX_RANKS];
    if (check_done && *done != 0) return;
    const unsigned seq = (unsigned)*L.rounds;
    const unsigned tag = cgh_ll_tag(epoch, idx);
    double         mine[2] = {0.0, 0.0};
    for (int slot = s0; slot <= s1; ++slot) mine[slot] = block_sum_partials(groupsum + slot * ngroups, (int)ngroups, 1, lds);
    const int t = threadIdx.x;
    if (t < 2 * L.nranks) {
        const int peer = t >> 1, w = t & 1;
        if (peer > L.me) {  // the ranks behind me want my totals
            uint64_t *q = xc_slot_of(L.peers[peer], seq, L.me) + 2 * w;
            ll_store(q, q + 1, mine[w], tag);
        }
        ex[t] = 0.0;
        if (peer < L.me) {
            const uint64_t *q = xc_slot_of(const_cast<uint64_t *>(L.box), seq, peer) + 2 * w;
            ex[t] = ll_wait(q, q + 1, tag, L.err);
        }
    }
    __syncthreads();
    if (t < 2) {
        double v = 0.0;
        for (int r = 0; r < L.me; ++r) v += ex[2 * r + t];
        offsets[t] = v;
    }
    if (t == 0) *L.rounds = (unsigned long long)seq + 1ull;
}

// batches[slot][bb] = join of the 64 items (groups; cells when there is no group level) of batch bb: one wave per (slot, batch)
__global__ __launch_bounds__(256) void xc_reduce_batches(const xs_summary *__restrict__ items, int64_t nitems, int nb, int s0, int s1,
                                                         xs_summary *__restrict__ batches, const int *__restrict__ done, int check_done)
{
    if (check_done && *done != 0) return;
    const int lane = threadIdx.x & 63;
    const int wv = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int slot = s0 + wv / nb, bb = wv % nb;
    if (slot > s1) return;
    const int64_t i = (int64_t)bb * 64 + lane;
    xs_summary    q = xs_identity();
    if (i < nitems) q = items[slot * nitems + i];
    const xs_summary j = wave_join(q);
    if (lane == 0) batches[slot * CGH_XC_MAX_BATCHES + bb] = j;
}

// s + (the elements of cell `cell` of this rank), left to right (one wave; `stage` = 128 doubles of LDS of its own)
template <int KIND>
__device__ __forceinline__ double xc_chain(double s, double *stage, int slot, int64_t n, const double *__restrict__ a,
                                           const double *__restrict__ b, const double *__restrict__ c, int64_t cell)
{
    const int lane = threadIdx.x & 63;
    double    t0, t1;
    xs_cell_terms<KIND>(slot, n, a, b, c, cell, t0, t1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the previous cell's reads are over
    *reinterpret_cast<double2 *>(stage + 2 * lane) = make_double2(t0, t1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return xs_chain(s, stage);
}

// the owner's walk of ONE of its batches from the exact running sum s: groups, then the cells of a group that fails, then
// the elements of a cell that fails (as xd_sum, without its staging pools: only the batches the chain could not apply come here)
template <int KIND>
__device__ double xc_walk_own_batch(double s, double *stage, int slot, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                    const double *__restrict__ c, int64_t ncells, int64_t ngroups, const xs_summary *__restrict__ cells,
                                    const xs_summary *__restrict__ groups, int bb)
{
    const int lane = threadIdx.x & 63;
    cells += slot * ncells;
    const xs_summary *items = groups ? groups + slot * ngroups : cells;
    const int64_t     nitems = groups ? ngroups : ncells, i0 = (int64_t)bb * 64;
    const int         cnt = (int)(nitems - i0 < 64 ? nitems - i0 : 64);
    xs_summary        Gq = xs_identity();
    if (lane < cnt) Gq = items[i0 + lane];
    const int64_t Ginc = xs_scan(Gq, cnt);
    int           gstart = 0;
    while (gstart < cnt) {
        const int f = xs_batch(Gq, Ginc, gstart, cnt, s);
        if (f >= cnt) break;
        const int64_t it = i0 + f;
        if (!groups) {
            s = xc_chain<KIND>(s, stage, slot, n, a, b, c, it);
        } else {
            const int64_t c0 = it * XS_GROUP;
            const int     ccnt = (int)(ncells - c0 < XS_GROUP ? ncells - c0 : XS_GROUP);
            xs_summary    Cq = xs_identity();
            if (lane < ccnt) Cq = cells[c0 + lane];
            const int64_t Cinc = xs_scan(Cq, ccnt);
            int           cstart = 0;
            while (cstart < ccnt) {
                const int cf = xs_batch(Cq, Cinc, cstart, ccnt, s);
                if (cf >= ccnt) break;
                s = xc_chain<KIND>(s, stage, slot, n, a, b, c, c0 + cf);
                cstart = cf + 1;
            }
        }
        gstart = f + 1;
    }
    return s;
}

// One workgroup per rank: all four waves send this rank's batch summaries to every other rank; then wave w (w < number of
// sums) walks sum S0 + w over the batch summaries of ALL ranks in rank order and finalises like seq_sums.
template <int KIND>
__global__ __launch_bounds__(256) void xc_walk(cgh_link L, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                               const double *__restrict__ c, int64_t ncells, int64_t ngroups,
                                               const xs_summary *__restrict__ cells, const xs_summary *__restrict__ groups,
                                               const xs_summary *__restrict__ batches, int nb, int mode, double *hist, int iter, double tol,
                                               int *done, int check_done, unsigned epoch, unsigned idx)
{
    constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1;
    __shared__ __attribute__((aligned(16))) double stage[2][XS_CELL];
    __shared__ double                              total[2];
    if (check_done && *done != 0) return;
    const unsigned seq = (unsigned)*L.rounds;
    const unsigned tag = cgh_ll_tag(epoch, idx);
    const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // (everybody has read the count before thread 0 moves it on at the end)
    // 1. my batch summaries (and their number) to every other rank
    for (int peer = 0; peer < L.nranks; ++peer) {
        if (peer == L.me) continue;
        uint64_t *dst = xc_slot_of(L.peers[peer], seq, L.me);
        if (threadIdx.x == 0) ll_store32(dst + 4, (unsigned)nb, tag);
        for (int k = threadIdx.x; k < (S1 - S0 + 1) * nb * 10; k += 256) {
            const int slot = S0 + k / (nb * 10), bb = (k / 10) % nb, w = k % 10;
            const xs_summary &q = batches[slot * CGH_XC_MAX_BATCHES + bb];
            const uint64_t    words[5] = {(uint64_t)q.A, (uint64_t)q.C, (uint64_t)q.Mlo, (uint64_t)q.Mhi,
                                          (uint64_t)(uint32_t)q.E | ((uint64_t)(uint32_t)q.meta << 32)};
            const uint64_t    v = words[w >> 1];
            ll_store32(dst + CGH_XC_SUMMARY0 + (slot * CGH_XC_MAX_BATCHES + bb) * 10 + w, (unsigned)((w & 1) ? v >> 32 : v), tag);
        }
    }
    // 2. the walk
    if (wave <= S1 - S0) {
        const int slot = S0 + wave;
        double    s = 0.0;
        bool      failed = false;
        for (int q = 0; q < L.nranks && !failed; ++q) {
            const uint64_t *src = xc_slot_of(const_cast<uint64_t *>(L.box), seq, q);
            int             nbq = nb;
            xs_summary      Bq = xs_identity();
            if (q == L.me) {
                if (lane < nb) Bq = batches[slot * CGH_XC_MAX_BATCHES + lane];
            } else {
                nbq = (int)ll_wait32(src + 4, tag, L.err, &failed);
                nbq = nbq < 0 ? 0 : (nbq > CGH_XC_MAX_BATCHES ? CGH_XC_MAX_BATCHES : nbq);
                if (lane < nbq) {
                    uint64_t        words[5];
                    const uint64_t *at = src + CGH_XC_SUMMARY0 + (slot * CGH_XC_MAX_BATCHES + lane) * 10;
#pragma unroll
                    for (int w = 0; w < 5; ++w) {
                        const uint64_t lo = ll_wait32(at + 2 * w, tag, L.err, &failed), hi = ll_wait32(at + 2 * w + 1, tag, L.err, &failed);
                        words[w] = (hi << 32) | lo;
                    }
                    Bq.A = (int64_t)words[0];
                    Bq.C = (int64_t)words[1];
                    Bq.Mlo = (int64_t)words[2];
                    Bq.Mhi = (int64_t)words[3];
                    Bq.E = (int32_t)(uint32_t)words[4];
                    Bq.meta = (int32_t)(uint32_t)(words[4] >> 32);
                }
            }
            failed = __ballot(failed) != 0ull;
            if (failed) break;
            const int64_t Binc = xs_scan(Bq, nbq);
            int           start = 0;
            while (start < nbq) {
                const int fb = xs_batch(Bq, Binc, start, nbq, s);  // the longest prefix of batches that applies to s, applied
                if (fb >= nbq) break;
                // batch fb of rank q does not apply to the sum that arrives: its owner walks it and tells the others
                if (q == L.me) {
                    s = xc_walk_own_batch<KIND>(s, stage[wave], slot, n, a, b, c, ncells, ngroups, cells, groups, fb);
                    if (lane < L.nranks && lane != L.me) {
                        uint64_t *dst = xc_slot_of(L.peers[lane], seq, L.me) + CGH_XC_RESULT0 + (slot * CGH_XC_MAX_BATCHES + fb) * 2;
                        ll_store(dst, dst + 1, s, tag);
                    }
                } else {
                    const uint64_t *at = src + CGH_XC_RESULT0 + (slot * CGH_XC_MAX_BATCHES + fb) * 2;
                    s = ll_wait(at, at + 1, tag, L.err);
                    if (__hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) failed = true;
                }
                start = fb + 1;
            }
        }
        if (lane == 0) total[slot] = failed ? __builtin_nan("") : s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        xs_finalize(KIND == XS_L1 ? 0.0 : total[0], KIND == XS_DOT ? 0.0 : total[1], mode, hist, iter, tol, done, nullptr);
        *L.rounds = (unsigned long long)seq + 1ull;
    }
}

long long *g_xs_wgtrace = nullptr;  // device buffer of 4 words per workgroup of a fused launch, or NULL (cgh_xsum_set_wgtrace)
long long *g_xs_prof = nullptr;  // device buffer of 16 words, or NULL (cgh_xsum_set_prof)
long long *g_xs_trace = nullptr; // device buffer of 224 words, or NULL (cgh_xsum_set_trace): time stamps of the fused driver

template <int KIND>
int launch_all(cgh_xsum *w, bool presummed, int64_t n, const double *a, const double *b, const double *c, int mode, double *hist, int iter,
               double tol, int *done, double *out, hipStream_t s, const double *s_in = nullptr, double *s_out = nullptr,
               double *xu_x = nullptr, const double *xu_p = nullptr, int *xu_taken = nullptr)
{
    if (xu_taken) *xu_taken = 0;
    const unsigned ng = (unsigned)w->ngroups;
    const double *gsum = w->groupsum;
    struct consume {
        cgh_xsum *w;
        ~consume() { w->groups_presummed = false; }
    } consume_flag{w};
    if (presummed && w->ncells <= 4096) {
        gsum = nullptr;  // every workgroup of xs_summarize adds up the cell sums before its group itself: no launch here
    } else if (presummed && w->groups_presummed) {
        // (cg_update_r left the group sums beside the cell sums: nothing to launch)
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
    const bool  flat = w->ncells <= XD_ITEMS;  // few cells: no group level
    xs_summary *grp = flat ? nullptr : w->groups;
    if (s_in == nullptr && s_out == nullptr && g_xs_prof == nullptr && g_cfg.xsum_fused) {
        constexpr int ND = KIND == XS_BOTH_SAME ? 2 : 1;
        // x += alpha p rides along with the sums of r (mode 2, both sums, group level): see xs_fused
        const bool xu = KIND == XS_BOTH_SAME && !flat && mode == 2 && xu_x != nullptr && xu_p != nullptr && g_cfg.xsum_xupd;
        if (xu_taken) *xu_taken = xu ? 1 : 0;
        // few cells: the vector update that consumes the verdict stays in the launch (xs_tail) -- r -= alpha Ap behind p.Ap
        // (mode 1, xu_x = r), x += alpha p and p = r + beta p behind the sums of r (mode 2, xu_x = x, xu_p = p)
        bool tail = flat && g_cfg.xsum_tail && w->tail != nullptr && iter > 0 && hist != nullptr && done != nullptr && xu_x != nullptr &&
                    ((KIND == XS_DOT && mode == 1) || (KIND == XS_BOTH_SAME && mode == 2 && xu_p != nullptr));
        if (tail) {
            // (not under stream capture: a replayed launch carries the same tag as the replay before it, whose verdict is
            // still in the box -- right only as long as nothing about the solve changed in between)
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(s, &cap);
            tail = cap == hipStreamCaptureStatusNone;
        }
        if (tail && xu_taken) *xu_taken = 2;
        const unsigned tag = tail ? 0x80000000u | (++w->tail_seq & 0x7fffffffu) : 0u;
        if (flat)
            hipLaunchKernelGGL((xs_fused<KIND, true>), dim3(4 * ng + ND), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups,
                               (const double *)w->cellsum, gsum, prefix, w->cells, w->groups, w->batches, w->ready, mode, hist, iter, tol, done,
                               out, w->xfer, g_xs_trace, tail ? xu_x : nullptr, tail ? xu_p : nullptr, g_xs_wgtrace, w->head,
                               tail ? w->tail : nullptr, tag);
        else
            hipLaunchKernelGGL((xs_fused<KIND, false>), dim3(xu ? 5 * ((ng + 3 + 3) / 4) + ND : ng + 3 + ND), dim3(256), 0, s, n, a, b, c, w->ncells,
                               w->ngroups, (const double *)w->cellsum, gsum, prefix, w->cells, w->groups, w->batches, w->ready, mode, hist, iter,
                               tol, done, out, w->xfer, g_xs_trace, xu ? xu_x : nullptr, xu ? xu_p : nullptr, g_xs_wgtrace, w->head,
                               (unsigned long long *)nullptr, 0u);
        CGH_HIP(hipGetLastError());
        return CGH_OK;
    }
    if (flat)
        hipLaunchKernelGGL((xs_summarize<KIND, true>), dim3(4 * ng), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups,
                           (const double *)w->cellsum, gsum, prefix, w->cells, grp, s_in);
    else
        hipLaunchKernelGGL((xs_summarize<KIND, false>), dim3(ng), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups,
                           (const double *)w->cellsum, gsum, prefix, w->cells, grp, s_in);
    const int64_t nbatch = ((flat ? w->ncells : w->ngroups) + 63) / 64;
    const unsigned dthreads = nbatch >= XD_WAVES ? XD_THREADS : 64u * (unsigned)(nbatch < 1 ? 1 : nbatch);
    hipLaunchKernelGGL((xs_drive<KIND>), dim3(KIND == XS_BOTH_SAME ? 2 : 1), dim3(dthreads), 0, s, n, a, b, c, w->ncells,
                       w->ngroups, (const xs_summary *)w->cells, (const xs_summary *)grp, mode, hist, iter, tol, done, out, w->xfer,
                       s_in, s_out, g_xs_prof);
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
    w->xfer = (double *)cgh_alloc(sizeof(double) * 4);
    if (w->xfer && hipMemset(w->xfer, 0, sizeof(double) * 4) != hipSuccess) {
        cgh_free(w->xfer);
        w->xfer = nullptr;
    }
    const size_t nready = 2 * (size_t)((w->ngroups + 63) / 64 > 32 ? (w->ngroups + 63) / 64 : 32);  // (few cells: one batch per group, <= 32)
    w->ready = (unsigned *)cgh_alloc(sizeof(unsigned) * (nready + 8));
    w->head = w->ready ? w->ready + nready : nullptr;
    if (w->ready && hipMemset(w->ready, 0, sizeof(unsigned) * (nready + 8)) != hipSuccess) {
        cgh_free(w->ready);
        w->ready = nullptr;
    }
    w->batches = (xs_summary *)cgh_alloc(sizeof(xs_summary) * nready);
    w->tail = (unsigned long long *)cgh_alloc(sizeof(unsigned long long) * 8);
    if (w->tail && hipMemset(w->tail, 0, sizeof(unsigned long long) * 8) != hipSuccess) {
        cgh_free(w->tail);
        w->tail = nullptr;
    }
    if (!w->cellsum || !w->groupsum || !w->gprefix || !w->cells || !w->groups || !w->xfer || !w->ready || !w->batches || !w->tail) {
        cgh_xsum_free(w);
        return nullptr;
    }
    return w;
}

// The same workspace laid out in memory the caller already holds -- nothing is allocated, on the host or on the device
// (cgh_cg_solve_noalloc: its built-in __device__ array).  obj: host storage for the descriptor (cgh_xsum_object_bytes());
// dev: 256-byte aligned device memory of at least cgh_xsum_device_bytes(n).  The counters the fused launch expects to find
// at zero are cleared on stream s (a previous call may have been cut short).  cgh_xsum_free leaves all of it alone.
static size_t xs_up256(size_t b) { return (b + 255) & ~(size_t)255; }
size_t cgh_xsum_object_bytes(void) { return sizeof(cgh_xsum); }
size_t cgh_xsum_device_bytes(int64_t n)
{
    const size_t ncells = (size_t)((n + XS_CELL - 1) / XS_CELL), ngroups = (ncells + XS_GROUP - 1) / XS_GROUP;
    const size_t nready = 2 * ((ngroups + 63) / 64 > 32 ? (ngroups + 63) / 64 : 32);
    return xs_up256(sizeof(double) * 2 * ncells) + 2 * xs_up256(sizeof(double) * 2 * ngroups) + xs_up256(sizeof(xs_summary) * 2 * ncells) +
           xs_up256(sizeof(xs_summary) * 2 * ngroups) + xs_up256(sizeof(double) * 4) + xs_up256(sizeof(unsigned) * (nready + 8)) +
           xs_up256(sizeof(xs_summary) * nready) + xs_up256(sizeof(unsigned long long) * 8);
}
cgh_xsum *cgh_xsum_place(void *obj, size_t obj_bytes, int64_t n, void *dev, size_t dev_bytes, hipStream_t s)
{
    if (n <= 0 || !obj || !dev || obj_bytes < sizeof(cgh_xsum) || dev_bytes < cgh_xsum_device_bytes(n) || (reinterpret_cast<uintptr_t>(dev) & 255u))
        return nullptr;
    cgh_xsum *w = new (obj) cgh_xsum();
    w->owns = false;
    w->n = n;
    w->ncells = (n + XS_CELL - 1) / XS_CELL;
    w->ngroups = (w->ncells + XS_GROUP - 1) / XS_GROUP;
    const size_t nready = 2 * (size_t)((w->ngroups + 63) / 64 > 32 ? (w->ngroups + 63) / 64 : 32);
    char *at = (char *)dev;
    auto  take = [&at](size_t bytes) { char *p = at; at += xs_up256(bytes); return (void *)p; };
    w->cellsum = (double *)take(sizeof(double) * 2 * (size_t)w->ncells);
    w->groupsum = (double *)take(sizeof(double) * 2 * (size_t)w->ngroups);
    w->gprefix = (double *)take(sizeof(double) * 2 * (size_t)w->ngroups);
    w->cells = (xs_summary *)take(sizeof(xs_summary) * 2 * (size_t)w->ncells);
    w->groups = (xs_summary *)take(sizeof(xs_summary) * 2 * (size_t)w->ngroups);
    w->xfer = (double *)take(sizeof(double) * 4);
    w->ready = (unsigned *)take(sizeof(unsigned) * (nready + 8));
    w->head = w->ready + nready;
    w->batches = (xs_summary *)take(sizeof(xs_summary) * nready);
    w->tail = (unsigned long long *)take(sizeof(unsigned long long) * 8);
    if (hipMemsetAsync(w->xfer, 0, sizeof(double) * 4, s) != hipSuccess || hipMemsetAsync(w->ready, 0, sizeof(unsigned) * (nready + 8), s) != hipSuccess ||
        hipMemsetAsync(w->tail, 0, sizeof(unsigned long long) * 8, s) != hipSuccess)
        return nullptr;
    return w;
}

void cgh_xsum_free(cgh_xsum *w)
{
    if (!w || !w->owns) return;
    cgh_free(w->cellsum);
    cgh_free(w->groupsum);
    cgh_free(w->gprefix);
    cgh_free(w->cells);
    cgh_free(w->groups);
    cgh_free(w->xfer);
    cgh_free(w->ready);
    cgh_free(w->batches);
    cgh_free(w->tail);
    delete w;
}

// Profile of the serial walks that follow (see XS_PROF_ADD): 16 zeroed 64-bit words of device memory, or NULL to stop.
extern "C" void cgh_xsum_set_trace(void *dev_224_words) { g_xs_trace = (long long *)dev_224_words; }
extern "C" void cgh_xsum_set_wgtrace(void *dev_4_words_per_workgroup) { g_xs_wgtrace = (long long *)dev_4_words_per_workgroup; }
extern "C" void cgh_xsum_set_prof(void *dev_16_words) { g_xs_prof = (long long *)dev_16_words; }

double *cgh_xsum_cellsums(cgh_xsum *w) { return w ? w->cellsum : nullptr; }
// where the producer of the next vector to be summed (both slots) may leave the sums of the 64-cell groups as well; a
// producer that did says so with cgh_xsum_groups_written right before the sums' launch
double *cgh_xsum_groupsums(cgh_xsum *w) { return w ? w->groupsum : nullptr; }
void    cgh_xsum_groups_written(cgh_xsum *w)
{
    if (w) w->groups_presummed = true;
}
int64_t cgh_xsum_cells(const cgh_xsum *w) { return w ? w->ncells : 0; }

// One rank's part of the sums of a vector spread over several ranks (rank order = element order): starts from the exact
// running sums s_in[0..1] the ranks before left (device; NULL = this is the first part), leaves its own in s_out[0..1].
// Nothing is finalised: cgh_xsum_finalize does that on every rank once the last part is in.
int cgh_xsum_launch_part(cgh_xsum *w, int64_t n, const double *a, const double *b, const double *c, const double *s_in,
                         double *s_out, const int *done, hipStream_t s, bool presummed)
{
    if (!w || n != w->n || !s_out) return CGH_EINVAL;
    int *dn = const_cast<int *>(done);
    if (a && b && !c) return launch_all<XS_DOT>(w, presummed, n, a, b, c, 1, nullptr, 0, 0.0, dn, nullptr, s, s_in, s_out);
    if (!a && !b && c) return launch_all<XS_L1>(w, presummed, n, a, b, c, 1, nullptr, 0, 0.0, dn, nullptr, s, s_in, s_out);
    if (a && a == b && a == c) return launch_all<XS_BOTH_SAME>(w, presummed, n, a, b, c, 1, nullptr, 0, 0.0, dn, nullptr, s, s_in, s_out);
    return CGH_EINVAL;
}


// One of the reference's sums over a vector spread over the ranks of L, composed (see xc_walk): collective, every rank
// launches the same sequence.  n / a / b / c / presummed as cgh_xsum_launch; offsets = 2 doubles of device scratch.
// Returns CGH_EINVAL if this rank's part has more batches than a message holds (the caller then chains the ranks).
int cgh_xsum_composed_batches(const cgh_xsum *w)
{
    if (!w) return 0;
    const int64_t items = w->ncells <= XD_ITEMS ? w->ncells : w->ngroups;
    return (int)((items + 63) / 64);
}

template <int KIND>
static int launch_composed(cgh_xsum *w, const cgh_link &L, bool presummed, int64_t n, const double *a, const double *b, const double *c,
                           int mode, double *hist, int iter, double tol, int *done, bool check_done, unsigned epoch, unsigned idx,
                           double *offsets, hipStream_t s)
{
    constexpr int S0 = KIND == XS_L1 ? 1 : 0, S1 = KIND == XS_DOT ? 0 : 1;
    const int      nb = cgh_xsum_composed_batches(w), cd = check_done ? 1 : 0;
    const unsigned ng = (unsigned)w->ngroups;
    if (nb < 1 || nb > CGH_XC_MAX_BATCHES) return CGH_EINVAL;
    // approximate sums of every cell and group (the SpMV / the r update may have left the cells')
    if (presummed) {
        const int64_t waves = (int64_t)(S1 - S0 + 1) * w->ngroups;
        hipLaunchKernelGGL(xs_group_sums, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, w->ncells, w->ngroups, (const double *)w->cellsum,
                           w->groupsum, S0, S1);
    } else {
        hipLaunchKernelGGL((xs_cell_sums<KIND>), dim3(ng), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups, w->cellsum, w->groupsum);
    }
    w->groups_presummed = false;
    hipLaunchKernelGGL(xc_offsets, dim3(1), dim3(256), 0, s, L, w->ngroups, (const double *)w->groupsum, S0, S1, offsets, (const int *)done, cd,
                       epoch, 2u * idx);
    const bool  flat = w->ncells <= XD_ITEMS;
    xs_summary *grp = flat ? nullptr : w->groups;
    if (flat)
        hipLaunchKernelGGL((xs_summarize<KIND, true>), dim3(4 * ng), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups,
                           (const double *)w->cellsum, (const double *)w->groupsum, (const double *)nullptr, w->cells, grp, (const double *)offsets);
    else
        hipLaunchKernelGGL((xs_summarize<KIND, false>), dim3(ng), dim3(256), 0, s, n, a, b, c, w->ncells, w->ngroups,
                           (const double *)w->cellsum, (const double *)w->groupsum, (const double *)nullptr, w->cells, grp, (const double *)offsets);
    hipLaunchKernelGGL(xc_reduce_batches, dim3((unsigned)(((S1 - S0 + 1) * nb + 3) / 4)), dim3(256), 0, s,
                       (const xs_summary *)(flat ? w->cells : w->groups), flat ? w->ncells : w->ngroups, nb, S0, S1, w->batches, (const int *)done, cd);
    hipLaunchKernelGGL((xc_walk<KIND>), dim3(1), dim3(256), 0, s, L, n, a, b, c, w->ncells, w->ngroups, (const xs_summary *)w->cells,
                       (const xs_summary *)grp, (const xs_summary *)w->batches, nb, mode, hist, iter, tol, done, cd, epoch, 2u * idx + 1u);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

int cgh_xsum_launch_composed(cgh_xsum *w, const cgh_link *L, int64_t n, const double *a, const double *b, const double *c, int mode,
                             double *hist, int iter, double tol, int *done, bool check_done, unsigned epoch, unsigned idx, double *offsets,
                             hipStream_t s, bool presummed)
{
    if (!w || !L || n != w->n || !offsets) return CGH_EINVAL;
    if (a && b && !c) return launch_composed<XS_DOT>(w, *L, presummed, n, a, b, c, mode, hist, iter, tol, done, check_done, epoch, idx, offsets, s);
    if (!a && !b && c) return launch_composed<XS_L1>(w, *L, presummed, n, a, b, c, mode, hist, iter, tol, done, check_done, epoch, idx, offsets, s);
    if (a && a == b && a == c)
        return launch_composed<XS_BOTH_SAME>(w, *L, presummed, n, a, b, c, mode, hist, iter, tol, done, check_done, epoch, idx, offsets, s);
    return CGH_EINVAL;
}

int cgh_xsum_finalize(const double *sums, int mode, double *hist, int iter, double tol, int *done, hipStream_t s)
{
    hipLaunchKernelGGL(xs_finalize_kernel, dim3(1), dim3(1), 0, s, sums, mode, hist, iter, tol, done);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

int cgh_xsum_launch(cgh_xsum *w, int64_t n, const double *a, const double *b, const double *c, int mode, double *hist,
                    int iter, double tol, int *done, double *out, hipStream_t s, bool presummed, double *xu_x, const double *xu_p,
                    int *xu_taken)
{
    if (xu_taken) *xu_taken = 0;
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
    if (a && b && !c) return launch_all<XS_DOT>(w, presummed, n, a, b, c, mode, hist, iter, tol, done, out, s, nullptr, nullptr, xu_x, xu_p, xu_taken);
    if (!a && !b && c) return launch_all<XS_L1>(w, presummed, n, a, b, c, mode, hist, iter, tol, done, out, s);
    if (a && a == b && a == c)
        return launch_all<XS_BOTH_SAME>(w, presummed, n, a, b, c, mode, hist, iter, tol, done, out, s, nullptr, nullptr, xu_x, xu_p, xu_taken);
    cgh_set_error("cgh_xsum_launch: unsupported operand combination");
    return CGH_EINVAL;
}
