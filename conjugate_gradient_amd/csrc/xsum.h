// xsum.h -- order-exact parallel summation: the arithmetic shared by the device kernels (xsum.hip) and the host
// model (xsum_model.cpp) that checks the algorithm on the CPU.
//
// Problem.  The reference sums left to right in double precision (vec_dot / vec_l1norm, src/vec.c:67-85 there):
//     s_0 = 0,  s_{k+1} = RN(s_k + t_k).
// RN is not associative, so a tree reduction gives different bits, and CG amplifies the difference.  A single lane
// reproduces the order but needs ~6.5 ns per element.  This scheme reproduces the SAME bits in parallel.
//
// Idea.  Write s = sigma * M * u with u = ulp(s) = 2^(E-1075) (E = biased exponent) and M an integer in
// [2^52, 2^53).  As long as s stays in that binade, adding t moves M by an integer:
//     RN(s + t) = sigma * u * (M + rint(tau)),   tau = sigma * t / u,
// provided tau is not an exact tie (|tau - rint(tau)| != 1/2) -- and integer addition IS associative.  So for a
// run of elements and an ASSUMED (sigma, E) one can precompute, in parallel,
//     D  = sum of rint(tau_k)              (how far M moves over the run)
//     lo = min over prefixes, hi = max over prefixes of the partial sums
// and the run is "regular" for a concrete incoming M iff  M+lo >= 2^52+1  and  M+hi <= 2^53-2  (M never leaves the
// binade; the margins make the pre-rounding value stay inside too) and no element was a tie, non-finite, or as
// large as the running sum (|tau| >= 2^52).  Regular runs compose by integer addition, so whole batches of them
// are validated and applied with one prefix sum.
//
// The binade to assume comes from a cheap approximate prefix sum (tree order): it predicts (sigma, E) of the running
// sum at the start of every 128-element cell.  Where the prediction is wrong, or the run is irregular (the sum
// crosses a power of two, a tie, a huge element), that one cell is added element by element.  The running sum
// crosses a binade only O(log) times, so almost everything goes through the integer path.
//
// Hierarchy: cell = 128 elements, group = 64 cells.  The driver walks group summaries 64 at a time, drops to the
// cells of the first group that fails, and to the elements of the first cell that fails.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define XS_HD __host__ __device__ inline
#else
#define XS_HD inline
#endif

constexpr int     XS_CELL = 128;            // elements per cell
constexpr int     XS_GROUP = 64;            // cells per group
constexpr int64_t XS_TWO52 = 1LL << 52;
constexpr int64_t XS_TWO53 = 1LL << 53;
constexpr int64_t XS_SAT = 1LL << 61;       // sums saturate here (any such run is invalid anyway)

enum { XS_IRREGULAR = 1 };                  // flags: tie / |tau| >= 2^52 / non-finite / no usable prediction / mixed binades

struct xs_summary {
    int64_t D;       // total movement of M
    int64_t lo, hi;  // min / max over ALL prefixes of the run, the empty one included (lo <= 0 <= hi)
    int32_t E;       // assumed biased exponent (1..2046); -1 = empty run (wildcard)
    int32_t meta;    // bit0: XS_IRREGULAR, bit1: assumed sign negative
};

XS_HD int64_t xs_sat_add(int64_t a, int64_t b)
{
    const int64_t r = a + b;  // |a|,|b| <= 2^61: no overflow
    return r > XS_SAT ? XS_SAT : (r < -XS_SAT ? -XS_SAT : r);
}

XS_HD uint64_t xs_bits(double v)
{
    uint64_t b;
    memcpy(&b, &v, sizeof b);
    return b;
}
XS_HD double xs_from_bits(uint64_t b)
{
    double v;
    memcpy(&v, &b, sizeof v);
    return v;
}

// (E, negative?) of a running sum usable as a binade assumption; false for 0, subnormal, inf, nan
XS_HD bool xs_binade_of(double s, int32_t *E, int32_t *neg)
{
    const uint64_t b = xs_bits(s);
    const int32_t  e = (int32_t)((b >> 52) & 0x7ff);
    *E = e;
    *neg = (int32_t)(b >> 63);
    return e >= 1 && e <= 2046;
}

XS_HD int64_t xs_mantissa(double s) { return (int64_t)((xs_bits(s) & ((1ULL << 52) - 1)) | (1ULL << 52)); }

XS_HD double xs_compose(int32_t E, int32_t neg, int64_t M)  // M in [2^52, 2^53)
{
    return xs_from_bits(((uint64_t)neg << 63) | ((uint64_t)E << 52) | ((uint64_t)M & ((1ULL << 52) - 1)));
}

// One element under the assumption (E, neg): d = rint(tau); *bad if the element cannot go through the integer path.
XS_HD int64_t xs_element(double t, int32_t E, int32_t neg, bool *bad)
{
    // tau = sigma * t * 2^(1075-E), exact (power-of-two scaling; overflow -> inf is caught by the range test,
    // underflow only affects values far below 1/2)
    const double tau = ldexp(neg ? -t : t, 1075 - E);
    if (!(fabs(tau) < 4503599627370496.0)) {  // 2^52; also catches NaN and inf
        *bad = true;
        return 0;
    }
    const double d = rint(tau);  // round to nearest, ties to even (default rounding mode)
    if (fabs(tau - d) == 0.5) {  // exact tie: the result depends on the parity of M -> element-by-element
        *bad = true;
        return 0;
    }
    return (int64_t)d;
}

// the empty run: joins with anything, applies to anything (E = -1 is the wildcard binade)
XS_HD xs_summary xs_identity()
{
    xs_summary q;
    q.D = q.lo = q.hi = 0;
    q.E = -1;
    q.meta = 0;
    return q;
}

// the run made of the single element t, under the assumption (E, neg)
XS_HD xs_summary xs_single(double t, int32_t E, int32_t neg)
{
    bool          bad = false;
    const int64_t d = xs_element(t, E, neg, &bad);
    xs_summary    q;
    q.D = d;
    q.lo = d < 0 ? d : 0;  // the empty prefix (0) is included: the incoming M must be inside the margins as well
    q.hi = d > 0 ? d : 0;
    q.E = E;
    q.meta = (neg ? 2 : 0) | (bad ? XS_IRREGULAR : 0);
    return q;
}

// summary of the concatenation A then B (same assumed binade, else irregular)
XS_HD xs_summary xs_join(const xs_summary &a, const xs_summary &b)
{
    if (a.E < 0 && !(a.meta & XS_IRREGULAR)) return b;
    if (b.E < 0 && !(b.meta & XS_IRREGULAR)) return a;
    xs_summary r;
    r.E = a.E;
    r.meta = a.meta | (b.meta & XS_IRREGULAR);
    if (a.E != b.E || ((a.meta ^ b.meta) & 2)) r.meta |= XS_IRREGULAR;
    r.D = xs_sat_add(a.D, b.D);
    const int64_t blo = xs_sat_add(a.D, b.lo), bhi = xs_sat_add(a.D, b.hi);
    r.lo = a.lo < blo ? a.lo : blo;
    r.hi = a.hi > bhi ? a.hi : bhi;
    return r;
}

// May a run with summary q be applied to a running sum currently at (E0, neg0, M)?
XS_HD bool xs_applies(const xs_summary &q, int32_t E0, int32_t neg0, int64_t M)
{
    if (q.meta & XS_IRREGULAR) return false;
    if (q.E < 0) return true;  // empty run
    return q.E == E0 && ((q.meta >> 1) & 1) == neg0 && xs_sat_add(M, q.lo) >= XS_TWO52 + 1 &&
           xs_sat_add(M, q.hi) <= XS_TWO53 - 2;
}
