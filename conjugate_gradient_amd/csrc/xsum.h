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
// Binade crossings.  A positive sum of N comparable terms crosses ~log2(N) powers of two, and every crossing used to
// send its cell (and the group around it) down the element-by-element path -- most of the serial walk's time.  An
// UPWARD crossing is still an integer map: with W = M + P the mantissa before the crossing element and tau = c + f
// (c = floor(tau), 0 < f < 1) that element in ulps, the exact sum (W + tau) u lies in the next binade, whose ulp is 2u,
// and RN gives  M' = rint((W + tau)/2) = (W + c + 1) >> 1  whatever the parity of W (f > 0 rules out the tie).  Further
// elements add integers in the new binade, so a run with K upward crossings is the map  M -> (M + A) >> K,  valid for
// the incoming mantissas in an interval [Mlo, Mhi] (those for which every crossing happens at the same element and
// nothing else leaves its binade).  Such maps compose:  ((M + A1) >> K1 + A2) >> K2 = (M + A1 + (A2 << K1)) >> (K1+K2),
// and the preimage of an interval under a floor-shift is an interval.  The crossing element is located with the
// PREDICTED mantissa; the interval makes the summary exact for whatever mantissa really arrives (or not applicable).
//
// Ties.  With k comparable terms already in the sum, tau has only ~log2(k) fractional bits, so an exact tie
// (tau = c + 1/2) turns up with probability ~1/k per element -- ~ln(N) ties per sum, as many as crossings.  RN then
// takes the even neighbour: with W = M + P before the element,  M' = (W + c + 1) & ~1.  A run with ONE tie is therefore
// the map  M -> ((M + A) & ~1) + C  (A = P + c + 1, C = what follows); it composes with plain runs on either side
// (they add to A or to C) and with nothing else.
//
// Hierarchy: cell = 128 elements, group = 64 cells.  The driver walks group summaries 64 at a time (plain runs are
// validated and applied with one prefix sum, a run holding crossings or a tie is applied by itself), drops to the
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
constexpr int64_t XS_BIG = 1LL << 61;       // |A| of any applicable run is far below this
constexpr int     XS_KMAX = 7;              // binade crossings one summary may hold

// meta bit0: not applicable whatever arrives (several ties / |tau| >= 2^52 / non-finite / no usable prediction / mixed
// binades / downward crossing); bit1: assumed sign negative; bit2: the run holds one tie; bits 8..15: K
enum { XS_IRREGULAR = 1, XS_NEG = 2, XS_TIE = 4 };

// The run maps an incoming mantissa M (binade E, sign as in meta), Mlo <= M <= Mhi, to
//     (M + A) >> K          in binade E + K      (plain run: K = 0;  K upward crossings otherwise;  C = 0)
//     ((M + A) & ~1) + C    in binade E          (XS_TIE)
struct xs_summary {
    int64_t A, C;
    int64_t Mlo, Mhi;
    int32_t E;       // assumed incoming biased exponent (1..2046); -1 = empty run (wildcard)
    int32_t meta;
};

XS_HD int xs_K(const xs_summary &q) { return (q.meta >> 8) & 0xff; }

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
    q.A = q.C = 0;
    q.Mlo = -XS_BIG;
    q.Mhi = XS_BIG;
    q.E = -1;
    q.meta = 0;
    return q;
}

XS_HD bool xs_is_identity(const xs_summary &q) { return q.E < 0 && !(q.meta & XS_IRREGULAR); }

// A run that stays in its binade: D = total movement of M, lo / hi = min / max over ALL prefixes of the run, the
// empty one included (lo <= 0 <= hi).  Regular for M iff M+lo >= 2^52+1 and M+hi <= 2^53-2 (the margins keep the
// value before rounding inside the binade too).
XS_HD xs_summary xs_regular(int64_t D, int64_t lo, int64_t hi, int32_t E, int32_t neg, bool bad)
{
    xs_summary q;
    q.A = D;
    q.C = 0;
    q.Mlo = XS_TWO52 + 1 - lo;
    q.Mhi = XS_TWO53 - 2 - hi;
    q.E = E;
    q.meta = (neg ? XS_NEG : 0) | (bad ? XS_IRREGULAR : 0);
    return q;
}

// A run with one upward crossing: part 1 (P = its total, lo1 / hi1 over its prefixes incl. the empty one and P itself)
// in binade E, then the crossing element tau = c + f (c = floor, f > 0 checked by the caller), then part 2 (D2, lo2,
// hi2 likewise) in binade E + 1.
XS_HD xs_summary xs_crossing(int64_t P, int64_t lo1, int64_t hi1, int64_t c, int64_t D2, int64_t lo2, int64_t hi2, int32_t E,
                             int32_t neg, bool bad)
{
    const int64_t B = P + c + 1;  // M' = (M + B) >> 1 right after the crossing element
    xs_summary    q;
    q.C = 0;
    q.A = B + 2 * D2;
    // part 1 stays inside; the crossing element lands at or above 2^53 + 2 ulps (so the sum IS in the next binade
    // whatever the fraction); part 2 stays inside the new binade:  M' >= L  <=>  M >= 2L - B,  M' <= H  <=>  M <= 2H + 1 - B
    int64_t lo = XS_TWO52 + 1 - lo1, hi = XS_TWO53 - 2 - hi1;
    const int64_t cross = XS_TWO53 + 2 - P - c, lo_b = 2 * (XS_TWO52 + 1 - lo2) - B, hi_b = 2 * (XS_TWO53 - 2 - hi2) + 1 - B;
    lo = lo > cross ? lo : cross;
    lo = lo > lo_b ? lo : lo_b;
    hi = hi < hi_b ? hi : hi_b;
    q.Mlo = lo;
    q.Mhi = hi;
    q.E = E;
    q.meta = (neg ? XS_NEG : 0) | (1 << 8) | ((bad || lo > hi || E + 1 > 2046) ? XS_IRREGULAR : 0);
    return q;
}

// A run with one exact tie tau = c + 1/2 (c = floor(tau)) between part 1 and part 2, all in binade E.
XS_HD xs_summary xs_tie(int64_t P, int64_t lo1, int64_t hi1, int64_t c, int64_t D2, int64_t lo2, int64_t hi2, int32_t E,
                        int32_t neg, bool bad)
{
    xs_summary q;
    q.A = P + c + 1;  // M' = (M + A) & ~1 right after the tie: the even one of W + c, W + c + 1
    q.C = D2;
    // M' is M + A - 1 or M + A: both must stay inside together with everything part 2 reaches from them
    int64_t       lo = XS_TWO52 + 1 - lo1, hi = XS_TWO53 - 2 - hi1;
    const int64_t lo_b = XS_TWO52 + 2 - lo2 - q.A, hi_b = XS_TWO53 - 2 - hi2 - q.A;
    lo = lo > lo_b ? lo : lo_b;
    hi = hi < hi_b ? hi : hi_b;
    q.Mlo = lo;
    q.Mhi = hi;
    q.E = E;
    q.meta = (neg ? XS_NEG : 0) | XS_TIE | ((bad || lo > hi) ? XS_IRREGULAR : 0);
    return q;
}

// the run made of the single element t, under the assumption (E, neg)
XS_HD xs_summary xs_single(double t, int32_t E, int32_t neg)
{
    bool          bad = false;
    const int64_t d = xs_element(t, E, neg, &bad);
    return xs_regular(d, d < 0 ? d : 0, d > 0 ? d : 0, E, neg, bad);
}

// summary of the concatenation a then b (b assumed in the binade a leaves, else irregular)
XS_HD xs_summary xs_join(const xs_summary &a, const xs_summary &b)
{
    if (xs_is_identity(a)) return b;
    if (xs_is_identity(b)) return a;
    const int  Ka = xs_K(a), Kb = xs_K(b);
    const bool ta = a.meta & XS_TIE, tb = b.meta & XS_TIE;
    xs_summary r;
    r.E = a.E;
    r.A = r.C = 0;
    r.Mlo = 1;
    r.Mhi = 0;
    bool irr = ((a.meta | b.meta) & XS_IRREGULAR) || ((a.meta ^ b.meta) & XS_NEG) || b.E != a.E + Ka || Ka + Kb > XS_KMAX ||
               a.E + Ka + Kb > 2046 || (ta && (tb || Kb)) || (tb && Ka);  // a tie composes with plain runs only
    int meta = a.meta & XS_NEG;
    if (!irr && tb) {  // plain, then the tie run: everything of b sees M + a.A
        r.A = a.A + b.A;
        r.C = b.C;
        const int64_t lo2 = b.Mlo - a.A, hi2 = b.Mhi - a.A;
        r.Mlo = a.Mlo > lo2 ? a.Mlo : lo2;
        r.Mhi = a.Mhi < hi2 ? a.Mhi : hi2;
        meta |= XS_TIE;
    } else if (!irr && ta) {  // the tie run, then plain: b sees M + a.A - 1 + a.C or M + a.A + a.C
        r.A = a.A;
        r.C = a.C + b.A;
        const int64_t lo2 = b.Mlo - a.A - a.C + 1, hi2 = b.Mhi - a.A - a.C;
        r.Mlo = a.Mlo > lo2 ? a.Mlo : lo2;
        r.Mhi = a.Mhi < hi2 ? a.Mhi : hi2;
        meta |= XS_TIE;
    } else if (!irr) {
        // bounds of b beyond the mantissa range bind nothing: clamp them so that the shifts below cannot overflow
        const int64_t cap = 1LL << 54;
        const int64_t blo = b.Mlo < 0 ? 0 : (b.Mlo > cap ? cap : b.Mlo), bhi = b.Mhi < -1 ? -1 : (b.Mhi > cap ? cap : b.Mhi);
        // (M + a.A) >> Ka >= blo  <=>  M >= (blo << Ka) - a.A ;   <= bhi  <=>  M <= ((bhi + 1) << Ka) - 1 - a.A
        const int64_t lo2 = blo * (1LL << Ka) - a.A, hi2 = (bhi + 1) * (1LL << Ka) - 1 - a.A;
        r.Mlo = a.Mlo > lo2 ? a.Mlo : lo2;
        r.Mhi = a.Mhi < hi2 ? a.Mhi : hi2;
        const int64_t lim = XS_BIG >> Ka;
        if (b.A >= lim || b.A <= -lim)
            irr = true;
        else
            r.A = a.A + b.A * (1LL << Ka);
        meta |= (Ka + Kb) << 8;
    }
    if (!irr) irr = r.A >= XS_BIG || r.A <= -XS_BIG || r.C >= XS_BIG || r.C <= -XS_BIG || r.Mlo > r.Mhi;
    r.meta = irr ? ((a.meta & XS_NEG) | XS_IRREGULAR) : meta;
    return r;
}

// May a run with summary q be applied to a running sum currently at (E0, neg0, M)?
XS_HD bool xs_applies(const xs_summary &q, int32_t E0, int32_t neg0, int64_t M)
{
    if (q.meta & XS_IRREGULAR) return false;
    if (q.E < 0) return true;  // empty run
    return q.E == E0 && ((q.meta >> 1) & 1) == neg0 && M >= q.Mlo && M <= q.Mhi;
}

// the running sum after an applicable run
XS_HD double xs_apply(const xs_summary &q, int32_t E0, int32_t neg0, int64_t M)
{
    if (q.E < 0) return xs_compose(E0, neg0, M);
    if (q.meta & XS_TIE) return xs_compose(E0, neg0, ((M + q.A) & ~1LL) + q.C);
    const int K = xs_K(q);
    return xs_compose(E0 + K, neg0, (M + q.A) >> K);
}

// 0: goes through the integer path (*d = rint(tau)); 1: exact tie (*d = floor(tau)); 2: cannot (huge, non-finite)
XS_HD int xs_classify(double t, int32_t E, int32_t neg, int64_t *d)
{
    const double tau = ldexp(neg ? -t : t, 1075 - E);
    *d = 0;
    if (!(fabs(tau) < 4503599627370496.0)) return 2;
    const double r = rint(tau);
    if (fabs(tau - r) == 0.5) {
        *d = (int64_t)floor(tau);
        return 1;
    }
    *d = (int64_t)r;
    return 0;
}

// Sequential statement of a cell summary (the device builds the same thing with 16 lanes, xsum.hip): the run t[0..cnt)
// assumed to start in binade (E, neg) with a mantissa near Mt.  The first element that is an exact tie, or that would
// take Mt out of the binade upwards, becomes the run's special element; a second one makes the run irregular.
XS_HD xs_summary xs_cell_summary(const double *t, int cnt, int32_t E, int32_t neg, int64_t Mt)
{
    int64_t D = 0, lo = 0, hi = 0, d;
    bool    bad = false, hard = false;
    int     kx = -1, type = 0;  // 1 tie, 2 crossing
    for (int k = 0; k < cnt; ++k) {
        const int cls = xs_classify(t[k], E, neg, &d);
        if (cls) d = 0, bad = true;
        hard = hard || cls == 2;
        if (kx < 0 && cls == 1) kx = k, type = 1;
        if (kx < 0 && cls == 0 && Mt + D + d > XS_TWO53 - 2) kx = k, type = 2;
        D += d;
        lo = D < lo ? D : lo;
        hi = D > hi ? D : hi;
    }
    const xs_summary plain = xs_regular(D, lo, hi, E, neg, bad);
    if (kx < 0 || hard) return plain;
    int64_t P = 0, lo1 = 0, hi1 = 0;
    for (int k = 0; k < kx; ++k) {
        xs_classify(t[k], E, neg, &d);
        P += d;
        lo1 = P < lo1 ? P : lo1;
        hi1 = P > hi1 ? P : hi1;
    }
    const double tau = ldexp(neg ? -t[kx] : t[kx], 1075 - E);
    const double c = floor(tau);
    if (type == 2 && tau == c) return plain;  // integer tau at a crossing: a tie in the new binade when W + c is odd
    const int32_t E2 = type == 2 ? E + 1 : E;
    int64_t       D2 = 0, lo2 = 0, hi2 = 0;
    for (int k = kx + 1; k < cnt; ++k) {
        if (xs_classify(t[k], E2, neg, &d)) return plain;  // a second special element
        D2 += d;
        lo2 = D2 < lo2 ? D2 : lo2;
        hi2 = D2 > hi2 ? D2 : hi2;
    }
    return type == 2 ? xs_crossing(P, lo1, hi1, (int64_t)c, D2, lo2, hi2, E, neg, false)
                     : xs_tie(P, lo1, hi1, (int64_t)c, D2, lo2, hi2, E, neg, false);
}
