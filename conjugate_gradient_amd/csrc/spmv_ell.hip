// spmv_ell.hip -- y = A x for the slot-major ("column-major") full ELL layout.
//
// Replaces reference src/ell.c:1004-1040 (mv_ell_symmetric_full_colmajor): entry (row i, slot s)
// lives at s*n+i, pad = col -1.  One thread owns R consecutive rows and keeps y[i] in a register,
// adding diag first and then slots 0..K-1 with pads skipped -- the same sequence of rounded
// multiplies and adds the reference performs on y[i], so the result is bit-identical.
//
// HBM-bound: per row it streams K*(8+4) matrix bytes + diag 8 + y 8 and gathers x.  Slot streams
// are read with 16-B (values) / 8-B (columns) per-lane loads, fully coalesced because consecutive
// lanes own consecutive row pairs of one slot.  Workgroups walk the 512-row tiles grid-stride with
// a grid that is a multiple of 8, so tile t always runs on the XCD group t%8 and the co-resident
// workgroups form one moving front; the gathered x lines (stencil neighbours) are then re-used
// from that XCD's L2 while the matrix streams through once.
//
// DOT=true additionally leaves sum_i x[i]*y[i] of the workgroup's rows in partials[blockIdx.x]
// (CG: x is p, y is Ap; saves a full re-read of p and Ap).
//
// Banded matrices (<= 255 distinct col - row offsets) get derived copies at their first SpMV (ell_dict_try_build): 1-byte
// column codes; rows wider than 8 slots also the values and code words tile-blocked (slot_at<3>) and, if the windows of x
// a tile reads fit 44 KiB of LDS, the kernel that gathers from LDS (spmv_ell_staged).  Same operands in the same order
// everywhere: every variant gives the same bits.
#include <algorithm>
#include <mutex>
#include <vector>

#include "cgh_internal.h"
#include "device_utils.h"
#include "halo_ll.h"

namespace {

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int    i2_t __attribute__((ext_vector_type(2)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));

// Matrix values/columns are read exactly once per SpMV: non-temporal loads keep them from
// displacing the gathered x lines in L2.
template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p)
{
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}

// Tuning variants (env CGH_ELL_VARIANT, default = the shipped one): bit0 = XCD-chunked tile walk,
// bit1 = non-temporal matrix loads, bit2 = ablation (no gather; wrong results), bit3 = non-temporal y store.
constexpr int VAR_DEFAULT = 11;
constexpr int CGH_ELL_SMALL_TILES = 256;

// Column ids of rows i0, i0+1 in slot s (-1 = pad).  DICT: one byte per entry, an index into the table of the
// matrix's distinct (col - row) offsets (staged in LDS), 255 = pad -- 9 instead of 12 bytes per stored entry.
template <bool NT, int DICT>
__device__ __forceinline__ i2_t load_cols(const void *__restrict__ cols, const int *tab, int64_t n, int s, int64_t i0)
{
    if constexpr (DICT == 1) {
        const unsigned short cc =
            stream_load<NT>(reinterpret_cast<const unsigned short *>(cols) + (((int64_t)s * n + i0) >> 1));
        const int c0 = cc & 0xff, c1 = cc >> 8;
        i2_t      c;
        c.x = c0 != 255 ? (int)i0 + tab[c0] : -1;
        c.y = c1 != 255 ? (int)i0 + 1 + tab[c1] : -1;
        return c;
    } else {
        return stream_load<NT>(reinterpret_cast<const i2_t *>(reinterpret_cast<const int *>(cols) + (int64_t)s * n + i0));
    }
}

// Where (slot s, row i0) lives.  Slot-major: s*n + i0 (the reference's column-major ELL).  DICT == 3, wide banded rows: the
// derived "tile-blocked" copy -- the 512 rows of a tile keep their K slots together, slot after slot, so a workgroup reads
// ONE contiguous run of K*4 KiB (+ the code words) per tile instead of K runs that lie n*8 bytes apart (26 streams 134 MB
// apart at 256^3, 1 GiB apart at 512^3: one page per stream and wave).  `per_tile` = slots (or code words) per row.
template <int DICT>
__device__ __forceinline__ int64_t slot_at(int64_t n, int per_tile, int s, int64_t i0)
{
    if constexpr (DICT == 3)
        return ((i0 >> 9) * per_tile + s) * 512 + (i0 & 511);
    else
        return (int64_t)s * n + i0;
}

// x[c] for a stored column id (pads, c < 0, read the row's own entry and are discarded by the caller).  HALO: ids from
// hl.nl on are halo entries of a multi-GPU block, read from the deposit buffer the neighbours write into (halo_ll.h) --
// the wait for a neighbour's push happens here, in the lanes that need the entry, and nowhere else.
template <int HALO>
__device__ __forceinline__ double gather_x(const double *__restrict__ x, int c, int64_t i0, const cgh_ll_halo &hl)
{
    if constexpr (HALO != 0) {
        if ((int64_t)c >= hl.nl) return ll_read(hl, (int64_t)c - hl.nl);
    }
    return x[c >= 0 ? (int64_t)c : i0];
}

template <int KT, int DOT, int VAR, int DICT = 0, int HALO = 0>
__global__ __launch_bounds__(CGH_BLOCK) void spmv_ell_r2(int64_t n, int Krt, const double *__restrict__ diag,
                                                        const double *__restrict__ vals, const void *__restrict__ cols,
                                                        const int *__restrict__ off_tab,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        double *__restrict__ partials, const int *__restrict__ done,
                                                        int rev, int64_t row_begin, int64_t row_end, int64_t row_begin2,
                                                        int64_t row_end2, cgh_ll_halo hl)
{
    __shared__ double lds[4];
    __shared__ int    tab[DICT ? 256 : 1];
    if (done != nullptr && *done != 0) return;
    if constexpr (DICT != 0) {
        tab[threadIdx.x] = off_tab[threadIdx.x];
        __syncthreads();
    }
    const int     K = KT > 0 ? KT : Krt;
    // rows [row_begin,row_end) followed by rows [row_begin2,row_end2) (all four even on this path): one launch serves the
    // two boundary row ranges of a multi-GPU block
    const int64_t pair0 = row_begin >> 1, pairs1 = (row_end >> 1) - pair0;
    const int64_t pair0b = row_begin2 >> 1;
    const int64_t pairs = pairs1 + ((row_end2 >> 1) - pair0b);
    double        local = 0.0;

    // Tile walk.  Workgroup-iteration j (= blockIdx + k*gridDim, gridDim a multiple of 8) runs on XCD
    // group j%8.  Tiles are dealt to XCD groups in chunks of 8 consecutive tiles (4096 rows):
    // t = 64*(q/8) + 8*(j%8) + q%8 with q = j/8, a bijection on every aligned group of 64 tiles.  A row's
    // +-1 / +-nx neighbours then almost always belong to the same XCD's chunk, so each x line is fetched
    // into one L2 instead of two or three (measured: SpMV FETCH_SIZE, profiles/).
    // Up to CGH_ELL_SMALL_TILES tiles everything is resident at once and lives in the caches: tile t is simply workgroup t's
    // (ell_grid_rows gives every tile its own workgroup), which is also the shape the resident CG launch of small systems
    // reproduces bit for bit (cg_persistent.h).
    const int64_t tiles = (pairs + CGH_BLOCK - 1) / CGH_BLOCK;
    const int64_t tiles64 = (tiles + 63) & ~(int64_t)63;
    constexpr bool NT = (VAR & 2) != 0, NOGATHER = (VAR & 4) != 0, NTSTORE = (VAR & 8) != 0;
    const bool     REMAP = (VAR & 1) != 0 && tiles > CGH_ELL_SMALL_TILES;
    for (int64_t jj = blockIdx.x; jj < tiles64; jj += gridDim.x) {
        const int64_t j = rev ? tiles64 - 1 - jj : jj;
        const int64_t q = j >> 3;
        const int64_t t = REMAP ? ((q >> 3) << 6) + ((j & 7) << 3) + (q & 7) : j;
        const int64_t pr = t * CGH_BLOCK + threadIdx.x;
        if (pr >= pairs) continue;
        const int64_t i0 = (pr < pairs1 ? pair0 + pr : pair0b + (pr - pairs1)) << 1;
        const double2 d = *reinterpret_cast<const double2 *>(diag + i0);
        const double2 xi = *reinterpret_cast<const double2 *>(x + i0);
        double        a0 = add_rn(0.0, mul_rn(d.x, xi.x));
        double        a1 = add_rn(0.0, mul_rn(d.y, xi.y));
        if constexpr (KT > 0) {
            d2_t v[KT];
            i2_t c[KT];
            if constexpr (DICT >= 2) {
                // wide rows: the codes of 4 slots of one row share a 32-bit word (word q of row i at q*n+i), so a lane
                // fetches 8 bytes per 4 slots of its row pair instead of four 2-byte loads
                constexpr int NW = (KT + 3) / 4;
                u2_t          w[NW];
#pragma unroll
                for (int q = 0; q < NW; ++q)
                    w[q] = stream_load<NT>(reinterpret_cast<const u2_t *>(reinterpret_cast<const unsigned *>(cols) + slot_at<DICT>(n, NW, q, i0)));
#pragma unroll
                for (int s = 0; s < KT; ++s) v[s] = stream_load<NT>(reinterpret_cast<const d2_t *>(vals + slot_at<DICT>(n, KT, s, i0)));
#pragma unroll
                for (int s = 0; s < KT; ++s) {
                    const int c0 = (w[s >> 2].x >> (8 * (s & 3))) & 0xff, c1 = (w[s >> 2].y >> (8 * (s & 3))) & 0xff;
                    c[s].x = c0 != 255 ? (int)i0 + tab[c0] : -1;
                    c[s].y = c1 != 255 ? (int)i0 + 1 + tab[c1] : -1;
                }
            } else {
#pragma unroll
                for (int s = 0; s < KT; ++s) {
                    c[s] = load_cols<NT, DICT>(cols, tab, n, s, i0);
                    v[s] = stream_load<NT>(reinterpret_cast<const d2_t *>(vals + (int64_t)s * n + i0));
                }
            }
            double g0[KT], g1[KT];
#pragma unroll
            for (int s = 0; s < KT; ++s) {
                if constexpr (NOGATHER) {  // ablation only: wrong results
                    g0[s] = xi.x + (double)c[s].x;
                    g1[s] = xi.y + (double)c[s].y;
                } else {
                    g0[s] = gather_x<HALO>(x, c[s].x, i0, hl);
                    g1[s] = gather_x<HALO>(x, c[s].y, i0, hl);
                }
            }
#pragma unroll
            for (int s = 0; s < KT; ++s) {
                const double t0 = madd_rn(a0, v[s].x, g0[s]);
                const double t1 = madd_rn(a1, v[s].y, g1[s]);
                a0 = c[s].x >= 0 ? t0 : a0;
                a1 = c[s].y >= 0 ? t1 : a1;
            }
        } else {
            u2_t w;
            w.x = w.y = 0;
#pragma unroll 4
            for (int s = 0; s < K; ++s) {
                i2_t c;
                if constexpr (DICT >= 2) {
                    if ((s & 3) == 0)
                        w = stream_load<NT>(reinterpret_cast<const u2_t *>(reinterpret_cast<const unsigned *>(cols) + slot_at<DICT>(n, (K + 3) / 4, s >> 2, i0)));
                    const int c0 = (w.x >> (8 * (s & 3))) & 0xff, c1 = (w.y >> (8 * (s & 3))) & 0xff;
                    c.x = c0 != 255 ? (int)i0 + tab[c0] : -1;
                    c.y = c1 != 255 ? (int)i0 + 1 + tab[c1] : -1;
                } else {
                    c = load_cols<NT, DICT>(cols, tab, n, s, i0);
                }
                const d2_t    v = stream_load<NT>(reinterpret_cast<const d2_t *>(vals + slot_at<DICT>(n, K, s, i0)));
                const double  g0 = gather_x<HALO>(x, c.x, i0, hl);
                const double  g1 = gather_x<HALO>(x, c.y, i0, hl);
                const double  t0 = madd_rn(a0, v.x, g0);
                const double  t1 = madd_rn(a1, v.y, g1);
                a0 = c.x >= 0 ? t0 : a0;
                a1 = c.y >= 0 ? t1 : a1;
            }
        }
        if constexpr (NTSTORE) {
            d2_t o;
            o.x = a0;
            o.y = a1;
            __builtin_nontemporal_store(o, reinterpret_cast<d2_t *>(y + i0));
        } else {
            *reinterpret_cast<double2 *>(y + i0) = make_double2(a0, a1);
        }
        if constexpr (DOT == 1) local += xi.x * a0 + xi.y * a1;
        if constexpr (DOT == 2) {
            // strict mode: a wave's 64 row pairs are one 128-row cell of the order-exact sums (xsum.hip); its approximate
            // x.y goes to partials[cell].  Launched only for one full range with n % 128 == 0: every wave is full or idle.
            const double q = wave_sum_lane63(xi.x * a0 + xi.y * a1);
            if ((threadIdx.x & 63) == 63) partials[pr >> 6] = q;
        }
    }
    if constexpr (DOT == 1) {
        const double tot = block_sum_256(local, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

// The same SpMV with the gathers served from LDS (banded matrices with dictionary-coded columns, one full row range).  A
// tile's 512 rows read x at row + d for the matrix's few distinct offsets d: per offset one contiguous run of 512 entries,
// and runs of neighbouring offsets overlap -- the plan (host, ell_stage_plan) merges them into at most 12 windows
// [first, first + len) relative to the tile's first row (7-point 256^3: 512 + 1024 + 512 doubles, 16 KiB; 27-point: three
// windows of 1028, 24 KiB).  The workgroup loads the windows with coalesced 16-byte loads -- every line once per tile,
// where the 2 K gather instructions of a wave requested each line of x again and again from L1 / L2 -- and the lanes
// gather from LDS.  A window is stored as its even and its odd entries apart, so the 64 lanes of one gather (rows 2 j, or
// 2 j + 1, at one offset) read consecutive doubles: no bank conflicts.  Tables t0 / t1 give, per code, the LDS index of the
// operand of lane 0's first / second row; lane j adds j.  Same operands, same order of operations: same bits.
struct cgh_stage_plan {
    int nwin = 0, total = 0;  // windows; doubles of LDS in all
    int first[CGH_STAGE_WINDOWS], len[CGH_STAGE_WINDOWS], base[CGH_STAGE_WINDOWS];
};

// TT: 512-row tiles per staging round.  With two, neighbouring tiles share what they stage -- a 512-row tile of an nx = 512 grid
// is ONE grid line and stages three lines per plane for it, two tiles stage four for two -- at twice the LDS per workgroup.
template <int KT, int DOT, int VAR, int DICT, int TT>
__global__ __launch_bounds__(CGH_BLOCK) void spmv_ell_staged(int64_t n, int Krt, const double *__restrict__ diag, const double *__restrict__ vals,
                                                            const void *__restrict__ cols, const int *__restrict__ stage_tab,
                                                            const double *__restrict__ x, double *__restrict__ y,
                                                            double *__restrict__ partials, const int *__restrict__ done, int rev,
                                                            int64_t row_begin, int64_t row_end, cgh_stage_plan sp)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];
    __shared__ double lds[4];
    __shared__ int    t0s[256], t1s[256];
    if (done != nullptr && *done != 0) return;
    t0s[threadIdx.x] = stage_tab[threadIdx.x];
    t1s[threadIdx.x] = stage_tab[256 + threadIdx.x];  // (the first round's barrier comes before anybody reads them)
    const int64_t pair0 = row_begin >> 1, pairs = (row_end >> 1) - pair0;
    double        local = 0.0;
    constexpr int RP = CGH_BLOCK * TT;  // row pairs per staging round
    const int64_t tiles = (pairs + RP - 1) / RP;
    const int64_t tiles64 = (tiles + 63) & ~(int64_t)63;
    constexpr bool NT = (VAR & 2) != 0, NTSTORE = (VAR & 8) != 0;
    const bool     REMAP = (VAR & 1) != 0 && tiles > CGH_ELL_SMALL_TILES;
    constexpr int  NW = (KT + 3) / 4;
    constexpr int  KA = KT > 0 ? KT : 1, NWA = DICT >= 2 && NW > 0 ? NW : 1;  // (array sizes of the compile-time path)
    const int      K = KT > 0 ? KT : Krt;
    for (int64_t jj = blockIdx.x; jj < tiles64; jj += gridDim.x) {
        const int64_t j = rev ? tiles64 - 1 - jj : jj;
        const int64_t q = j >> 3;
        const int64_t t = REMAP ? ((q >> 3) << 6) + ((j & 7) << 3) + (q & 7) : j;
        if (t * RP >= pairs) continue;  // (the whole workgroup)
        // compile-time width: the first tile's matrix entries go out before the staging (they have the longest way)
        d2_t           v[KA];
        u2_t           w[NWA];
        unsigned short cb[DICT >= 2 ? 1 : KA];
        auto load_entries = [&](int64_t i0) {
            if constexpr (KT > 0) {
                if constexpr (DICT >= 2) {
#pragma unroll
                    for (int qq = 0; qq < NW; ++qq)
                        w[qq] = stream_load<NT>(reinterpret_cast<const u2_t *>(reinterpret_cast<const unsigned *>(cols) + slot_at<DICT>(n, NW, qq, i0)));
                } else {
#pragma unroll
                    for (int s = 0; s < KT; ++s)
                        cb[s] = stream_load<NT>(reinterpret_cast<const unsigned short *>(cols) + (((int64_t)s * n + i0) >> 1));
                }
#pragma unroll
                for (int s = 0; s < KT; ++s) v[s] = stream_load<NT>(reinterpret_cast<const d2_t *>(vals + slot_at<DICT>(n, KT, s, i0)));
            }
        };
        auto first_row = [&](int sub, bool &mine, int64_t &pr) {
            pr = t * RP + (int64_t)sub * CGH_BLOCK + threadIdx.x;
            mine = pr < pairs;
            return (pair0 + (mine ? pr : t * RP + (int64_t)sub * CGH_BLOCK)) << 1;  // (idle lanes: the tile's first rows)
        };
        bool    mine;
        int64_t pr;
        int64_t i0 = first_row(0, mine, pr);
        load_entries(i0);
        double2 d = *reinterpret_cast<const double2 *>(diag + i0);
        double2 xi = *reinterpret_cast<const double2 *>(x + i0);
        __syncthreads();  // the previous round's gathers are over
        const int64_t ts = (pair0 + t * RP) << 1;
        for (int wi = 0; wi < sp.nwin; ++wi) {
            const int     half = sp.len[wi] >> 1, base = sp.base[wi];
            const int64_t e0 = ts + sp.first[wi];  // even
            for (int k = threadIdx.x; k < half; k += CGH_BLOCK) {
                const int64_t e = e0 + 2 * (int64_t)k;
                double2       xv = make_double2(0.0, 0.0);
                if (e >= 0 && e < n) xv = *reinterpret_cast<const double2 *>(x + e);  // (n is even: e + 1 < n as well)
                xs[base + k] = xv.x;
                xs[base + half + k] = xv.y;
            }
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < TT; ++sub) {
            if (sub > 0) {
                if (t * RP + (int64_t)sub * CGH_BLOCK >= pairs) break;  // (the whole workgroup; no barrier inside this loop)
                i0 = first_row(sub, mine, pr);
                load_entries(i0);
                d = *reinterpret_cast<const double2 *>(diag + i0);
                xi = *reinterpret_cast<const double2 *>(x + i0);
            }
            if (!mine) continue;
            const int     lane = (int)threadIdx.x + sub * CGH_BLOCK;  // row pair inside the staged round
            double        a0 = add_rn(0.0, mul_rn(d.x, xi.x));
            double        a1 = add_rn(0.0, mul_rn(d.y, xi.y));
            if constexpr (KT > 0) {
#pragma unroll
                for (int s = 0; s < KT; ++s) {
                    int c0, c1;
                    if constexpr (DICT >= 2) {
                        c0 = (w[s >> 2].x >> (8 * (s & 3))) & 0xff;
                        c1 = (w[s >> 2].y >> (8 * (s & 3))) & 0xff;
                    } else {
                        c0 = cb[s] & 0xff;
                        c1 = cb[s] >> 8;
                    }
                    const double g0 = xs[t0s[c0] + lane];
                    const double g1 = xs[t1s[c1] + lane];
                    const double p0 = madd_rn(a0, v[s].x, g0);
                    const double p1 = madd_rn(a1, v[s].y, g1);
                    a0 = c0 != 255 ? p0 : a0;
                    a1 = c1 != 255 ? p1 : a1;
                }
            } else {
                // any width (run-time K): slot after slot -- codes, value, two gathers from LDS
                const int nw = (K + 3) / 4;
                u2_t      wr;
                wr.x = wr.y = 0;
#pragma unroll 4
                for (int s = 0; s < K; ++s) {
                    int c0, c1;
                    if constexpr (DICT >= 2) {
                        if ((s & 3) == 0)
                            wr = stream_load<NT>(reinterpret_cast<const u2_t *>(reinterpret_cast<const unsigned *>(cols) + slot_at<DICT>(n, nw, s >> 2, i0)));
                        c0 = (wr.x >> (8 * (s & 3))) & 0xff;
                        c1 = (wr.y >> (8 * (s & 3))) & 0xff;
                    } else {
                        const unsigned short cc = stream_load<NT>(reinterpret_cast<const unsigned short *>(cols) + (((int64_t)s * n + i0) >> 1));
                        c0 = cc & 0xff;
                        c1 = cc >> 8;
                    }
                    const d2_t   vv = stream_load<NT>(reinterpret_cast<const d2_t *>(vals + slot_at<DICT>(n, K, s, i0)));
                    const double g0 = xs[t0s[c0] + lane];
                    const double g1 = xs[t1s[c1] + lane];
                    const double p0 = madd_rn(a0, vv.x, g0);
                    const double p1 = madd_rn(a1, vv.y, g1);
                    a0 = c0 != 255 ? p0 : a0;
                    a1 = c1 != 255 ? p1 : a1;
                }
            }
            if constexpr (NTSTORE) {
                d2_t o;
                o.x = a0;
                o.y = a1;
                __builtin_nontemporal_store(o, reinterpret_cast<d2_t *>(y + i0));
            } else {
                *reinterpret_cast<double2 *>(y + i0) = make_double2(a0, a1);
            }
            if constexpr (DOT == 1) local += xi.x * a0 + xi.y * a1;
            if constexpr (DOT == 2) {
                const double qsum = wave_sum_lane63(xi.x * a0 + xi.y * a1);  // (n % 128 == 0: every wave is full or idle)
                if ((threadIdx.x & 63) == 63) partials[pr >> 6] = qsum;
            }
        }
    }
    if constexpr (DOT == 1) {
        const double tot = block_sum_256(local, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

// Any n, any alignment: one row per thread.
template <bool DOT>
__global__ __launch_bounds__(CGH_BLOCK) void spmv_ell_r1(int64_t n, int K, const double *__restrict__ diag,
                                                        const double *__restrict__ vals, const int *__restrict__ cols,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        double *__restrict__ partials, const int *__restrict__ done,
                                                        int64_t row_begin, int64_t row_end)
{
    __shared__ double lds[4];
    if (done != nullptr && *done != 0) return;
    double local = 0.0;
    for (int64_t i = row_begin + (int64_t)blockIdx.x * CGH_BLOCK + threadIdx.x; i < row_end; i += (int64_t)gridDim.x * CGH_BLOCK) {
        const double xi = x[i];
        double       a = add_rn(0.0, mul_rn(diag[i], xi));
#pragma unroll 4
        for (int s = 0; s < K; ++s) {
            const int    c = cols[(int64_t)s * n + i];
            const double v = vals[(int64_t)s * n + i];
            const double g = x[c >= 0 ? (int64_t)c : i];
            const double t = madd_rn(a, v, g);
            a = c >= 0 ? t : a;
        }
        y[i] = a;
        if constexpr (DOT) local += xi * a;
    }
    if constexpr (DOT) {
        const double tot = block_sum_256(local, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int DOT, int VAR, int DICT = 0, int HALO = 0>
void launch_r2(int K, int grid, hipStream_t s, int64_t n, const double *diag, const double *vals, const void *cols,
               const int *off_tab, const double *x, double *y, double *partials, const int *done, int64_t rb, int64_t re,
               int64_t rb2, int64_t re2, int reverse = 0, cgh_ll_halo hl = cgh_ll_halo{nullptr, 0, 0, 0u, nullptr})
{
#define CGH_ELL_CASE(KK)                                                                                       \
    case KK:                                                                                                   \
        hipLaunchKernelGGL((spmv_ell_r2<KK, DOT, VAR, DICT, HALO>), dim3(grid), dim3(CGH_BLOCK), 0, s, n, K, diag, vals, cols, \
                           off_tab, x, y, partials, done, reverse, rb, re, rb2, re2, hl);                             \
        break;
    switch (g_cfg.ell_generic ? -1 : K) {
        CGH_ELL_CASE(1)
        CGH_ELL_CASE(2)
        CGH_ELL_CASE(3)
        CGH_ELL_CASE(4)
        CGH_ELL_CASE(5)
        CGH_ELL_CASE(6)
        CGH_ELL_CASE(7)
        CGH_ELL_CASE(8)
        CGH_ELL_CASE(13)
        CGH_ELL_CASE(26)
    default:
        hipLaunchKernelGGL((spmv_ell_r2<0, DOT, VAR, DICT, HALO>), dim3(grid), dim3(CGH_BLOCK), 0, s, n, K, diag, vals, cols, off_tab,
                           x, y, partials, done, reverse, rb, re, rb2, re2, hl);
    }
#undef CGH_ELL_CASE
}

// ------------------------------------------------------------------------------- dictionary-coded columns
constexpr int DICT_SLOTS = 512, DICT_EMPTY = INT32_MIN;

// distinct (col - row) offsets -> open-addressing table; gives up (overflow) beyond 255.
// The lanes of a wave mostly hold the SAME offset (consecutive rows of one slot), and at kernel start every wave
// finds the table empty: one elected lane per distinct offset does the insert, or the 64-fold same-address
// compare-and-swap stampede costs more than a millisecond (measured 1.4 ms on a 128k-row matrix).
__global__ __launch_bounds__(256) void dict_collect(int64_t total, int64_t n, const int *__restrict__ cols, int *table, int *count,
                                                   int *overflow)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t rounds = (total + stride - 1) / stride;
    for (int64_t it = 0; it < rounds; ++it) {
        if (__atomic_load_n(overflow, __ATOMIC_RELAXED)) return;
        const int64_t k = it * stride + (int64_t)blockIdx.x * 256 + threadIdx.x;
        const int     c = k < total ? cols[k] : -1;
        bool          pending = c != -1;
        const int     off = pending ? c - (int)(k % n) : 0;
        while (true) {
            const unsigned long long todo = __ballot(pending);
            if (todo == 0ull) break;
            const int leader = (int)__builtin_ctzll(todo);
            const int loff = __shfl(off, leader, 64);
            if (pending && off == loff) {
                pending = false;
                if ((int)(threadIdx.x & 63) == leader) {
                    unsigned h = ((unsigned)off * 2654435761u) >> 23;  // 9 bits
                    for (int probe = 0;; ++probe) {
                        if (probe == DICT_SLOTS) {
                            atomicExch(overflow, 1);
                            break;
                        }
                        // an unstructured matrix has a distinct offset in almost every entry: once the 256th has been
                        // inserted everybody else must stop probing the (full) table -- 3.6 ms of atomics on foam-32k otherwise
                        if ((probe & 3) == 3 && __atomic_load_n(overflow, __ATOMIC_RELAXED)) break;
                        int cur = __atomic_load_n(&table[h], __ATOMIC_RELAXED);
                        if (cur == DICT_EMPTY) {
                            cur = atomicCAS(&table[h], DICT_EMPTY, off);
                            if (cur == DICT_EMPTY) {
                                if (atomicAdd(count, 1) >= 255) atomicExch(overflow, 1);
                                break;
                            }
                        }
                        if (cur == off) break;
                        h = (h + 1) & (DICT_SLOTS - 1);
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void dict_encode(int64_t total, int64_t n, const int *__restrict__ cols,
                                                  const int *__restrict__ sorted, int count, unsigned char *__restrict__ out)
{
    __shared__ int tab[256];
    tab[threadIdx.x] = (int)threadIdx.x < count ? sorted[threadIdx.x] : INT32_MAX;
    __syncthreads();
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int c = cols[k];
        int       code = 255;
        if (c != -1) {
            const int off = c - (int)(k % n);
            int       lo = 0, hi = count - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (tab[mid] < off) lo = mid + 1;
                else hi = mid;
            }
            code = lo;
        }
        out[k] = (unsigned char)code;
    }
}

// packed layout: word q of row i (at q*n+i) = codes of slots 4q..4q+3 of that row, low byte first; 255 beyond K
__global__ __launch_bounds__(256) void dict_encode_packed(int64_t n, int K, const int *__restrict__ cols, const int *__restrict__ sorted,
                                                         int count, unsigned *__restrict__ out)
{
    __shared__ int tab[256];
    tab[threadIdx.x] = (int)threadIdx.x < count ? sorted[threadIdx.x] : INT32_MAX;
    __syncthreads();
    const int64_t words = (int64_t)((K + 3) / 4) * n;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < words; k += (int64_t)gridDim.x * 256) {
        const int64_t i = k % n;
        const int     q = (int)(k / n);
        unsigned      word = 0;
        for (int j = 0; j < 4; ++j) {
            const int sl = 4 * q + j;
            int       code = 255;
            const int c = sl < K ? cols[(int64_t)sl * n + i] : -1;
            if (c != -1) {
                const int off = c - (int)i;
                int       lo = 0, hi = count - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (tab[mid] < off) lo = mid + 1;
                    else hi = mid;
                }
                code = lo;
            }
            word |= (unsigned)code << (8 * j);
        }
        out[k] = word;
    }
}

// tile-blocked copies for wide rows (slot_at<3>): the values, and the packed code words written where the kernel reads them.
// Rows beyond n in the last tile hold zeros / pads (never read: the kernel's row pairs end at n).
__global__ __launch_bounds__(256) void ell_block_vals(int64_t n, int K, const double *__restrict__ vals, double *__restrict__ out)
{
    const int64_t total = ((n + 511) >> 9) * (int64_t)K * 512;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t tile = k / ((int64_t)K * 512);
        const int     s = (int)((k >> 9) % K);
        const int64_t i = tile * 512 + (k & 511);
        out[k] = i < n ? vals[(int64_t)s * n + i] : 0.0;
    }
}

__global__ __launch_bounds__(256) void ell_block_words(int64_t n, int NW, const unsigned *__restrict__ words, unsigned *__restrict__ out)
{
    const int64_t total = ((n + 511) >> 9) * (int64_t)NW * 512;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t tile = k / ((int64_t)NW * 512);
        const int     q = (int)((k >> 9) % NW);
        const int64_t i = tile * 512 + (k & 511);
        out[k] = i < n ? words[(int64_t)q * n + i] : 0xffffffffu;
    }
}

// largest column id stored (pads are -1): the staged kernel only serves matrices whose columns all lie below n (a
// multi-GPU block's halo columns do not)
__global__ __launch_bounds__(256) void ell_max_col(int64_t total, const int *__restrict__ cols, int *out)
{
    int m = -1;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) m = max(m, cols[k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// LDS per workgroup at most: 44 KiB with one tile per staging round, 50 KiB with two (three workgroups of 153 VGPRs per CU still fit)
// smallest column stored under each of the `count` sorted offsets (INT_MAX: none).  An offset whose every column lies at or
// beyond n belongs to halo entries of a multi-GPU block only: interior rows never use it, the staging plan leaves it out.
__global__ __launch_bounds__(256) void ell_offset_min_col(int64_t total, int64_t n, const int *__restrict__ cols, const int *__restrict__ sorted,
                                                         int count, int *__restrict__ mincol)
{
    __shared__ int tab[256], lmin[256];
    tab[threadIdx.x] = (int)threadIdx.x < count ? sorted[threadIdx.x] : INT32_MAX;
    lmin[threadIdx.x] = INT32_MAX;
    __syncthreads();
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int c = cols[k];
        if (c == -1) continue;
        const int off = c - (int)(k % n);
        int       lo = 0, hi = count - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tab[mid] < off) lo = mid + 1;
            else hi = mid;
        }
        if (c < lmin[lo]) atomicMin(&lmin[lo], c);
    }
    __syncthreads();
    if ((int)threadIdx.x < count && lmin[threadIdx.x] != INT32_MAX) atomicMin(&mincol[threadIdx.x], lmin[threadIdx.x]);
}

constexpr int STAGE_MAX_WINDOWS = CGH_STAGE_WINDOWS, STAGE_MAX_DOUBLES = 5632, STAGE_MAX_DOUBLES_2 = 6400;

// Windows of x a 512-row tile needs, from the sorted offsets: [d, d + 512) per offset d, overlapping or abutting runs
// merged; starts and lengths even (16-byte loads), each window stored as even entries then odd entries.  tab[256 + c] /
// tab[512 + c]: LDS index of the operand of lane 0's first / second row for code c.  Returns false if the plan does not
// fit (too many windows, too much LDS).
bool ell_stage_plan(const std::vector<int> &offs, int count, const cgh_matrix *m, std::vector<int> &tab, int rows, int max_doubles,
                    const std::vector<char> &halo_only)
{
    int nwin = 0, total = 0, first[CGH_STAGE_WINDOWS], len[CGH_STAGE_WINDOWS], base[CGH_STAGE_WINDOWS], win_of[256];
    for (int k = 0; k < count;) {
        if (halo_only[k]) {  // (never read by the rows a staged launch serves)
            win_of[k] = -1;
            ++k;
            continue;
        }
        if (nwin == STAGE_MAX_WINDOWS) return false;
        const int f = offs[k] & ~1;  // rounded down to even
        int       last = offs[k];
        win_of[k] = nwin;
        int kk = k + 1;
        while (kk < count && (halo_only[kk] || (int64_t)offs[kk] < (int64_t)last + rows)) {
            if (halo_only[kk]) {
                win_of[kk] = -1;
            } else {
                last = offs[kk];
                win_of[kk] = nwin;
            }
            ++kk;
        }
        const int64_t l = ((int64_t)last - f + rows + 1) & ~(int64_t)1;
        if (l > max_doubles) return false;
        first[nwin] = f;
        len[nwin] = (int)l;
        base[nwin] = total;
        total += (int)l;
        if (total > max_doubles) return false;
        ++nwin;
        k = kk;
    }
    if (nwin == 0 || total < rows) return false;
    for (int c = 0; c < 256; ++c) tab[256 + c] = tab[512 + c] = 0;
    for (int c = 0; c < count; ++c) {
        if (win_of[c] < 0) continue;
        const int w = win_of[c], e = offs[c] - first[w], half = len[w] >> 1;
        tab[256 + c] = base[w] + (e & 1) * half + (e >> 1);
        tab[512 + c] = base[w] + ((e + 1) & 1) * half + ((e + 1) >> 1);
    }
    m->stage_nwin = nwin;
    m->stage_total = total;
    m->stage_tiles = rows / 512;
    for (int w = 0; w < nwin; ++w) {
        m->stage_first[w] = first[w];
        m->stage_len[w] = len[w];
        m->stage_base[w] = base[w];
    }
    return true;
}

std::mutex g_dict_mutex;

// Build m->cols8 / m->off_tab if the matrix has at most 255 distinct diagonals.  Synchronises the stream.  Called under
// g_dict_mutex; dict_state becomes 1 (in use) only after everything it guards is in place, 2 (not applicable) otherwise.
void ell_dict_try_build(const cgh_matrix *m, hipStream_t s)
{
    struct publish {  // whatever path leaves this function: "tried" at the very end
        const cgh_matrix *m;
        int               state = 2;
        ~publish() { __atomic_store_n(&m->dict_state, state, __ATOMIC_RELEASE); }
    } done{m};
    const int     enabled = g_cfg.ell_dict < 0 ? 1 : g_cfg.ell_dict;
    const int64_t total = m->n * (int64_t)m->K;
    if (!enabled || m->K <= 0 || total < 2) return;
    int *work = (int *)cgh_alloc(sizeof(int) * (DICT_SLOTS + 2));
    if (!work) return;
    std::vector<int> h(DICT_SLOTS + 2, DICT_EMPTY);
    h[DICT_SLOTS] = h[DICT_SLOTS + 1] = 0;
    const int grid = cgh_stream_grid((total + 255) / 256, 8);
    bool      ok = cgh_h2d(work, h.data(), sizeof(int) * h.size()) == CGH_OK;
    if (ok) {
        hipLaunchKernelGGL(dict_collect, dim3(grid), dim3(256), 0, s, total, m->n, (const int *)m->cols, work, work + DICT_SLOTS,
                           work + DICT_SLOTS + 1);
        ok = hipGetLastError() == hipSuccess && cgh_d2h(h.data(), work, sizeof(int) * h.size()) == CGH_OK;
    }
    cgh_free(work);
    if (!ok || h[DICT_SLOTS + 1] != 0 || h[DICT_SLOTS] > 255) return;
    std::vector<int> offs;
    for (int k = 0; k < DICT_SLOTS; ++k)
        if (h[k] != DICT_EMPTY) offs.push_back(h[k]);
    std::sort(offs.begin(), offs.end());
    const int count = (int)offs.size();
    offs.resize(256, 0);
    // gathers from LDS (spmv_ell_staged): the plan of the windows of x a staging round reads, if they fit
    std::vector<int> tab(768, 0);
    bool             staged = false;
    if (g_cfg.ell_stage && m->n % 2 == 0) {
        int *mx = (int *)cgh_alloc(sizeof(int));
        int  hmx = -1;
        if (mx && cgh_h2d(mx, &hmx, sizeof(int)) == CGH_OK) {
            hipLaunchKernelGGL(ell_max_col, dim3(grid), dim3(256), 0, s, total, (const int *)m->cols, mx);
            std::vector<char> halo_only(256, 0);
            bool              have = hipGetLastError() == hipSuccess && cgh_d2h(&hmx, mx, sizeof(int)) == CGH_OK;
            if (have && (int64_t)hmx >= m->n) {
                // a multi-GPU block: which offsets are used by halo entries only?
                int *srt = (int *)cgh_alloc(sizeof(int) * 512);
                std::vector<int> init(512, INT32_MAX);
                for (int k = 0; k < count; ++k) init[k] = offs[k];
                have = srt && cgh_h2d(srt, init.data(), sizeof(int) * 512) == CGH_OK;
                if (have) {
                    hipLaunchKernelGGL(ell_offset_min_col, dim3(grid), dim3(256), 0, s, total, m->n, (const int *)m->cols, (const int *)srt, count,
                                       srt + 256);
                    have = hipGetLastError() == hipSuccess && cgh_d2h(init.data(), srt, sizeof(int) * 512) == CGH_OK;
                }
                cgh_free(srt);
                for (int k = 0; have && k < count; ++k) halo_only[k] = (int64_t)init[256 + k] >= m->n;
            }
            if (have) {
                // columns at or beyond n (the halo entries of a multi-GPU block) are never staged (the windows end at n): such a
                // matrix is served for launches over its interior rows only (cgh_spmv_opts::interior_rows)
                // Wide rows: two tiles per staging round where that saves at least 30 % of the staged doubles per row -- an
                // nx = 512 grid, whose 512-row tile is ONE grid line staging three lines per plane (27-point 512^3: 6.22 ->
                // 5.59 ms; 512x512x128: 1.620 -> 1.590); at nx = 256 it saves 25 % and loses (0.718 -> 0.731 ms: the second
                // tile's matrix loads do not overlap any staging).  CGH_ELL_STAGE_TILES=1: always one; 3: two whenever they fit.
                const bool wide_rows = m->K > 8;
                const bool ok2 = wide_rows && g_cfg.ell_stage_tiles >= 2 && ell_stage_plan(offs, count, m, tab, 1024, STAGE_MAX_DOUBLES_2, halo_only);
                const int  total2 = m->stage_total;
                staged = ell_stage_plan(offs, count, m, tab, 512, STAGE_MAX_DOUBLES, halo_only);
                if (ok2 && (!staged || g_cfg.ell_stage_tiles >= 3 || (int64_t)total2 * 10 <= (int64_t)m->stage_total * 14))
                    staged = ell_stage_plan(offs, count, m, tab, 1024, STAGE_MAX_DOUBLES_2, halo_only);
                m->stage_any_row = (int64_t)hmx < m->n;
            }
        }
        cgh_free(mx);
    }
    for (int k = 0; k < 256; ++k) tab[k] = offs[k];
    // narrow rows: one byte per entry, slot-major; wide rows (more load instructions than bytes are worth): 4 slots
    // of a row per 32-bit word.  CGH_ELL_DICT=1 / 2 forces a layout.
    const int mode = enabled == 2 ? 2 : (g_cfg.ell_dict == 1 ? 1 : (m->K > 8 ? 2 : 1));
    const size_t code_bytes = mode == 2 ? (size_t)((m->K + 3) / 4) * (size_t)m->n * 4 : (size_t)total;
    m->off_tab = (int *)cgh_alloc(sizeof(int) * 768);
    m->cols8 = (unsigned char *)cgh_alloc(code_bytes);
    if (!m->off_tab || !m->cols8 || cgh_h2d(m->off_tab, tab.data(), sizeof(int) * 768) != CGH_OK) {
        cgh_ell_dict_free(m);
        return;
    }
    if (!staged) m->stage_nwin = 0;
    if (mode == 2)
        hipLaunchKernelGGL(dict_encode_packed, dim3(grid), dim3(256), 0, s, m->n, m->K, (const int *)m->cols, (const int *)m->off_tab,
                           count, reinterpret_cast<unsigned *>(m->cols8));
    else
        hipLaunchKernelGGL(dict_encode, dim3(grid), dim3(256), 0, s, total, m->n, (const int *)m->cols, (const int *)m->off_tab, count,
                           m->cols8);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        cgh_ell_dict_free(m);
        return;
    }
    int final_mode = mode;
    if (mode == 2 && g_cfg.ell_blocked) {
        // wide rows: tile-blocked copies of the values and of the code words (slot_at<3>).  The slot-major values stay (other
        // paths read them: the one-row-per-lane kernel, int32 columns, downloads); if the copy cannot be had, the packed
        // slot-major layout serves.
        const int     NW = (m->K + 3) / 4;
        const int64_t tiles = (m->n + 511) >> 9;
        double       *vb = (double *)cgh_alloc(sizeof(double) * (size_t)tiles * (size_t)m->K * 512);
        unsigned     *wb = (unsigned *)cgh_alloc(sizeof(unsigned) * (size_t)tiles * (size_t)NW * 512);
        bool          okb = vb && wb;
        if (okb) {
            hipLaunchKernelGGL(ell_block_vals, dim3(grid), dim3(256), 0, s, m->n, m->K, (const double *)m->vals, vb);
            hipLaunchKernelGGL(ell_block_words, dim3(grid), dim3(256), 0, s, m->n, NW, reinterpret_cast<const unsigned *>(m->cols8), wb);
            okb = hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        }
        if (okb) {
            cgh_free(m->cols8);
            m->cols8 = reinterpret_cast<unsigned char *>(wb);
            m->vals_blk = vb;
            final_mode = 3;
        } else {
            cgh_free(vb);
            cgh_free(wb);
        }
    }
    m->n_offsets = count;
    m->dict_mode = final_mode;
    done.state = 1;
}

} // namespace

extern "C" int cgh_matrix_dict_offsets(const cgh_matrix *m)
{
    if (!m || m->layout != CGH_LAYOUT_ELL) return -1;
    return m->dict_state == 0 ? -1 : (m->dict_state == 1 ? m->n_offsets : 0);
}

extern "C" int cgh_matrix_dict_mode(const cgh_matrix *m)
{
    if (!m || m->layout != CGH_LAYOUT_ELL) return 0;
    return __atomic_load_n(&m->dict_state, __ATOMIC_ACQUIRE) == 1 ? m->dict_mode : 0;
}

void cgh_ell_dict_free(const cgh_matrix *m)
{
    if (!m) return;
    cgh_free(m->cols8);
    cgh_free(m->off_tab);
    cgh_free(m->vals_blk);
    m->cols8 = nullptr;
    m->off_tab = nullptr;
    m->vals_blk = nullptr;
    m->stage_nwin = 0;
    if (m->dict_state == 1) m->dict_state = 2;
}

// Rows per thread on the vector path.
static inline bool ell_vector_path(const cgh_matrix *m, const double *x, const double *y)
{
    return (m->n % 2 == 0) && aligned16(x) && aligned16(y) && aligned16(m->vals) && aligned16(m->diag) &&
           (reinterpret_cast<uintptr_t>(m->cols) & 7u) == 0;
}

bool cgh_ell_vector_path(const cgh_matrix *m, const double *x, const double *y) { return ell_vector_path(m, x, y); }

// Does one CG iteration's working set on this matrix (matrix + the four vectors) fit the Infinity Cache with room to
// spare (`times` = 1), or `times` of them?  CGH_LLC_MB sets the limit (default 224; 0 = never).  Measured (us/iteration,
// 7-point, hints on -> off): 1M rows 38.9 -> 34.2, 2.1M 59.9 -> 53.7; at 4.2M (394 MB) only the vector kernels gain
// (106.9 -> 103.1) and the SpMV loses; at 8.4M everything loses (202.9 -> 233.2).
bool cgh_fits_llc(const cgh_matrix *m, double times)
{
    const double limit = g_cfg.llc_mb * 1048576.0;
    const double per_entry = m->dict_state == 1 ? 9.0 : 12.0;
    return (double)m->stored * per_entry + (double)m->n * 8.0 * 5.0 <= limit * times;
}

// Does the kernel that gathers from LDS (spmv_ell_staged) serve this matrix's full-range launches?  0 no; 1 narrow rows with
// byte codes (only on request, CGH_ELL_STAGE=2: the 7-point kernel gains nothing -- 0.203 against 0.200 ms stand-alone at
// 256^3, 385 against 375 us per CG iteration -- its 12 gathers per row pair are served by L1 / L2 as fast as the staging
// costs; the 27-point kernel's 52 are not); 2 wide rows with tile-blocked code words (K = 13 / 26).
static int ell_staged_kind(const cgh_matrix *m, bool interior_rows = false)
{
    if (m->dict_state != 1 || m->stage_nwin <= 0 || !g_cfg.ell_stage) return 0;
    if (!m->stage_any_row && !interior_rows) return 0;
    if (g_cfg.ell_stage >= 2 && m->dict_mode == 1 && m->K >= 1 && m->K <= 8) return 1;
    return m->dict_mode == 3 && m->K > 8 ? 2 : 0;  // (13 and 26 slots at compile time, any other width with a run-time K)
}

extern "C" int cgh_matrix_stage_doubles(const cgh_matrix *m)
{
    if (!m || m->layout != CGH_LAYOUT_ELL || __atomic_load_n(&m->dict_state, __ATOMIC_ACQUIRE) != 1) return 0;
    return ell_staged_kind(m) ? m->stage_total : 0;
}

// Workgroups per CU of the tile walk.  Narrow rows with byte-coded columns (the 7-point kernel: 64 VGPRs, so eight
// workgroups per CU would be resident) run fastest with FIVE per CU -- measured in round 3 at 256^3, us per CG iteration:
// 2048 workgroups 400.3, 1536 388.5, 1280 373.4, 1024 378.9, 768 396.6, 512 460 (grids that are no multiple of the CU count
// lose: 1152 393, 1408 411); 4.2M rows 101.9 -> 98.4, 8.4M 194.8 -> 189.8.  Wide rows (27-point, 220 VGPRs: two workgroups per
// CU resident whatever the grid) and int32 columns keep eight (0.845 vs 0.854 ms; 445.5 vs 450.1 us).  CGH_ELL_WGS overrides.
static int ell_wgs_per_cu(const cgh_matrix *m, bool interior_rows = false)
{
    if (g_cfg.ell_wgs > 0) return g_cfg.ell_wgs;
    // wide rows with the gathers staged in LDS (153 VGPRs at K = 26: three workgroups per CU are resident): one round of
    // resident workgroups -- 27-point 256^3: 0.716 ms with three per CU, 0.732 with two, 0.741 with eight, 0.866 with four
    if (m != nullptr && ell_staged_kind(m, interior_rows) == 2) {
        if (!g_cfg.ell_generic && (m->K == 13 || m->K == 26)) return 3;
        // the run-time-width kernel is lean in registers: five per CU (27-point 256^3 through it: 0.731 ms; three 0.750, eight
        // 0.781) -- or as many as the staged windows leave room for in the CU's 160 KiB of LDS
        const int fit = (int)((150 * 1024) / (sizeof(double) * (size_t)m->stage_total + 2200));
        return fit >= 5 ? 5 : (fit < 1 ? 1 : fit);
    }
    return m != nullptr && m->dict_state == 1 && m->dict_mode == 1 && m->K <= 8 ? 5 : 8;
}

static int ell_grid_rows(int64_t rows, int per_cu)
{
    // tiles of 512 rows (256 threads x 2 rows); grid a multiple of 8 so tile%8 -> XCD group is stable
    const int64_t tiles = (rows + 2 * CGH_BLOCK - 1) / (2 * CGH_BLOCK);
    if (tiles <= CGH_ELL_SMALL_TILES) return (int)(tiles < 1 ? 1 : tiles);  // small: one workgroup per tile, in order
    int           g = cgh_stream_grid(tiles, per_cu);
    if (g >= 8) g -= g % 8;
    return g;
}

int cgh_ell_grid(const cgh_matrix *m) { return ell_grid_rows(m->n, ell_wgs_per_cu(m)); }

// The staged kernel serves: a dictionary-coded matrix with a plan, byte codes (narrow rows, K <= 8) or tile-blocked words
// (K = 13 / 26), one row range.  Returns false if this launch is not one of those (the caller goes on with spmv_ell_r2).
template <int DOT>
static bool launch_staged(const cgh_matrix *m, int var, int grid, hipStream_t s, const double *x, double *y, double *partials,
                          const int *done, int rev, int64_t rb, int64_t re, bool interior_rows)
{
    const int  kind = ell_staged_kind(m, interior_rows);
    const bool narrow = kind == 1, wide = kind == 2;
    if (!kind) return false;
    cgh_stage_plan sp;
    sp.nwin = m->stage_nwin;
    sp.total = m->stage_total;
    for (int w = 0; w < CGH_STAGE_WINDOWS; ++w) {
        sp.first[w] = m->stage_first[w];
        sp.len[w] = m->stage_len[w];
        sp.base[w] = m->stage_base[w];
    }
    const size_t  shm = sizeof(double) * (size_t)sp.total;
    const double *vals = wide ? m->vals_blk : m->vals;
#define CGH_STAGED_LAUNCH(KK, V, MODE, TILES)                                                                                    \
    hipLaunchKernelGGL((spmv_ell_staged<KK, DOT, V, MODE, TILES>), dim3(grid), dim3(CGH_BLOCK), shm, s, m->n, m->K, (const double *)m->diag, \
                       vals, (const void *)m->cols8, (const int *)m->off_tab + 256, x, y, partials, done, rev, rb, re, sp);
#define CGH_STAGED_CASE(KK, V, MODE)                                                                                             \
    case KK:                                                                                                                     \
        CGH_STAGED_LAUNCH(KK, V, MODE, 1)                                                                                        \
        break;
#define CGH_STAGED_CASE_W(KK, V)                                                                                                 \
    case KK:                                                                                                                     \
        if (m->stage_tiles == 2) {                                                                                               \
            CGH_STAGED_LAUNCH(KK, V, 3, 2)                                                                                       \
        } else {                                                                                                                 \
            CGH_STAGED_LAUNCH(KK, V, 3, 1)                                                                                       \
        }                                                                                                                        \
        break;
#define CGH_STAGED_VAR(V)                                                                                                        \
    if (narrow) {                                                                                                                \
        switch (g_cfg.ell_generic ? 0 : m->K) {                                                                                  \
            CGH_STAGED_CASE(1, V, 1) CGH_STAGED_CASE(2, V, 1) CGH_STAGED_CASE(3, V, 1) CGH_STAGED_CASE(4, V, 1)                  \
            CGH_STAGED_CASE(5, V, 1) CGH_STAGED_CASE(6, V, 1) CGH_STAGED_CASE(7, V, 1) CGH_STAGED_CASE(8, V, 1)                  \
        default:                                                                                                                 \
            CGH_STAGED_CASE(0, V, 1)                                                                                             \
        }                                                                                                                        \
    } else {                                                                                                                     \
        switch (g_cfg.ell_generic ? 0 : m->K) {                                                                                  \
            CGH_STAGED_CASE_W(13, V) CGH_STAGED_CASE_W(26, V)                                                                    \
        default: /* any other width: run-time K */                                                                               \
            CGH_STAGED_CASE_W(0, V)                                                                                              \
        }                                                                                                                        \
    }
    if (var == 1) {
        CGH_STAGED_VAR(1)
    } else if (var == 3) {
        CGH_STAGED_VAR(3)
    } else {
        CGH_STAGED_VAR(VAR_DEFAULT)
    }
#undef CGH_STAGED_VAR
#undef CGH_STAGED_CASE
#undef CGH_STAGED_CASE_W
#undef CGH_STAGED_LAUNCH
    return true;
}

// y[rb:re) = (A x)[rb:re).  The slot stride is m->n; x may be longer than n (multi-GPU: halo entries follow the owned ones).
int cgh_launch_spmv_ell_range(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                              const int *done, hipStream_t s, int64_t rb, int64_t re, const cgh_spmv_opts *opts)
{
    return cgh_launch_spmv_ell_range2(m, x, y, dot_partials, grid_out, done, s, rb, re, 0, 0, opts);
}

bool cgh_spmv_ell_halo_capable(const cgh_matrix *m, const double *x, double *y, int64_t rb, int64_t re, int64_t rb2, int64_t re2)
{
    return ell_vector_path(m, x, y) && rb % 2 == 0 && re % 2 == 0 && rb2 % 2 == 0 && re2 % 2 == 0;
}

// Two row ranges in one launch (second may be empty).
int cgh_launch_spmv_ell_range2(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                               const int *done, hipStream_t s, int64_t rb, int64_t re, int64_t rb2, int64_t re2,
                               const cgh_spmv_opts *opts)
{
    const int  rev = opts ? opts->reverse : 0;
    double    *cell_sums = opts ? opts->cell_sums : nullptr;
    if (opts && opts->cell_sums_written) *opts->cell_sums_written = false;
    if (re < rb) re = rb;
    if (re2 < rb2) re2 = rb2;
    const int64_t rows = (re - rb) + (re2 - rb2);
    if (grid_out) *grid_out = 0;
    if (rows <= 0) return CGH_OK;
    const bool vector_path = ell_vector_path(m, x, y) && rb % 2 == 0 && re % 2 == 0 && rb2 % 2 == 0 && re2 % 2 == 0;
    // the column dictionary is decided before the grid is: the grid depends on which kernel runs
    if (vector_path && __atomic_load_n(&m->dict_state, __ATOMIC_ACQUIRE) == 0 && !(opts && opts->no_alloc)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (cap == hipStreamCaptureStatusNone) {
            // the one place a launch path changes a (logically const) matrix: once, under a lock; the state word is
            // published last, so a thread that does not take the lock sees either "not built" or a complete dictionary
            std::lock_guard<std::mutex> guard(g_dict_mutex);
            if (m->dict_state == 0) ell_dict_try_build(m, s);
        }
    }
    const bool interior = opts && opts->interior_rows && re2 <= rb2;
    const int  grid = ell_grid_rows(rows, vector_path ? ell_wgs_per_cu(m, interior) : 8);
    if (grid_out) *grid_out = grid;
    if (!(ell_vector_path(m, x, y) && rb % 2 == 0 && re % 2 == 0 && rb2 % 2 == 0 && re2 % 2 == 0) && re2 > rb2) {
        // scalar path: one launch per range
        int g1 = 0, g2 = 0;
        int rc = cgh_launch_spmv_ell_range2(m, x, y, dot_partials, &g1, done, s, rb, re, 0, 0, opts);
        if (rc) return rc;
        rc = cgh_launch_spmv_ell_range2(m, x, y, dot_partials ? dot_partials + g1 : nullptr, &g2, done, s, rb2, re2, 0, 0, opts);
        if (grid_out) *grid_out = g1 + g2;
        return rc;
    }
    if (ell_vector_path(m, x, y) && rb % 2 == 0 && re % 2 == 0 && rb2 % 2 == 0 && re2 % 2 == 0) {
        // strict mode's per-cell sums of x.y instead of per-workgroup partials
        if (cell_sums != nullptr && dot_partials == nullptr && rb == 0 && re == m->n && re2 <= rb2 && m->n % 128 == 0) {
            if (launch_staged<2>(m, VAR_DEFAULT, grid, s, x, y, cell_sums, done, rev, rb, re, interior)) {
            } else if (m->dict_state == 1 && m->dict_mode == 3)
                launch_r2<2, VAR_DEFAULT, 3>(m->K, grid, s, m->n, m->diag, m->vals_blk, m->cols8, m->off_tab, x, y, cell_sums, done,
                                             rb, re, rb2, re2, rev);
            else if (m->dict_state == 1 && m->dict_mode == 2)
                launch_r2<2, VAR_DEFAULT, 2>(m->K, grid, s, m->n, m->diag, m->vals, m->cols8, m->off_tab, x, y, cell_sums, done,
                                             rb, re, rb2, re2, rev);
            else if (m->dict_state == 1)
                launch_r2<2, VAR_DEFAULT, 1>(m->K, grid, s, m->n, m->diag, m->vals, m->cols8, m->off_tab, x, y, cell_sums, done,
                                             rb, re, rb2, re2, rev);
            else
                launch_r2<2, VAR_DEFAULT, 0>(m->K, grid, s, m->n, m->diag, m->vals, m->cols, nullptr, x, y, cell_sums, done, rb,
                                             re, rb2, re2, rev);
            if (opts->cell_sums_written) *opts->cell_sums_written = true;
            CGH_HIP(hipGetLastError());
            return CGH_OK;
        }
        if (m->dict_state == 1) {
            // Variant: 11 = non-temporal matrix loads and y store (the matrix streams through once, x stays in L2); 1 = plain
            // loads and stores, for working sets the 256 MiB Infinity Cache can hold from one kernel to the next (there the
            // hints cost 4-8 % of an iteration: tools/quick_mall.py); CGH_ELL_VARIANT=1|3|11 overrides (3: NT loads only)
            const int forced = g_cfg.ell_variant;
            const int dvar = (forced == 1 || forced == 3 || forced == 11) ? forced : (cgh_fits_llc(m, 1.0) ? 1 : VAR_DEFAULT);
            if (re2 <= rb2 && (dot_partials ? launch_staged<1>(m, dvar, grid, s, x, y, dot_partials, done, rev, rb, re, interior)
                                            : launch_staged<0>(m, dvar, grid, s, x, y, nullptr, done, rev, rb, re, interior))) {
                CGH_HIP(hipGetLastError());
                return CGH_OK;
            }
#define CGH_ELL_DICT_LAUNCH(V, MODE)                                                                                       \
    if (dot_partials)                                                                                                      \
        launch_r2<true, V, MODE>(m->K, grid, s, m->n, m->diag, MODE == 3 ? m->vals_blk : m->vals, m->cols8, m->off_tab, x, y,  \
                                 dot_partials, done, rb, re, rb2, re2, rev);                                               \
    else                                                                                                                   \
        launch_r2<false, V, MODE>(m->K, grid, s, m->n, m->diag, MODE == 3 ? m->vals_blk : m->vals, m->cols8, m->off_tab, x, y, \
                                  nullptr, done, rb, re, rb2, re2, rev);
            if (m->dict_mode == 3) {
                if (dvar == 3) {
                    CGH_ELL_DICT_LAUNCH(3, 3)
                } else if (dvar == 1) {
                    CGH_ELL_DICT_LAUNCH(1, 3)
                } else {
                    CGH_ELL_DICT_LAUNCH(VAR_DEFAULT, 3)
                }
            } else if (m->dict_mode == 2) {
                if (dvar == 3) {
                    CGH_ELL_DICT_LAUNCH(3, 2)
                } else if (dvar == 1) {
                    CGH_ELL_DICT_LAUNCH(1, 2)
                } else {
                    CGH_ELL_DICT_LAUNCH(VAR_DEFAULT, 2)
                }
            } else if (dvar == 3) {
                CGH_ELL_DICT_LAUNCH(3, 1)
            } else if (dvar == 1) {
                CGH_ELL_DICT_LAUNCH(1, 1)
            } else {
                CGH_ELL_DICT_LAUNCH(VAR_DEFAULT, 1)
            }
#undef CGH_ELL_DICT_LAUNCH
            CGH_HIP(hipGetLastError());
            return CGH_OK;
        }
        const int v_cfg = g_cfg.ell_variant;
        const int var_env = (v_cfg == 0 || v_cfg == 1 || v_cfg == 2 || v_cfg == 3 || v_cfg == 7 || v_cfg == 11) ? v_cfg : -1;
        const int var = var_env >= 0 ? var_env : (cgh_fits_llc(m, 1.0) ? 1 : VAR_DEFAULT);
#define CGH_ELL_VAR(V)                                                                                                  \
    case V:                                                                                                             \
        if (dot_partials)                                                                                               \
            launch_r2<true, V>(m->K, grid, s, m->n, m->diag, m->vals, m->cols, nullptr, x, y, dot_partials, done, rb, re, rb2, re2, rev); \
        else                                                                                                            \
            launch_r2<false, V>(m->K, grid, s, m->n, m->diag, m->vals, m->cols, nullptr, x, y, nullptr, done, rb, re, rb2, re2, rev); \
        break;
        switch (var) {
            CGH_ELL_VAR(0)
            CGH_ELL_VAR(1)
            CGH_ELL_VAR(2)
            CGH_ELL_VAR(3)
            CGH_ELL_VAR(7)
            CGH_ELL_VAR(11)
        }
#undef CGH_ELL_VAR
    } else {
        if (dot_partials)
            hipLaunchKernelGGL((spmv_ell_r1<true>), dim3(grid), dim3(CGH_BLOCK), 0, s, m->n, m->K, m->diag, m->vals,
                               m->cols, x, y, dot_partials, done, rb, re);
        else
            hipLaunchKernelGGL((spmv_ell_r1<false>), dim3(grid), dim3(CGH_BLOCK), 0, s, m->n, m->K, m->diag, m->vals,
                               m->cols, x, y, (double *)nullptr, done, rb, re);
    }
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

// Rows [rb,re) + [rb2,re2) of a multi-GPU block whose halo entries (column ids >= hl->nl) are read from the deposit
// buffer instead of x's tail; partials of x.y are always produced.  Returns CGH_EINVAL if this matrix cannot take
// the vector path (the caller then unpacks the deposit into x's tail and uses cgh_launch_spmv_ell_range2).
int cgh_launch_spmv_ell_halo(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                             const int *done, hipStream_t s, int64_t rb, int64_t re, int64_t rb2, int64_t re2,
                             const cgh_ll_halo *hl)
{
    if (re < rb) re = rb;
    if (re2 < rb2) re2 = rb2;
    const int64_t rows = (re - rb) + (re2 - rb2);
    if (grid_out) *grid_out = 0;
    if (!cgh_spmv_ell_halo_capable(m, x, y, rb, re, rb2, re2) || !dot_partials) return CGH_EINVAL;
    if (rows <= 0) return CGH_OK;
    if (__atomic_load_n(&m->dict_state, __ATOMIC_ACQUIRE) == 0) {
        std::lock_guard<std::mutex> guard(g_dict_mutex);
        if (m->dict_state == 0) ell_dict_try_build(m, s);
    }
    const int grid = ell_grid_rows(rows, ell_wgs_per_cu(m));
    if (grid_out) *grid_out = grid;
#define CGH_ELL_HALO_LAUNCH(V)                                                                                                 \
    if (m->dict_state == 1 && m->dict_mode == 3)                                                                               \
        launch_r2<1, V, 3, 1>(m->K, grid, s, m->n, m->diag, m->vals_blk, m->cols8, m->off_tab, x, y, dot_partials, done, rb, re, rb2, re2, \
                              0, *hl);                                                                                            \
    else if (m->dict_state == 1 && m->dict_mode == 2)                                                                               \
        launch_r2<1, V, 2, 1>(m->K, grid, s, m->n, m->diag, m->vals, m->cols8, m->off_tab, x, y, dot_partials, done, rb, re, rb2, re2, \
                              0, *hl);                                                                                            \
    else if (m->dict_state == 1)                                                                                               \
        launch_r2<1, V, 1, 1>(m->K, grid, s, m->n, m->diag, m->vals, m->cols8, m->off_tab, x, y, dot_partials, done, rb, re, rb2, re2, \
                              0, *hl);                                                                                            \
    else                                                                                                                       \
        launch_r2<1, V, 0, 1>(m->K, grid, s, m->n, m->diag, m->vals, m->cols, nullptr, x, y, dot_partials, done, rb, re, rb2, re2, 0, *hl);
    if (cgh_fits_llc(m, 1.0)) {
        CGH_ELL_HALO_LAUNCH(1)
    } else {
        CGH_ELL_HALO_LAUNCH(VAR_DEFAULT)
    }
#undef CGH_ELL_HALO_LAUNCH
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

int cgh_launch_spmv_ell(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                        const int *done, hipStream_t s, const cgh_spmv_opts *opts)
{
    if (opts && opts->cell_sums_written) *opts->cell_sums_written = false;
    if (m->n == 0) {
        if (grid_out) *grid_out = 0;
        return CGH_OK;
    }
    return cgh_launch_spmv_ell_range(m, x, y, dot_partials, grid_out, done, s, 0, m->n, opts);
}
