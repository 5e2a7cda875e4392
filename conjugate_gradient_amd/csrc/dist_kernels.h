// dist_kernels.h -- the two vector kernels of the multi-GPU iteration with their communication folded in, and the
// stand-alone halo kernels for the cases the fused path does not cover.  New module, no reference counterpart
// (SURVEY.md 8e); the arithmetic on r, x, p is that of cg_update_r / cg_update_xp (cg_kernels.h), i.e. reference
// src/cg.c:463-489.
//
//   dist_update_r    prologue: p.Ap of this rank from the SpMV's partials, all-reduced over the ranks through the
//                    mailboxes (link_allreduce); then r -= alpha Ap and the partials of r.r, |r|_1
//   dist_update_xp   prologue: r.r and |r|_1 all-reduced the same way; x += alpha p, convergence test, p = r + beta p;
//                    rows of p inside a neighbour's send range also go, tagged, into that neighbour's deposit buffer
//                    (the halo exchange of the NEXT iteration costs no launch and starts as early as it can)
//   halo_push_ll     the same push as a kernel of its own: the first exchange of a solve (p = r0 comes from cg_init_rp)
//                    and send lists that are not contiguous
//   halo_pull_ll     deposit slot -> halo tail of p, for SpMV paths that cannot read the deposit themselves
// Per iteration a rank launches the SpMV (rows with halo columns last; they read the deposit), dist_update_r and
// dist_update_xp -- three kernels on one stream, as on a single GPU; no events, no library call.
#pragma once
#include "cg_kernels.h"
#include "halo_ll.h"

namespace {

constexpr int DIST_PRE = 2;  // chunks per workgroup whose loads precede the prologue

template <int VV>
__global__ __launch_bounds__(CGH_BLOCK) void dist_update_r(int64_t n, double *__restrict__ r, const double *__restrict__ Ap,
                                                          const double *__restrict__ pAp_partials, int pAp_count,
                                                          double *__restrict__ out_partials, double *hist, int iter,
                                                          const int *__restrict__ done, cgh_link L)
{
    __shared__ double lds[4];
    __shared__ double ex[2 * CGH_LINK_MAX_RANKS];
    const int64_t pairs = n >> 1;
    const int64_t chunks = (pairs + CGH_BLOCK - 1) / CGH_BLOCK;
    // The first DIST_PRE chunks' loads are issued BEFORE the prologue: at 2M rows per GPU a workgroup owns two chunks,
    // and the HBM latency then overlaps the partial sums and the mailbox round trip instead of following them.
    double2 rv[DIST_PRE], av[DIST_PRE];
#pragma unroll
    for (int j = 0; j < DIST_PRE; ++j) {
        const int64_t k = ((int64_t)blockIdx.x + (int64_t)j * gridDim.x) * CGH_BLOCK + threadIdx.x;
        if (k < pairs) {
            rv[j] = ld2<VV>(r, k);
            av[j] = ld2<VV>(Ap, k);
        }
    }
    if (*done != 0) return;  // frozen on every rank alike (the flag derives from all-reduced scalars): nobody exchanges
    double mine[1] = {block_sum_partials(pAp_partials, pAp_count, 1, lds)}, pAp[1];
    link_allreduce<1>(L, mine, pAp, ex);
    const double alpha = hist[(size_t)(iter - 1) * HIST + 1] / pAp[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) hist[(size_t)iter * HIST + 2] = alpha;
    const double nalpha = -alpha;
    double       rr = 0.0, l1 = 0.0;
#pragma unroll
    for (int j = 0; j < DIST_PRE; ++j) {
        const int64_t k = ((int64_t)blockIdx.x + (int64_t)j * gridDim.x) * CGH_BLOCK + threadIdx.x;
        if (k < pairs) {
            const double2 rn = make_double2(madd_rn(rv[j].x, nalpha, av[j].x), madd_rn(rv[j].y, nalpha, av[j].y));  // vec.c:88-94, cg.c:469
            st2<false>(r, k, rn.x, rn.y);
            rr += rn.x * rn.x + rn.y * rn.y;
            l1 += fabs(rn.x) + fabs(rn.y);
        }
    }
    for (int64_t cc = (int64_t)blockIdx.x + (int64_t)DIST_PRE * gridDim.x; cc < chunks; cc += gridDim.x) {
        const int64_t k = cc * CGH_BLOCK + threadIdx.x;
        if (k >= pairs) continue;
        const double2 r2 = ld2<VV>(r, k), a2 = ld2<VV>(Ap, k);
        const double2 rn = make_double2(madd_rn(r2.x, nalpha, a2.x), madd_rn(r2.y, nalpha, a2.y));
        st2<false>(r, k, rn.x, rn.y);
        rr += rn.x * rn.x + rn.y * rn.y;
        l1 += fabs(rn.x) + fabs(rn.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const double rn = madd_rn(r[n - 1], nalpha, Ap[n - 1]);
        r[n - 1] = rn;
        rr += rn * rn;
        l1 += fabs(rn);
    }
    const double trr = block_sum_256(rr, lds);
    const double tl1 = block_sum_256(l1, lds);
    if (threadIdx.x == 0) {
        out_partials[2 * blockIdx.x] = trr;
        out_partials[2 * blockIdx.x + 1] = tl1;
    }
}

// One round of the chained order-exact sums (dist.hip, strict_chain): two doubles all-reduced over the ranks through the
// mailboxes.  Exactly one rank contributes non-zero values, so the sum IS that rank's pair, bit for bit, on every rank.
// The tag counts (solve, iteration, round) like every other exchange of the loop, and a frozen solve exchanges nothing
// (the flag derives from the same totals on every rank), so ranks may queue different numbers of iterations after
// convergence without losing step.
// The mailbox slot of a round comes from a device-side count of the rounds actually EXECUTED (the same on every rank:
// all ranks execute the same rounds), so two rounds that follow each other never share a slot -- also across the end of
// a solve, where the round numbers jump.  (A rank may send round X+1 while a peer still reads round X, never more.)
__global__ __launch_bounds__(CGH_BLOCK) void chain_round(cgh_link L, const double *__restrict__ mine, double *__restrict__ run,
                                                        const int *__restrict__ done, int check_done, unsigned epoch, unsigned idx)
{
    __shared__ double ex[2 * CGH_LINK_MAX_RANKS];
    if (check_done && *done != 0) return;
    unsigned long long *executed = L.rounds;  // (one count per process: the solves of all objects share the mailbox)
    const unsigned      seq = (unsigned)*executed;
    L.region = 2;
    L.tag = cgh_ll_tag(epoch, (idx << 2) | (seq & 3u));
    double m[2] = {mine[0], mine[1]}, o[2];
    link_allreduce<2>(L, m, o, ex);  // (its barriers separate the read of the count above from the store below)
    if (threadIdx.x == 0) {
        run[0] = o[0];
        run[1] = o[1];
        *executed = seq + 1ull;
    }
}

__device__ __forceinline__ void push_row(const cgh_pushes &P, int64_t row, double v)
{
    for (int j = 0; j < P.n; ++j) {
        const int64_t k = row - P.start[j];
        if (k >= 0 && k < P.cnt[j]) ll_store(P.dst[j] + k, P.dst[j] + P.plane[j] + k, v, P.tag);
    }
}

// rows row0, row0+1 of the new p
__device__ __forceinline__ void push_pair(const cgh_pushes &P, int64_t row0, double v0, double v1)
{
    for (int j = 0; j < P.n; ++j) {
        const int64_t k = row0 - P.start[j];
        if (k < -1 || k >= P.cnt[j]) continue;
        uint64_t *lo = P.dst[j] + k, *hi = lo + P.plane[j];
        if (k >= 0 && k + 1 < P.cnt[j] && ((reinterpret_cast<uintptr_t>(lo) | reinterpret_cast<uintptr_t>(hi)) & 15u) == 0) {
            ll_store_pair(P.dst[j], k, P.plane[j] + k, v0, v1, P.tag);
        } else {
            if (k >= 0) ll_store(lo, hi, v0, P.tag);
            if (k + 1 < P.cnt[j]) ll_store(lo + 1, hi + 1, v1, P.tag);
        }
    }
}

template <int VV>
__global__ __launch_bounds__(CGH_BLOCK) void dist_update_xp(int64_t n, double *__restrict__ x, const double *__restrict__ r,
                                                           double *__restrict__ p, const double *__restrict__ partials, int count,
                                                           double *hist, int iter, double tol, int *done, cgh_link L, cgh_pushes P)
{
    __shared__ double lds[4];
    __shared__ double ex[2 * CGH_LINK_MAX_RANKS];
    const int64_t pairs = n >> 1;
    const int64_t chunks = (pairs + CGH_BLOCK - 1) / CGH_BLOCK;
    // sweep position cc -> chunk: rotated so that the chunks at P.rot come first (see cgh_pushes)
    auto chunk_of = [&](int64_t cc) {
        const int64_t c = cc + P.rot;
        return c >= chunks ? c - chunks : c;
    };
    double2       xv[DIST_PRE], rv[DIST_PRE], pv[DIST_PRE];  // loads before the prologue, as in dist_update_r
#pragma unroll
    for (int j = 0; j < DIST_PRE; ++j) {
        const int64_t cc = (int64_t)blockIdx.x + (int64_t)j * gridDim.x;
        const int64_t k = chunk_of(cc) * CGH_BLOCK + threadIdx.x;
        if (cc < chunks && k < pairs) {
            xv[j] = ld2<VV>(x, k);
            rv[j] = ld2<VV>(r, k);
            pv[j] = ld2<VV>(p, k);
        }
    }
    // The flag may name THIS iteration: workgroup 0 of this very launch publishes the verdict (below) while later
    // workgroups are still starting -- they go on and reach the same verdict (the peers' messages stay in the mailbox).
    const int frozen_at = *done;
    if (frozen_at != 0 && frozen_at != iter) return;
    double mine[2], tot[2];
    mine[0] = block_sum_partials(partials, count, 2, lds);
    mine[1] = block_sum_partials(partials + 1, count, 2, lds);
    link_allreduce<2>(L, mine, tot, ex);
    const double alpha = hist[(size_t)iter * HIST + 2];
    const double res = tot[1] / 2.0;  // norm_factor = 2.0, cg.c:427,475
    const bool   conv = res < tol;
    const double beta = tot[0] / hist[(size_t)(iter - 1) * HIST + 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hist[(size_t)iter * HIST + 0] = res;
        hist[(size_t)iter * HIST + 1] = tot[0];
        hist[(size_t)iter * HIST + 3] = conv ? NAN : beta;
        done[2] = iter;
    }
    if (conv) {
        // the reference updates x, then returns before touching p (cg.c:467-485); every workgroup (on every rank) reaches
        // the same verdict from the same scalars, workgroup 0 publishes it for the launches behind this one
        for (int64_t cc = blockIdx.x; cc < chunks; cc += gridDim.x) {
            const int64_t k = cc * CGH_BLOCK + threadIdx.x;
            if (k >= pairs) continue;
            const double2 x2 = ld2<VV>(x, k), p2 = ld2<VV>(p, k);
            st2<(VV & 2) != 0>(x, k, madd_rn(x2.x, alpha, p2.x), madd_rn(x2.y, alpha, p2.y));
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[n - 1] = madd_rn(x[n - 1], alpha, p[n - 1]);
        if (blockIdx.x == 0 && threadIdx.x == 0) *done = iter;
        return;
    }
#pragma unroll
    for (int j = 0; j < DIST_PRE; ++j) {
        const int64_t cc = (int64_t)blockIdx.x + (int64_t)j * gridDim.x;
        const int64_t k = chunk_of(cc) * CGH_BLOCK + threadIdx.x;
        if (cc < chunks && k < pairs) {
            // reference vec_axpy: out = a + alpha*b (vec.c:88-94): x += alpha p (cg.c:467), p = r + beta p (cg.c:489)
            st2<(VV & 2) != 0>(x, k, madd_rn(xv[j].x, alpha, pv[j].x), madd_rn(xv[j].y, alpha, pv[j].y));
            const double p0 = madd_rn(rv[j].x, beta, pv[j].x), p1 = madd_rn(rv[j].y, beta, pv[j].y);
            st2<false>(p, k, p0, p1);
            if (P.n > 0) push_pair(P, 2 * k, p0, p1);
        }
    }
    for (int64_t cc = (int64_t)blockIdx.x + (int64_t)DIST_PRE * gridDim.x; cc < chunks; cc += gridDim.x) {
        const int64_t k = chunk_of(cc) * CGH_BLOCK + threadIdx.x;
        if (k >= pairs) continue;
        const double2 x2 = ld2<VV>(x, k), r2 = ld2<VV>(r, k), p2 = ld2<VV>(p, k);
        st2<(VV & 2) != 0>(x, k, madd_rn(x2.x, alpha, p2.x), madd_rn(x2.y, alpha, p2.y));
        const double p0 = madd_rn(r2.x, beta, p2.x), p1 = madd_rn(r2.y, beta, p2.y);
        st2<false>(p, k, p0, p1);
        if (P.n > 0) push_pair(P, 2 * k, p0, p1);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        x[n - 1] = madd_rn(x[n - 1], alpha, p[n - 1]);
        const double pn = madd_rn(r[n - 1], beta, p[n - 1]);
        p[n - 1] = pn;
        push_row(P, n - 1, pn);
    }
}

// one neighbour's transfer as the stand-alone halo kernels see it
struct halo_xfer {
    uint64_t  *dst;      // the neighbour's deposit (slot 0, low words) at my region, as mapped here
    const int *idx;      // rows to send, NULL when they are contiguous from `start`
    int64_t    start, cnt;
    int64_t    plane;    // words from a slot's low words to its high words (= the neighbour's halo size); a slot is 2*plane
};

// grid: (workgroups per neighbour, neighbours)
__global__ __launch_bounds__(256) void halo_push_ll(const halo_xfer *__restrict__ xf, const double *__restrict__ p, unsigned tag,
                                                   int slot, const int *__restrict__ done)
{
    if (done != nullptr && *done != 0) return;
    const halo_xfer a = xf[blockIdx.y];
    uint64_t       *dst = a.dst + (size_t)slot * 2 * (size_t)a.plane;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < a.cnt; k += (int64_t)gridDim.x * 256)
        ll_store(dst + k, dst + a.plane + k, p[a.idx ? (int64_t)a.idx[k] : a.start + k], tag);
}

__global__ __launch_bounds__(256) void halo_pull_ll(cgh_ll_halo h, int64_t n_halo, double *__restrict__ halo_tail,
                                                   const int *__restrict__ done)
{
    if (done != nullptr && *done != 0) return;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n_halo; k += (int64_t)gridDim.x * 256) halo_tail[k] = ll_read(h, k);
}

} // namespace
