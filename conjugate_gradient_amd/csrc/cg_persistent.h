// cg_persistent.h -- the throughput-mode CG loop of a SMALL system as ONE resident launch per batch of iterations.
//
// Replaces, for systems of at most 256 tiles of 512 rows (131 072 rows), the three launches per iteration of cg.hip
// (reference loop: src/cg.c:459-496): at that size an iteration is three ~5 us latency-bound launches, none of them
// limited by bytes.  Here workgroup w owns rows [512 w, 512 w + 512) for a whole batch of iterations and keeps its rows
// of the matrix, x, r, p and Ap in REGISTERS; what crosses workgroups per iteration is
//   * the two reductions (p.Ap; r.r and |r|_1): every workgroup stores its partial as data-tagged words (two 64-bit
//     words per double: 32 bits of data under the 32-bit tag of the iteration, the LL encoding of halo_ll.h), every
//     workgroup reads ALL partials (one per lane) and adds them in the order block_sum_partials uses;
//   * the SpMV's gather of columns outside the workgroup's own tile.  p itself is NOT exchanged (p = r + beta p needs
//     beta, i.e. the second reduction, and the exchange would be a third dependent round trip per iteration): every
//     lane keeps the p of its off-tile columns in registers and advances them itself, p_c = r_c + beta p_c -- the
//     owner's arithmetic on the owner's operands, so the owner's bits -- from the NEW r of those columns, which the
//     owners store as tagged words together with their partials of r.r and which is gathered in the same wait as
//     that reduction.  Columns inside the tile come from an LDS copy of p.
// So an iteration has two dependent cross-workgroup round trips.  No grid barrier, no flag, no fence: a word is valid
// exactly when its tag matches.  One slot per buffer suffices: a workgroup can only overwrite a word after a
// reduction that needed every other workgroup to be past its last read of that word (pa: the r.r reduction lies
// between; pb and the image of r: the next p.Ap reduction does).  Every wait gives up after 2 s (error word, NaN),
// after which nobody waits again, so the grid always drains.  All workgroups derive the same scalars from the same
// partials, reach the same verdict and leave at the same iteration.  (Tags hold 13 bits of the iteration number: every
// word is rewritten every iteration, so a word never carries a tag 8192 iterations old.)
//
// Bits: identical to the three-launch path (same per-row arithmetic, same per-tile partials -- cg.hip gives every tile
// its own workgroup at these sizes --, same order of the partials' sum).
#pragma once
#include "cg_kernels.h"
#include "halo_ll.h"

namespace {

constexpr int       CGP_MAX_TILES = 256;              // one lane per partial in the all-gather
constexpr long long CGP_TIMEOUT_TICKS = 200000000ll;  // 2 s of the 100 MHz constant clock

typedef int i2p_t __attribute__((ext_vector_type(2)));

// Tag of iteration `it` (at most 8191: the history window holds 4096) of solve `epoch`: never 0, the value of a cleared
// buffer.  18 bits of epoch: the host clears the buffers whenever the epoch's higher bits have changed since it last did.
constexpr unsigned CGP_EPOCH_BITS = 18;
__host__ __device__ inline unsigned cgp_tag(unsigned epoch, unsigned it)
{
    return 0x80000000u | ((epoch & ((1u << CGP_EPOCH_BITS) - 1u)) << 13) | (it & 0x1fffu);
}

// Tagged entries travel as ONE 16-byte access each: buffer_load / buffer_store_dwordx4 with the sc1 bit (agent scope:
// stores written through, loads served below L1) through the raw-buffer builtins, so that the compiler keeps many in
// flight and waits where the data is used.  (volatile accesses are completed one by one, s_waitcnt vmcnt(0) after
// each: ~1 us per access; 8-byte agent atomics are one fabric transaction per word: 250 workgroups storing their 512
// entries of r that way stall each other.)  An entry is {low half, tag, high half, tag}; each aligned 8-byte half
// arrives whole, which is all the tags need (halo_ll.h).
typedef unsigned cgp_u4 __attribute__((ext_vector_type(4)));
constexpr int    CGP_SC1 = 16;  // cache-policy operand of the buffer builtins: sc1

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cgp_rsrc(uint64_t *base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ cgp_u4 cgp_entry(double v, unsigned tag)
{
    cgp_u4 e;
    e.x = (unsigned)__double2loint(v);
    e.y = tag;
    e.z = (unsigned)__double2hiint(v);
    e.w = tag;
    return e;
}
// entry k, k + 1 of the buffer <- v0, v1
__device__ __forceinline__ void cgp_store2(__amdgpu_buffer_rsrc_t rs, unsigned k, double v0, double v1, unsigned tag)
{
    __builtin_amdgcn_raw_buffer_store_b128(cgp_entry(v0, tag), rs, k * 16u, 0, CGP_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(cgp_entry(v1, tag), rs, k * 16u + 16u, 0, CGP_SC1);
}
__device__ __forceinline__ cgp_u4 cgp_load(__amdgpu_buffer_rsrc_t rs, unsigned k)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rs, k * 16u, 0, CGP_SC1);
}
__device__ __forceinline__ bool   cgp_valid(const cgp_u4 &e, unsigned tag) { return e.y == tag && e.w == tag; }
__device__ __forceinline__ double cgp_value(const cgp_u4 &e) { return __hiloint2double((int)e.z, (int)e.x); }
// entry k, already loaded once into e: wait until it carries `tag` (the clock is only consulted every 1024 looks:
// reading it costs about as much as a look)
__device__ __forceinline__ double cgp_await(__amdgpu_buffer_rsrc_t rs, unsigned k, cgp_u4 e, unsigned tag, int *err)
{
#ifdef CGP_NOWAIT
    return cgp_value(e);
#endif
    if (!cgp_valid(e, tag)) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return __builtin_nan("");
        long long t0 = 0;
        for (unsigned spin = 1;; ++spin) {
            e = cgp_load(rs, k);  // (no s_sleep between looks: 7.2 vs 6.9 us per iteration at 127k rows)
            if (cgp_valid(e, tag)) break;
            if ((spin & 1023u) == 0) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > CGP_TIMEOUT_TICKS) {
                    atomicExch(err, 1);
                    return __builtin_nan("");
                }
            }
        }
    }
    return cgp_value(e);
}

// wave_sum's operands in wave_sum's order (lane i adds lane i + 32, then + 16, 8, 4, 2, 1; result in lane 0) without
// the LDS crossbar: the two halves / the odd and even rows meet through v_permlane32_swap / v_permlane16_swap, the steps
// inside a row of 16 lanes are DPP row shifts.  ~50 ns instead of ~350 for six ds_bpermute round trips.
__device__ __forceinline__ double cgp_wave_sum(double v)
{
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto     a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto     b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);  // lane i < 32: v[i] + v[i + 32]
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto     a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto     b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);  // lane i < 16: v[i] + v[i + 16]
    }
    v += dpp_mov_f64<0x108, 0xf>(v);  // row_shl:8 (lanes without a source lane add +0.0; only lane 0's chain matters)
    v += dpp_mov_f64<0x104, 0xf>(v);
    v += dpp_mov_f64<0x102, 0xf>(v);
    v += dpp_mov_f64<0x101, 0xf>(v);
    return v;
}

// Sums of a and of b over the 256 threads, to every thread: block_sum_256's order (lane tree, then waves 0..3 left to
// right).  `lds` holds 2 x 8 doubles used in turn, so one barrier per call suffices (a slot is rewritten two calls
// later, behind the barrier of the call in between).
__device__ __forceinline__ void cgp_block_sum2(double &a, double &b, double *lds, int &flip)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a = cgp_wave_sum(a);
    b = cgp_wave_sum(b);
    double *L = lds + 8 * flip;
    flip ^= 1;
    if (lane == 0) {
        L[wave] = a;
        L[4 + wave] = b;
    }
    __syncthreads();
    a = ((L[0] + L[1]) + L[2]) + L[3];
    b = ((L[4] + L[5]) + L[6]) + L[7];
}
__device__ __forceinline__ double cgp_block_sum(double a, double *lds, int &flip)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a = cgp_wave_sum(a);
    double *L = lds + 8 * flip;
    flip ^= 1;
    if (lane == 0) L[wave] = a;
    __syncthreads();
    return ((L[0] + L[1]) + L[2]) + L[3];
}

template <int KT, int DICT>
__global__ __launch_bounds__(CGH_BLOCK) void cg_persistent(int64_t n, const double *__restrict__ diag, const double *__restrict__ vals,
                                                          const void *__restrict__ cols, const int *__restrict__ off_tab,
                                                          double *__restrict__ x, double *__restrict__ r, double *__restrict__ p,
                                                          uint64_t *rl, uint64_t *pa, uint64_t *pb, double *hist, int first_iter,
                                                          int count, double tol, int *done, unsigned epoch, long long *trace)
{
    __shared__ double                                 lds[16];
    int                                               flip = 0;
    __shared__ __attribute__((aligned(16))) double    ptile[2 * CGH_BLOCK];
    __shared__ int                                    tab[DICT ? 256 : 1];
    if (*done != 0) return;  // frozen by an earlier launch (every workgroup reads the same word)
    int          *err = done + 3;
    const int     tid = threadIdx.x, w = blockIdx.x, G = gridDim.x;
    const int64_t pairs = n >> 1;
    const int64_t pr = (int64_t)w * CGH_BLOCK + tid;
    const bool    live = pr < pairs;
    const int64_t i0 = pr << 1;
    const int     tile0 = w * 2 * CGH_BLOCK;
    if constexpr (DICT != 0) {
        tab[tid] = off_tab[tid];
        __syncthreads();
    }
    // this lane's two rows: matrix, x, r, p
    d2_t    v[KT];
    i2p_t   c[KT];
    double2 dg = make_double2(0.0, 0.0), xv = dg, rv = dg, pv = dg;
#pragma unroll
    for (int s = 0; s < KT; ++s) {
        v[s].x = v[s].y = 0.0;
        c[s].x = c[s].y = -1;
    }
    if (live) {
        dg = *reinterpret_cast<const double2 *>(diag + i0);
        xv = *reinterpret_cast<const double2 *>(x + i0);
        rv = *reinterpret_cast<const double2 *>(r + i0);
        pv = *reinterpret_cast<const double2 *>(p + i0);
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            v[s] = *reinterpret_cast<const d2_t *>(vals + (int64_t)s * n + i0);
            if constexpr (DICT != 0) {
                const unsigned short cc = reinterpret_cast<const unsigned short *>(cols)[((int64_t)s * n + i0) >> 1];
                const int            c0 = cc & 0xff, c1 = cc >> 8;
                c[s].x = c0 != 255 ? (int)i0 + tab[c0] : -1;
                c[s].y = c1 != 255 ? (int)i0 + 1 + tab[c1] : -1;
            } else {
                c[s] = *reinterpret_cast<const i2p_t *>(reinterpret_cast<const int *>(cols) + (int64_t)s * n + i0);
            }
        }
    }
    double   rr_prev = hist[(size_t)(first_iter - 1) * HIST + 1];
    int      conv_at = 0, last = first_iter - 1;
    // The image of r holds the even rows, then the odd rows (entry of row i at (i & 1) half + (i >> 1)): a lane's two rows
    // have neighbours of equal parity in a banded matrix, so each of its gathers is then one contiguous run of 16-byte
    // entries over the wave -- every 128-byte line is fetched once, not by two instructions of half its entries each.
    // Workgroup w's partial of p.Ap = entry w of pa, its partials of r.r and |r|_1 = entries 2 w, 2 w + 1 of pb.
    const int                    tiles = (int)((pairs + CGH_BLOCK - 1) / CGH_BLOCK);
    const unsigned               half = (unsigned)tiles * CGH_BLOCK;
    auto                         slot_of = [half](int row) { return ((unsigned)row & 1u) * half + ((unsigned)row >> 1); };
    const __amdgpu_buffer_rsrc_t RL = cgp_rsrc(rl, (size_t)tiles * 2 * CGH_BLOCK * 16), PA = cgp_rsrc(pa, (size_t)CGP_MAX_TILES * 32),
                                 PB = cgp_rsrc(pb, (size_t)CGP_MAX_TILES * 32);

    // p of this lane's columns: inside the tile from LDS (every iteration), outside it kept in registers from here on
    ptile[2 * tid] = pv.x;
    ptile[2 * tid + 1] = pv.y;
    double gp0[KT], gp1[KT];
#pragma unroll
    for (int s = 0; s < KT; ++s) {
        const bool o0 = c[s].x >= 0 && (unsigned)(c[s].x - tile0) >= 2u * CGH_BLOCK;
        const bool o1 = c[s].y >= 0 && (unsigned)(c[s].y - tile0) >= 2u * CGH_BLOCK;
        gp0[s] = o0 ? p[c[s].x] : 0.0;
        gp1[s] = o1 ? p[c[s].y] : 0.0;
    }
    __syncthreads();

    // trace (tools/quick_persistent.py): where thread 0 of each workgroup spends its time, in ticks of the 100 MHz clock
    long long tk[5] = {0, 0, 0, 0, 0}, tprev = trace ? wall_clock64() : 0;
#define CGP_STAMP(K)                          \
    if (trace) {                              \
        const long long tnow = wall_clock64(); \
        tk[K] += tnow - tprev;                \
        tprev = tnow;                         \
    }
    for (int it = first_iter; it < first_iter + count; ++it) {
        const unsigned tg = cgp_tag(epoch, (unsigned)it);
        // ---- Ap = A p (rows of this lane), the arithmetic of spmv_ell_r2
        double a0 = add_rn(0.0, mul_rn(dg.x, pv.x));
        double a1 = add_rn(0.0, mul_rn(dg.y, pv.y));
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            // (every lane reads LDS, at a wrapped index where its column is outside the tile: no branches)
            const int    l0 = c[s].x - tile0, l1 = c[s].y - tile0;
            const double q0 = ptile[l0 & (2 * CGH_BLOCK - 1)], q1 = ptile[l1 & (2 * CGH_BLOCK - 1)];
            const double g0 = (unsigned)l0 < 2u * CGH_BLOCK ? q0 : gp0[s];
            const double g1 = (unsigned)l1 < 2u * CGH_BLOCK ? q1 : gp1[s];
            const double t0 = madd_rn(a0, v[s].x, g0);
            const double t1 = madd_rn(a1, v[s].y, g1);
            a0 = c[s].x >= 0 ? t0 : a0;
            a1 = c[s].y >= 0 ? t1 : a1;
        }
        if (!live) a0 = a1 = 0.0;
        // ---- p.Ap: this tile's partial, then everybody's
        double local = 0.0;
        local += pv.x * a0 + pv.y * a1;
        const double tot = cgp_block_sum(local, lds, flip);
        if (tid == 0) __builtin_amdgcn_raw_buffer_store_b128(cgp_entry(tot, tg), PA, (unsigned)w * 16u, 0, CGP_SC1);
        CGP_STAMP(0)
        double part = 0.0;
        if (tid < G) part = cgp_await(PA, (unsigned)tid, cgp_load(PA, (unsigned)tid), tg, err);
        CGP_STAMP(1)
        double vp = 0.0;
        if (tid < G) vp += part;  // block_sum_partials with at most 256 partials: lane t holds 0.0 + partial t
        const double pAp = cgp_block_sum(vp, lds, flip);
        const double alpha = rr_prev / pAp;
        const double nalpha = -alpha;
        // ---- r -= alpha Ap; r.r and |r|_1
        double2 rn = make_double2(0.0, 0.0);
        if (live) rn = make_double2(madd_rn(rv.x, nalpha, a0), madd_rn(rv.y, nalpha, a1));
        rv = rn;
        double rr = 0.0, l1 = 0.0;
        rr += rn.x * rn.x + rn.y * rn.y;
        l1 += fabs(rn.x) + fabs(rn.y);
        // the new r of this tile for the lanes elsewhere that gather it, then the tile's partials
        if (live) {
            __builtin_amdgcn_raw_buffer_store_b128(cgp_entry(rn.x, tg), RL, slot_of((int)i0) * 16u, 0, CGP_SC1);
            __builtin_amdgcn_raw_buffer_store_b128(cgp_entry(rn.y, tg), RL, slot_of((int)i0 + 1) * 16u, 0, CGP_SC1);
        }
        double trr = rr, tl1 = l1;
        cgp_block_sum2(trr, tl1, lds, flip);
        if (tid == 0) cgp_store2(PB, 2u * w, trr, tl1, tg);
        CGP_STAMP(2)
        // one wait for both: everybody's partials and the new r of this lane's off-tile columns (all first looks go
        // out before any is examined; looking again at everything that missed TOGETHER, round after round, was tried and
        // is slower -- 8.6 vs 6.9 us per iteration at 127k rows: the conditional reloads serialise)
        double prr = 0.0, pl1 = 0.0;
        cgp_u4 none, f0, f1;
        none.x = none.y = none.z = none.w = 0u;
        f0 = f1 = none;
        if (tid < G) {
            f0 = cgp_load(PB, 2u * tid);
            f1 = cgp_load(PB, 2u * tid + 1u);
        }
        cgp_u4 e0[KT], e1[KT];
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            const bool o0 = c[s].x >= 0 && (unsigned)(c[s].x - tile0) >= 2u * CGH_BLOCK;
            const bool o1 = c[s].y >= 0 && (unsigned)(c[s].y - tile0) >= 2u * CGH_BLOCK;
            e0[s] = e1[s] = none;
            if (o0) e0[s] = cgp_load(RL, slot_of(c[s].x));
            if (o1) e1[s] = cgp_load(RL, slot_of(c[s].y));
        }
        if (tid < G) {
            prr = cgp_await(PB, 2u * tid, f0, tg, err);
            pl1 = cgp_await(PB, 2u * tid + 1u, f1, tg, err);
        }
        double gr0[KT], gr1[KT];
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            const bool o0 = c[s].x >= 0 && (unsigned)(c[s].x - tile0) >= 2u * CGH_BLOCK;
            const bool o1 = c[s].y >= 0 && (unsigned)(c[s].y - tile0) >= 2u * CGH_BLOCK;
            gr0[s] = o0 ? cgp_await(RL, slot_of(c[s].x), e0[s], tg, err) : 0.0;
            gr1[s] = o1 ? cgp_await(RL, slot_of(c[s].y), e1[s], tg, err) : 0.0;
        }
        CGP_STAMP(3)
        double rrn = 0.0, l1n = 0.0;
        if (tid < G) {
            rrn += prr;
            l1n += pl1;
        }
        cgp_block_sum2(rrn, l1n, lds, flip);
        const double res = l1n / 2.0;  // norm_factor = 2.0, cg.c:427,475
        const bool   conv = res < tol;
        const double beta = rrn / rr_prev;
        // ---- x += alpha p; verdict; p = r + beta p
        if (live) xv = make_double2(madd_rn(xv.x, alpha, pv.x), madd_rn(xv.y, alpha, pv.y));
        if (w == 0 && tid == 0) {
            hist[(size_t)it * HIST + 0] = res;
            hist[(size_t)it * HIST + 1] = rrn;
            hist[(size_t)it * HIST + 2] = alpha;
            hist[(size_t)it * HIST + 3] = conv ? NAN : beta;
        }
        last = it;
        if (conv) {
            conv_at = it;  // the reference updates x, then returns before touching p (cg.c:467-485)
            break;
        }
        if (live) pv = make_double2(madd_rn(rv.x, beta, pv.x), madd_rn(rv.y, beta, pv.y));
#pragma unroll
        for (int s = 0; s < KT; ++s) {  // the owners' p = r + beta p, on the owners' operands
            gp0[s] = madd_rn(gr0[s], beta, gp0[s]);
            gp1[s] = madd_rn(gr1[s], beta, gp1[s]);
        }
        rr_prev = rrn;
        // every lane of the workgroup has passed several barriers since its last read of ptile
        ptile[2 * tid] = pv.x;
        ptile[2 * tid + 1] = pv.y;
        __syncthreads();
        CGP_STAMP(4)
    }
#undef CGP_STAMP
    if (trace && tid == 0)
        for (int k = 0; k < 5; ++k) trace[(size_t)w * 8 + k] = tk[k];
    if (live) {
        *reinterpret_cast<double2 *>(x + i0) = xv;
        *reinterpret_cast<double2 *>(r + i0) = rv;
        *reinterpret_cast<double2 *>(p + i0) = pv;
    }
    if (w == 0 && tid == 0) {
        done[1] = last;
        done[2] = last;
        if (conv_at) done[0] = conv_at;
    }
}

} // namespace
