// spmv_pb.hip -- two-phase ("propagation blocking") SpMV: y = A x for matrices whose columns are scattered over a
// vector that fits no cache close to the CUs (BASELINE config 5: 8.4M rows, power-law row lengths, uniformly random
// columns; x = 67 MB against 4 MiB of L2 per XCD).
//
// No reference counterpart as a format: the reference's SpMV for such a matrix is mv_csr_symmetric
// (src/csr.c:190-229), a serial scatter.  Any one-pass GPU SpMV pays one random access per stored entry -- a gather of
// x[col] or an update of y[row] -- and even when those hit L2 (spmv_csrb.hip) the CU's L1-miss path delivers only about
// a quarter of a line per cycle: 0.33 ms of the 0.68 ms that kernel takes on config 5.  Here every random access is an
// LDS access instead, at the price of streaming the products through HBM once:
//
//   * rows (= columns: the matrices are structurally symmetric) are cut into BLOCKS of at most SPAN consecutive rows
//     holding about equal numbers of entries (cgh_pb_partition_host, boundaries on multiples of 256 rows);
//   * an entry is an 8-byte value, a 2-byte column inside its block and a 2-byte row inside its block -- 12 bytes, as
//     CSR.  The entries of one (row block b, column block s) are a SEGMENT.  Values and local columns are stored
//     sorted by (s, b) -- what one workgroup of phase 1 reads is ONE contiguous run --, local rows and the products
//     sorted by (b, s) -- what one workgroup of phase 2 reads is one contiguous run;
//   * phase 1, pb_expand, one workgroup per column block s: the block's part of x is staged in LDS (<= 128 KB), the
//     workgroup streams its run (segments b = 0..B-1), multiplies each value by its x from LDS and stores the
//     product at the entry's position of the product stream (B runs of one segment each, coalesced run by run);
//   * phase 2, pb_accumulate, one workgroup per row block b: the block's row sums live in LDS; the workgroup streams
//     its products (ONE contiguous run) with their 2-byte local rows, adds them with LDS atomics, then writes
//     y = diag*x + sum and the block's partial of x.y.
//
// Bytes per entry: 10 read + 8 written in phase 1, 10 read in phase 2 = 28 against the 12 algorithmic ones; no access
// to HBM or L2 is random.  Throughput layout only: a row's products are added in the order of the product stream
// (ORDERED, the default: the same bits on every run) or, cgh_config_pb(.., ordered = 0) / CGH_PB_ORDERED=0, in the order
// the LDS atomics retire (2 % faster, last bits differ from run to run) -- never the reference's order (results agree
// with the exact-order kernels to a few ulps of the row's magnitude).  Strict solves and exact_rows callers keep
// CGH_LAYOUT_CSR / ELL.
#include <algorithm>
#include <exception>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "cgh_internal.h"
#include "device_utils.h"

int cgh_exclusive_scan_i64(const int64_t *d_in, int64_t *d_out, int64_t count, int64_t *h_total, hipStream_t s);  // convert.hip

namespace {

constexpr int GROUP = 256;   // rows per partition unit: block boundaries are multiples of this
constexpr int PAD = 16;      // a segment holds a multiple of this many entries (128 bytes of values / products)
constexpr int MAXB = 2048;   // blocks: p.Ap partials of a launch (CGH_MAX_PARTIALS) and the LDS tables of pb_expand

struct dbuf {
    void *p = nullptr;
    explicit dbuf(size_t bytes) { p = cgh_alloc(bytes ? bytes : 16); }
    ~dbuf() { cgh_free(p); }
    dbuf(const dbuf &) = delete;
    dbuf &operator=(const dbuf &) = delete;
    template <typename T> T *as() { return (T *)p; }
};

inline int grid_for(int64_t work) { return cgh_stream_grid((work + 255) / 256, 8); }

// ------------------------------------------------------------------------------------------------ construction
__global__ __launch_bounds__(256) void pb_sample_row_ptr(int64_t n, int64_t G, const int64_t *__restrict__ row_ptr, int64_t *__restrict__ out)
{
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g <= G; g += (int64_t)gridDim.x * 256) {
        const int64_t r = g * GROUP < n ? g * GROUP : n;
        out[g] = row_ptr[r];
    }
}

// key of entry e of row r: block(r) * B + block(col); one wave per row keeps the loads coalesced
__global__ __launch_bounds__(256) void pb_entry_keys(int64_t n, int B, const int64_t *__restrict__ row_ptr, const int *__restrict__ cols,
                                                    const unsigned short *__restrict__ blk_of, unsigned *__restrict__ keys,
                                                    unsigned *__restrict__ idx, int *__restrict__ rowid)
{
    const int     lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t r = wave; r < n; r += nwaves) {
        const unsigned rb = blk_of[r / GROUP];
        const int64_t  e1 = row_ptr[r + 1];
        for (int64_t e = row_ptr[r] + lane; e < e1; e += 64) {
            keys[e] = rb * (unsigned)B + (unsigned)blk_of[cols[e] / GROUP];
            idx[e] = (unsigned)e;
            rowid[e] = (int)r;
        }
    }
}

// sorted position q of segment `key` = (b, s) -> position pstart[key] + (q - start[key]) of the product stream (local row
// there), and position cbase[s] + pre[s][b] + (q - start[key]) of column block s's own run (value and local column there)
__global__ __launch_bounds__(256) void pb_gather_sorted(int64_t nnz, int B, const unsigned *__restrict__ keys, const unsigned *__restrict__ idx,
                                                       const double *__restrict__ vals, const int *__restrict__ cols,
                                                       const int *__restrict__ rowid, const int *__restrict__ bnd,
                                                       const int64_t *__restrict__ start, const int64_t *__restrict__ pstart,
                                                       const int *__restrict__ pre, const int64_t *__restrict__ cbase,
                                                       double *__restrict__ svals, unsigned short *__restrict__ lcol,
                                                       unsigned short *__restrict__ lrow)
{
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * 256) {
        const unsigned e = idx[q], key = keys[q];
        const int      b = (int)(key / (unsigned)B), s = (int)(key % (unsigned)B);
        const int64_t  in_seg = q - start[key];
        const int64_t  at = pstart[key] + in_seg, at2 = cbase[s] + pre[(size_t)s * (B + 1) + b] + in_seg;
        svals[at2] = vals[e];
        lcol[at2] = (unsigned short)(cols[e] - bnd[s]);
        lrow[at] = (unsigned short)(rowid[e] - bnd[b]);
    }
}

__global__ __launch_bounds__(256) void pb_padded_lengths(int64_t nseg, const int64_t *__restrict__ start, int64_t *__restrict__ plen)
{
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nseg; t += (int64_t)gridDim.x * 256)
        plen[t] = (start[t + 1] - start[t] + (PAD - 1)) / PAD * PAD;
}

// start[t] = first sorted position whose key is >= t, for t = 0..nseg (start[nseg] = nnz)
__global__ __launch_bounds__(256) void pb_seg_starts(int64_t nnz, int64_t nseg, const unsigned *__restrict__ keys, int64_t *__restrict__ start)
{
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q <= nnz; q += (int64_t)gridDim.x * 256) {
        const int64_t lo = q == 0 ? 0 : (int64_t)keys[q - 1] + 1;
        const int64_t hi = q == nnz ? nseg : (int64_t)keys[q];
        for (int64_t t = lo; t <= hi; ++t) start[t] = q;
    }
}

// Per column block s, what pb_expand stages in LDS: pre[s][b] = entries of the segments (0..b-1, s) (flat position of
// segment (b, s) inside the block's work = inside its run of values and local columns), seg0[s][b] = where that segment
// starts in the product stream; bin_ptr[b] = start of row block b there; col_total[s] = length of the block's run.
__global__ __launch_bounds__(256) void pb_tables(int B, const int64_t *__restrict__ start, int *__restrict__ pre, unsigned *__restrict__ seg0,
                                                int64_t *__restrict__ bin_ptr, int64_t *__restrict__ col_total)
{
    for (int s = blockIdx.x * 256 + threadIdx.x; s < B; s += gridDim.x * 256) {
        int run = 0;
        for (int b = 0; b < B; ++b) {
            const int64_t a = start[(int64_t)b * B + s];
            pre[(size_t)s * (B + 1) + b] = run;
            seg0[(size_t)s * B + b] = (unsigned)a;
            run += (int)(start[(int64_t)b * B + s + 1] - a);
        }
        pre[(size_t)s * (B + 1) + B] = run;
        col_total[s] = run;
        bin_ptr[s] = start[(int64_t)s * B];
        if (s == B - 1) bin_ptr[B] = start[(int64_t)B * B];
    }
}

// ------------------------------------------------------------------------------------------------ the kernels
//
// Every segment is padded to a multiple of PAD = 16 entries (value 0, local row 0xffff): a segment's values and products
// start on a 128-byte line, so no line of the product stream is written by two workgroups, and a lane owns two adjacent
// entries (16-byte loads and stores).
//
// pb_expand.  The work of column block s is the concatenation of its B segments (flat positions 0..tot); it is dealt in
// contiguous pieces of 128-entry chunks to the workgroup's waves (and to `split` workgroups per block), so a wave walks
// a few neighbouring segments front to back.  A lane finds its segment by stepping a cursor through the LDS table from
// the wave's own (uniform) cursor; two pieces of UN chunks are in flight per lane.
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int SPAN, int UN = 4>
__global__ __launch_bounds__(SPAN / 16) void pb_expand(int B, int split, const int *__restrict__ bnd, const int *__restrict__ pre,
                                                      const unsigned *__restrict__ seg0, const int64_t *__restrict__ cbase,
                                                      const double *__restrict__ vals,
                                                      const unsigned short *__restrict__ lcol, const double *__restrict__ x,
                                                      double *__restrict__ prod, const int *__restrict__ done,
                                                      long long *__restrict__ trace)
{
    constexpr int T = SPAN / 16, NW = T / 64;
    __shared__ double   xs[SPAN];
    __shared__ int      pre_s[MAXB + 1];
    __shared__ unsigned st_s[MAXB];
    if (done != nullptr && *done != 0) return;
    const int s = B - 1 - (int)(blockIdx.x / split), part = (int)(blockIdx.x % split);  // later blocks (denser rows) first
    if (trace != nullptr && threadIdx.x == 0) trace[4 * blockIdx.x] = wall_clock64();  // tools/trace_pb.py
    const int r0 = bnd[s], cnt = bnd[s + 1] - r0;
    {
        // cnt <= SPAN = 16 T: the block's x in one round trip (a loop of load -> LDS store pays the memory latency 16 times
        // over while the CU's one workgroup does nothing else), the two tables behind it
        double t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = threadIdx.x + q * T;
            t[q] = i < cnt ? x[r0 + i] : 0.0;
        }
        for (int i = threadIdx.x; i <= B; i += T) pre_s[i] = pre[(size_t)s * (B + 1) + i];
        for (int i = threadIdx.x; i < B; i += T) st_s[i] = seg0[(size_t)s * B + i];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = threadIdx.x + q * T;
            if (i < cnt) xs[i] = t[q];
        }
    }
    __syncthreads();
    if (trace != nullptr && threadIdx.x == 0) trace[4 * blockIdx.x + 1] = wall_clock64();
    const int tot = pre_s[B];  // a multiple of PAD
    vals += cbase[s];          // the block's own run: flat position f of its work is entry f of the run
    lcol += cbase[s];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nchunks = (tot + 127) >> 7, units = split * NW, u = part * NW + wave;
    const int c0 = (int)((int64_t)nchunks * u / units), c1 = (int)((int64_t)nchunks * (u + 1) / units);
    int       f = c0 * 128;
    const int fend = c1 * 128 < tot ? c1 * 128 : tot;
    if (trace != nullptr && lane == 0) trace[4 * blockIdx.x + 3] = tot;
    if (f >= fend) return;
    int b;  // uniform cursor: pre_s[b] <= f (and f < tot = pre_s[B])
    {
        int lo = 0, hi = B;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre_s[mid] <= f) lo = mid;
            else hi = mid;
        }
        b = lo;
    }
    // A piece = UN chunks of 128 entries.  The loads of the next piece are issued before the products of this one are formed
    // and stored: two pieces' loads are in flight per lane (the positions depend on the tables in LDS only).
    struct piece {
        d2_t     v[UN];
        unsigned c[UN], k[UN];
        bool     act[UN];
    };
    auto fetch = [&](piece &pc, int at) {
        int bu = b;
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int fu = at + q * 128 + 2 * lane;  // even, segments are multiples of 16: the pair lies in one segment
            pc.act[q] = fu < fend;
            pc.v[q] = d2_t{0.0, 0.0};
            pc.c[q] = 0;
            pc.k[q] = 0;
            if (pc.act[q]) {
                pc.v[q] = __builtin_nontemporal_load((const d2_t *)(vals + fu));
                pc.c[q] = __builtin_nontemporal_load((const unsigned *)(lcol + fu));
                while (fu >= pre_s[bu + 1]) ++bu;  // fu < tot = pre_s[B]: stops at bu <= B - 1
                pc.k[q] = st_s[bu] + (unsigned)(fu - pre_s[bu]);  // where the products of this pair go
            }
        }
        b = __builtin_amdgcn_readfirstlane(bu);  // lane 0's cursor: its last position is below the next piece's first
    };
    auto emit = [&](const piece &pc) {
#pragma unroll
        for (int q = 0; q < UN; ++q)
            if (pc.act[q]) {
                d2_t pr;
                pr.x = mul_rn(pc.v[q].x, xs[pc.c[q] & 0xffffu]);
                pr.y = mul_rn(pc.v[q].y, xs[pc.c[q] >> 16]);
                // streamed: plain stores finish this kernel 7 % sooner and make the next one 25 % slower
                __builtin_nontemporal_store(pr, (d2_t *)(prod + pc.k[q]));
            }
    };
    piece pa, pb;
    fetch(pa, f);
    for (; f < fend; f += 2 * 128 * UN) {
        fetch(pb, f + 128 * UN);  // (past fend: nothing is active, the cursor stays)
        emit(pa);
        fetch(pa, f + 2 * 128 * UN);
        emit(pb);
    }
    if (trace != nullptr && lane == 0) atomicMax((unsigned long long *)&trace[4 * blockIdx.x + 2], (unsigned long long)wall_clock64());
}

// pb_accumulate.  One workgroup per row block: row sums in LDS, products streamed from the block's one contiguous run.
// ORDERED: the waves add their pieces of the stream in the stream's order -- a wave issues its LDS atomics only once the
// ticket names its piece, and hands the ticket on behind them (LDS executes a CU's instructions in arrival order, so whoever
// reads the new ticket is behind those adds; lanes of one instruction that meet on a row are applied in lane order).  The
// order of additions into a row is then fixed by the layout: same bits on every run.  The loads of a piece are in flight
// while its wave waits for its turn, and everything but the adds themselves is done before the turn (the first version
// computed addresses and skipped pads by branches inside it: phase 2 229 us on config 5; now 170, against 161 unordered).
template <int SPAN, bool DOT, bool ORDERED>
__global__ __launch_bounds__(SPAN / 16) void pb_accumulate(int B, const int *__restrict__ bnd, const int64_t *__restrict__ bin_ptr,
                                                          const double *__restrict__ prod, const unsigned short *__restrict__ lrow,
                                                          const double *__restrict__ diag, const double *__restrict__ x,
                                                          double *__restrict__ y, double *__restrict__ partials,
                                                          const int *__restrict__ done, long long *__restrict__ trace)
{
    constexpr int T = SPAN / 16, NW = T / 64;
    __shared__ double acc[SPAN + 64];  // the block's row sums; behind them one slot per lane where the pads' products go
    __shared__ double red[NW];
    __shared__ int    ticket;
    if (done != nullptr && *done != 0) return;
    const int b = B - 1 - (int)blockIdx.x;
    if (trace != nullptr && threadIdx.x == 0) trace[4 * blockIdx.x] = wall_clock64();
    const int r0 = bnd[b], cnt = bnd[b + 1] - r0;
    for (int i = threadIdx.x; i < cnt; i += T) acc[i] = 0.0;
    if (threadIdx.x < 64) acc[SPAN + threadIdx.x] = 0.0;
    if (threadIdx.x == 0) ticket = 0;
    __syncthreads();
    const int64_t k0 = bin_ptr[b], k1 = bin_ptr[b + 1];  // both ends are multiples of PAD
    const int     wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // a piece = UN x 128 consecutive entries of ONE wave (ORDERED) / UN x 2 T entries of the workgroup, lane pairs interleaved
    constexpr int UN = ORDERED ? 8 : 4;
    const int64_t pieces = ORDERED ? (k1 - k0 + UN * 128 - 1) / (UN * 128) : (k1 - k0 + 2 * UN * T - 1) / (2 * UN * T);
    if constexpr (ORDERED) {
        for (int64_t pc = wave; pc < pieces; pc += NW) {
            d2_t     p[UN];
            unsigned r[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const int64_t kq = k0 + pc * (UN * 128) + q * 128 + 2 * lane;
                p[q] = d2_t{0.0, 0.0};
                r[q] = 0xffffffffu;
                if (kq < k1) {
                    p[q] = __builtin_nontemporal_load((const d2_t *)(prod + kq));
                    r[q] = __builtin_nontemporal_load((const unsigned *)(lrow + kq));
                }
            }
            // Everything that is not an LDS add happens BEFORE the wave's turn: the addresses (pads go to the lane's own
            // spare slot: no branch per entry inside the turn) need the last local rows loaded, and loads return in order,
            // so the products are in registers too.  What the other waves wait for is then 2 UN adds and the hand-over.
            unsigned a0[UN], a1[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const unsigned lo = r[q] & 0xffffu, hi = r[q] >> 16;
                a0[q] = (lo != 0xffffu ? lo : (unsigned)(SPAN + lane)) * 8u;  // byte offsets into acc
                a1[q] = (hi != 0xffffu ? hi : (unsigned)(SPAN + lane)) * 8u;
                // (pinned here: the compiler would otherwise sink the wait for the loads and this arithmetic into the turn)
                asm volatile("" : "+v"(a0[q]), "+v"(a1[q]), "+v"(p[q]));
            }
            // acquire / release (an lgkmcnt wait each on gfx950): the adds below stay behind the ticket they wait for and
            // ahead of the one they hand on, whatever the compiler would like to do with relaxed atomics to other addresses
            while (__hip_atomic_load(&ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (int)pc) {
            }
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                unsafeAtomicAdd((double *)((char *)acc + a0[q]), p[q].x);
                unsafeAtomicAdd((double *)((char *)acc + a1[q]), p[q].y);
            }
            if (lane == 0) __hip_atomic_store(&ticket, (int)pc + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        for (int64_t pc = 0; pc < pieces; ++pc) {
            d2_t     p[UN];
            unsigned r[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const int64_t kq = k0 + pc * 2 * UN * T + q * 2 * T + 2 * threadIdx.x;
                p[q] = d2_t{0.0, 0.0};
                r[q] = 0xffffffffu;
                if (kq < k1) {
                    p[q] = __builtin_nontemporal_load((const d2_t *)(prod + kq));
                    r[q] = __builtin_nontemporal_load((const unsigned *)(lrow + kq));
                }
            }
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                if ((r[q] & 0xffffu) != 0xffffu) unsafeAtomicAdd(&acc[r[q] & 0xffffu], p[q].x);
                if ((r[q] >> 16) != 0xffffu) unsafeAtomicAdd(&acc[r[q] >> 16], p[q].y);
            }
        }
    }
    __syncthreads();
    if (trace != nullptr && threadIdx.x == 0) {
        trace[4 * blockIdx.x + 1] = wall_clock64();
        trace[4 * blockIdx.x + 3] = k1 - k0;
    }
    double local = 0.0;
    // diag and x are read once, here; cnt <= SPAN = 16 T: four rows per lane and round trip
    for (int i0 = threadIdx.x; i0 < cnt; i0 += 4 * T) {
        double dg[4], xr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * T;
            dg[q] = i < cnt ? diag[r0 + i] : 0.0;
            xr[q] = i < cnt ? x[r0 + i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * T;
            if (i < cnt) {
                const double yv = add_rn(0.0, mul_rn(dg[q], xr[q])) + acc[i];
                __builtin_nontemporal_store(yv, y + r0 + i);
                if (DOT) local += xr[q] * yv;
            }
        }
    }
    if (trace != nullptr && (threadIdx.x & 63) == 0)
        atomicMax((unsigned long long *)&trace[4 * blockIdx.x + 2], (unsigned long long)wall_clock64());
    if constexpr (DOT) {
        local = wave_sum(local);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += red[w];
            partials[blockIdx.x] = tot;
        }
    }
}

// workgroup time stamps of one SpMV (tools/trace_pb.py): 4 words per workgroup {start, operands staged / stream done, end,
// entries}, pb_expand's from word 0, pb_accumulate's from word 4 * 4096; NULL = off
long long *g_pb_trace = nullptr;

template <int SPAN>
void launch(const cgh_matrix *m, const double *x, double *y, double *dot_partials, const int *done, hipStream_t s)
{
    const int B = m->pb_blocks, split = m->pb_split > 0 ? m->pb_split : 1;
    hipLaunchKernelGGL((pb_expand<SPAN>), dim3(B * split), dim3(SPAN / 16), 0, s, B, split, (const int *)m->pb_bnd, (const int *)m->pb_pre,
                       (const unsigned *)m->pb_seg0, (const int64_t *)m->pb_cbase, (const double *)m->vals, (const unsigned short *)m->pb_lcol, x, m->pb_prod, done, B * split <= 4096 ? g_pb_trace : nullptr);
#define PB_ACC(D, O)                                                                                                                \
    hipLaunchKernelGGL((pb_accumulate<SPAN, D, O>), dim3(B), dim3(SPAN / 16), 0, s, B, (const int *)m->pb_bnd, (const int64_t *)m->pb_bin, \
                       (const double *)m->pb_prod, (const unsigned short *)m->pb_lrow, (const double *)m->diag, x, y, dot_partials, done,       \
                       g_pb_trace ? g_pb_trace + 4 * 4096 : nullptr)
    if (dot_partials && m->pb_ordered) PB_ACC(true, true);
    else if (dot_partials) PB_ACC(true, false);
    else if (m->pb_ordered) PB_ACC(false, true);
    else PB_ACC(false, false);
#undef PB_ACC
}

} // namespace

int cgh_launch_spmv_pb(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out, const int *done, hipStream_t s)
{
    if (grid_out) *grid_out = m->pb_blocks;
    if (m->pb_span == 8192) launch<8192>(m, x, y, dot_partials, done, s);
    else launch<16384>(m, x, y, dot_partials, done, s);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

// Cuts G groups of 256 rows (cum[g] = entries before group g, cum[G] = all) into blocks of at most max_groups groups
// and, where that leaves a choice, at most `target` entries (a single group heavier than that is a block of its own).
// bnd_g[0..B] receives the boundaries in groups.  Returns B.
extern "C" int64_t cgh_pb_partition_host(int64_t G, const int64_t *cum, int max_groups, int64_t target, int64_t *bnd_g)
{
    if (G <= 0 || !cum || !bnd_g || max_groups < 1) return 0;
    int64_t B = 0, g = 0;
    bnd_g[0] = 0;
    while (g < G) {
        int64_t e = g + 1;
        while (e < G && e - g < max_groups && cum[e + 1] - cum[g] <= target) ++e;
        bnd_g[++B] = e;
        g = e;
    }
    return B;
}

// CSR (full rows, device) -> PB.  `csr` is consumed: its diag is adopted, its arrays are freed.
static cgh_matrix *cgh_pb_from_csr_impl(cgh_matrix *csr)
{
    hipStream_t   s = g_cgh.stream;
    const int64_t n = csr->n, nnz = csr->nnz_off;
    const int     span = g_cfg.pb_span_bits == 13 ? 8192 : 16384;
    const int64_t G = (n + GROUP - 1) / GROUP;
    if (nnz >= 0xffffffffll || (G + span / GROUP - 1) / (span / GROUP) > MAXB) {
        cgh_set_error("two-phase SpMV layout: %lld rows / %lld entries exceed %d blocks of %d rows or 32-bit entry positions",
                      (long long)n, (long long)nnz, MAXB, span);
        cgh_matrix_free(csr);
        return nullptr;
    }
    cgh_matrix *m = new cgh_matrix();
    m->n = n;
    m->K = csr->K;
    m->layout = CGH_LAYOUT_PB;
    m->nnz_off = nnz;
    m->stored = nnz;
    m->pb_span = span;
    m->pb_split = g_cfg.pb_split > 0 ? g_cfg.pb_split : 1;
    m->pb_ordered = g_cfg.pb_ordered != 0;
    bool ok = true;
    // 1. the partition (host, from the row pointer sampled every 256 rows)
    std::vector<int64_t> cum((size_t)(G + 1)), bg((size_t)(G + 1));
    int64_t              B = 0;
    {
        dbuf d(sizeof(int64_t) * (size_t)(G + 1));
        ok = d.p != nullptr;
        if (ok) {
            hipLaunchKernelGGL(pb_sample_row_ptr, dim3(grid_for(G + 1)), dim3(256), 0, s, n, G, (const int64_t *)csr->row_ptr, d.as<int64_t>());
            ok = cgh_d2h(cum.data(), d.p, sizeof(int64_t) * (size_t)(G + 1)) == CGH_OK;
        }
    }
    if (ok) {
        int64_t target = g_cfg.pb_target_entries > 0 ? g_cfg.pb_target_entries : 131072;
        for (;;) {
            B = cgh_pb_partition_host(G, cum.data(), span / GROUP, target, bg.data());
            if (B <= MAXB) break;
            target *= 2;  // more blocks than a launch has partials for: heavier blocks (ends: the span cap alone admits <= MAXB)
        }
    }
    const int64_t nseg = B * B;
    if (ok) {
        std::vector<int>            hb((size_t)(B + 1));
        std::vector<unsigned short> hof((size_t)G);
        for (int64_t k = 0; k <= B; ++k) hb[(size_t)k] = (int)(bg[(size_t)k] * GROUP < n ? bg[(size_t)k] * GROUP : n);
        for (int64_t k = 0; k < B; ++k)
            for (int64_t g = bg[(size_t)k]; g < bg[(size_t)k + 1]; ++g) hof[(size_t)g] = (unsigned short)k;
        m->pb_blocks = (int)B;
        m->pb_bnd = (int *)cgh_alloc(sizeof(int) * (size_t)(B + 1));
        dbuf          of(sizeof(unsigned short) * (size_t)G);
        const size_t  e = (size_t)(nnz ? nnz : 1);
        dbuf          k0(sizeof(unsigned) * e), k1(sizeof(unsigned) * e), i0(sizeof(unsigned) * e), i1(sizeof(unsigned) * e), rid(sizeof(int) * e);
        dbuf          start(sizeof(int64_t) * (size_t)(nseg + 2));
        ok = m->pb_bnd && of.p && k0.p && k1.p && i0.p && i1.p && rid.p && start.p &&
             cgh_h2d(m->pb_bnd, hb.data(), sizeof(int) * (size_t)(B + 1)) == CGH_OK &&
             cgh_h2d(of.p, hof.data(), sizeof(unsigned short) * (size_t)G) == CGH_OK;
        // 2. entries sorted by (row block, column block); the sort is stable, so a segment keeps the rows' CSR order
        if (ok && nnz > 0) {
            hipLaunchKernelGGL(pb_entry_keys, dim3(g_cgh.num_cu * 8), dim3(256), 0, s, n, (int)B, (const int64_t *)csr->row_ptr,
                               (const int *)csr->cols, (const unsigned short *)of.p, k0.as<unsigned>(), i0.as<unsigned>(), rid.as<int>());
            int bits = 1;
            while (bits < 32 && (nseg >> bits) != 0) ++bits;
            size_t     tmp_bytes = 0;
            hipError_t he = rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0.as<unsigned>(), k1.as<unsigned>(), i0.as<unsigned>(),
                                                      i1.as<unsigned>(), (size_t)nnz, 0u, (unsigned)bits, s);
            dbuf       tmp(tmp_bytes);
            ok = he == hipSuccess && tmp.p;
            if (ok)
                he = rocprim::radix_sort_pairs(tmp.p, tmp_bytes, k0.as<unsigned>(), k1.as<unsigned>(), i0.as<unsigned>(), i1.as<unsigned>(),
                                               (size_t)nnz, 0u, (unsigned)bits, s);
            ok = ok && he == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        }
        // 3. segment starts, padded to multiples of PAD; the arrays the kernels read
        dbuf    pstart(sizeof(int64_t) * (size_t)(nseg + 2));
        int64_t epad = 0;
        ok = ok && pstart.p;
        if (ok) {
            hipLaunchKernelGGL(pb_seg_starts, dim3(grid_for(nnz + 1)), dim3(256), 0, s, nnz, nseg, (const unsigned *)k1.p, start.as<int64_t>());
            hipLaunchKernelGGL(pb_padded_lengths, dim3(grid_for(nseg)), dim3(256), 0, s, nseg, (const int64_t *)start.p, pstart.as<int64_t>());
            ok = hipGetLastError() == hipSuccess &&
                 cgh_exclusive_scan_i64(pstart.as<int64_t>(), pstart.as<int64_t>(), nseg, &epad, s) == CGH_OK;
        }
        if (ok && epad >= 0xffffffffll) {
            cgh_set_error("two-phase SpMV layout: %lld padded entries exceed 32-bit entry positions", (long long)epad);
            ok = false;
        }
        const size_t ep = (size_t)(epad ? epad : PAD);
        if (ok) {
            m->stored = epad;
            m->vals = (double *)cgh_alloc(sizeof(double) * ep);
            m->pb_lcol = (unsigned short *)cgh_alloc(sizeof(unsigned short) * ep);
            m->pb_lrow = (unsigned short *)cgh_alloc(sizeof(unsigned short) * ep);
            m->pb_prod = (double *)cgh_alloc(sizeof(double) * ep);
            m->pb_pre = (int *)cgh_alloc(sizeof(int) * (size_t)(B * (B + 1)));
            m->pb_seg0 = (unsigned *)cgh_alloc(sizeof(unsigned) * (size_t)nseg);
            m->pb_bin = (int64_t *)cgh_alloc(sizeof(int64_t) * (size_t)(B + 1));
            m->pb_cbase = (int64_t *)cgh_alloc(sizeof(int64_t) * (size_t)(B + 1));
            ok = m->vals && m->pb_lcol && m->pb_lrow && m->pb_prod && m->pb_pre && m->pb_seg0 && m->pb_bin && m->pb_cbase;
        }
        if (ok)  // pads: value 0, column 0, row 0xffff (skipped by pb_accumulate)
            ok = hipMemsetAsync(m->vals, 0, sizeof(double) * ep, s) == hipSuccess &&
                 hipMemsetAsync(m->pb_lcol, 0, sizeof(unsigned short) * ep, s) == hipSuccess &&
                 hipMemsetAsync(m->pb_lrow, 0xff, sizeof(unsigned short) * ep, s) == hipSuccess;
        if (ok) {
            // tables first: the gather needs each column block's run (its start = the totals of the blocks before it)
            int64_t total = 0;
            hipLaunchKernelGGL(pb_tables, dim3(grid_for(B)), dim3(256), 0, s, (int)B, (const int64_t *)pstart.p, m->pb_pre, m->pb_seg0, m->pb_bin,
                               m->pb_cbase);
            ok = hipGetLastError() == hipSuccess && cgh_exclusive_scan_i64(m->pb_cbase, m->pb_cbase, B, &total, s) == CGH_OK && total == epad;
            if (ok && nnz > 0)
                hipLaunchKernelGGL(pb_gather_sorted, dim3(grid_for(nnz)), dim3(256), 0, s, nnz, (int)B, (const unsigned *)k1.p,
                                   (const unsigned *)i1.p, (const double *)csr->vals, (const int *)csr->cols, (const int *)rid.p,
                                   (const int *)m->pb_bnd, (const int64_t *)start.p, (const int64_t *)pstart.p, (const int *)m->pb_pre,
                                   (const int64_t *)m->pb_cbase, m->vals, m->pb_lcol, m->pb_lrow);
            ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        }
    }
    if (ok) {
        m->diag = csr->diag;  // adopt
        csr->diag = nullptr;
    }
    cgh_matrix_free(csr);
    if (!ok) {
        if (!g_cgh.err[0]) cgh_set_error("two-phase SpMV layout: construction failed");
        cgh_matrix_free(m);
        return nullptr;
    }
    return m;
}

// (host containers inside: nothing may leave through the extern "C" callers as an exception)
cgh_matrix *cgh_pb_from_csr(cgh_matrix *csr)
{
    try {
        return cgh_pb_from_csr_impl(csr);
    } catch (const std::exception &e) {
        cgh_set_error("two-phase SpMV layout: %s", e.what());
        return nullptr;  // (csr may be lost with it: the process is out of host memory)
    }
}

extern "C" void cgh_pb_set_trace(void *dev_8_x_4096_words) { g_pb_trace = (long long *)dev_8_x_4096_words; }

extern "C" int cgh_config_pb(int span_bits, int64_t target_entries, int split, int ordered)
{
    if ((span_bits != 0 && span_bits != 13 && span_bits != 14) || target_entries < 0 || split < 0 || split > 16 || ordered > 1) {
        cgh_set_error("cgh_config_pb: rows per block 2^13 | 2^14 (0 = keep), entries per block >= 0 (0 = keep), workgroups per column block 0..16, "
                      "ordered 0 | 1 (< 0 = keep)");
        return CGH_EINVAL;
    }
    if (ordered >= 0) cgh_config_set(&g_cfg.pb_ordered, ordered);
    if (span_bits) cgh_config_set(&g_cfg.pb_span_bits, span_bits);
    if (target_entries) cgh_config_set(&g_cfg.pb_target_entries, target_entries);
    if (split) cgh_config_set(&g_cfg.pb_split, split);
    return CGH_OK;
}
