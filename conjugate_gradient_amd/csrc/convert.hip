// convert.hip -- device-side format construction: COO (LDU triple) -> ELL / CSR / sliced ELL.
//
// Replaces reference src/ell.c:262-309 (coo_to_ell_symmetric_upper), :315-396
// (compute_max_nnz_row_{upper,full}), :714-776 (..._upper_colmajor), :791-858
// (..._full_colmajor), :873-940 (..._sdtint) and src/csr.c:233-272 (coo_to_csr), :287-336
// (sorting_coo_to_csr).
//
// The reference converters are sequential scatters with one running counter per row: the slot an
// entry gets is the number of EARLIER placements into the same row.  Entry k = (i, j, v) makes up to
// two placements, numbered seq = 2k (into row i, column j) and seq = 2k+1 (into row j, column i);
// the reference performs them in increasing seq.  So
//
//     slot(placement) = #{ placements into the same row with a smaller seq }
//
// which is order-independent and therefore parallel:
//   1. count      per-row placement counts (atomics)                      -> K = max, row_ptr = scan
//   2. bucket     every placement drops its seq into its row's bucket (atomic cursor; bucket order arbitrary)
//   3. place      every placement counts the smaller seqs in its bucket = its slot, and writes
//                 (value, column) at the layout's address for (row, slot)
// The result is bit-identical to the sequential reference for any input order (row-sorted or not).
// The reference stops at the first non-upper entry or the first placement whose slot is >= K and
// leaves what it wrote: `limit_seq` reproduces that (only placements with seq < limit are made).
#include <vector>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "cgh_internal.h"

// ------------------------------------------------------------------------------------ scans
namespace {

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

struct OpAdd {
    __device__ static int64_t id() { return 0; }
    __device__ static int64_t f(int64_t a, int64_t b) { return a + b; }
};
struct OpMax {
    __device__ static int64_t id() { return 0; }  // identity 0: inputs are non-negative row ids
    __device__ static int64_t f(int64_t a, int64_t b) { return a > b ? a : b; }
};

// Exclusive scan of the 256 per-thread totals of a workgroup; returns this thread's exclusive prefix and
// leaves the workgroup total in *total.
template <typename Op>
__device__ int64_t block_exclusive(int64_t v, int64_t *lds /*256*/, int64_t *total)
{
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int64_t t = threadIdx.x >= off ? lds[threadIdx.x - off] : Op::id();
        __syncthreads();
        lds[threadIdx.x] = Op::f(lds[threadIdx.x], t);
        __syncthreads();
    }
    *total = lds[255];
    const int64_t excl = threadIdx.x ? lds[threadIdx.x - 1] : Op::id();
    __syncthreads();
    return excl;
}

template <typename Tin, typename Op>
__global__ __launch_bounds__(256) void scan_tile_totals(const Tin *__restrict__ in, int64_t count, int64_t *__restrict__ totals)
{
    __shared__ int64_t lds[256];
    const int64_t      base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t            acc = Op::id();
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < count) acc = Op::f(acc, (int64_t)in[base + k]);
    int64_t total;
    block_exclusive<Op>(acc, lds, &total);
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

template <typename Tin, typename Op>
__global__ __launch_bounds__(256) void scan_tile_apply(const Tin *in, int64_t *out, int64_t count,
                                                      const int64_t *__restrict__ tile_offsets)
{
    __shared__ int64_t lds[256];
    const int64_t      base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t            v[SCAN_ITEMS];
    int64_t            acc = Op::id();
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = base + k < count ? (int64_t)in[base + k] : Op::id();
        acc = Op::f(acc, v[k]);
    }
    int64_t       total;
    const int64_t excl = block_exclusive<Op>(acc, lds, &total);  // every item is in registers: in-place is safe
    int64_t       run = Op::f(tile_offsets ? tile_offsets[blockIdx.x] : Op::id(), excl);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < count) out[base + k] = run;
        run = Op::f(run, v[k]);
    }
}

template <typename Tin, typename Op>
int exclusive_scan(const Tin *d_in, int64_t *d_out, int64_t count, hipStream_t s)
{
    if (count <= 0) return CGH_OK;
    const int64_t tiles = (count + SCAN_TILE - 1) / SCAN_TILE;
    if (tiles == 1) {
        hipLaunchKernelGGL((scan_tile_apply<Tin, Op>), dim3(1), dim3(256), 0, s, d_in, d_out, count, (const int64_t *)nullptr);
        CGH_HIP(hipGetLastError());
        return CGH_OK;
    }
    int64_t *totals = (int64_t *)cgh_alloc(sizeof(int64_t) * (size_t)tiles);
    if (!totals) return CGH_ENOMEM;
    hipLaunchKernelGGL((scan_tile_totals<Tin, Op>), dim3((unsigned)tiles), dim3(256), 0, s, d_in, count, totals);
    int rc = exclusive_scan<int64_t, Op>(totals, totals, tiles, s);
    if (rc == CGH_OK) {
        hipLaunchKernelGGL((scan_tile_apply<Tin, Op>), dim3((unsigned)tiles), dim3(256), 0, s, d_in, d_out, count,
                           (const int64_t *)totals);
        if (hipGetLastError() != hipSuccess) rc = CGH_EHIP;
    }
    (void)hipStreamSynchronize(s);  // totals is freed below
    cgh_free(totals);
    return rc;
}

} // namespace

// d_out[i] = sum of d_in[0..i); d_out may alias d_in.  If h_total != NULL the grand total is returned
// (d_in must then hold count+1 slots: slot [count] is written with the total).
int cgh_exclusive_scan_i64(const int64_t *d_in, int64_t *d_out, int64_t count, int64_t *h_total, hipStream_t s)
{
    if (h_total) {
        // scan count+1 items with a zero appended: out[count] = total
        CGH_HIP(hipMemsetAsync((void *)(d_in + count), 0, sizeof(int64_t), s));
        int rc = exclusive_scan<int64_t, OpAdd>(d_in, d_out, count + 1, s);
        if (rc) return rc;
        CGH_HIP(hipMemcpyAsync(h_total, d_out + count, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        CGH_HIP(hipStreamSynchronize(s));
        return CGH_OK;
    }
    return exclusive_scan<int64_t, OpAdd>(d_in, d_out, count, s);
}

int cgh_exclusive_scan_i32_to_i64(const int *d_in, int64_t *d_out, int64_t count, hipStream_t s)
{
    return exclusive_scan<int, OpAdd>(d_in, d_out, count, s);
}

// ------------------------------------------------------------------------------ the bucket engine
namespace {

constexpr unsigned long long NO_STOP = ~0ull;

enum { OUT_ELL_SLOT_MAJOR = 0, OUT_ELL_ROW_MAJOR = 1, OUT_CSR = 2, OUT_SELL = 3 };

// First seq at which the reference would stop because of a bad entry (col <= row, or an index outside [0,n)).
__global__ __launch_bounds__(256) void find_bad_entry(int64_t U, int64_t n, const int *__restrict__ rows,
                                                     const int *__restrict__ cols, int need_upper,
                                                     unsigned long long *stop)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < U; k += (int64_t)gridDim.x * 256) {
        const int i = rows[k], j = cols[k];
        const bool bad = i < 0 || j < 0 || i >= n || j >= n || (need_upper == 1 && j <= i) || (need_upper == 2 && j < i);
        if (bad) atomicMin(stop, (unsigned long long)(2 * k));
    }
}

template <bool BOTH>
__global__ __launch_bounds__(256) void count_placements(int64_t U, unsigned long long limit, const int *__restrict__ rows,
                                                       const int *__restrict__ cols, int *__restrict__ cnt, int diag_twice)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < U; k += (int64_t)gridDim.x * 256) {
        const unsigned long long s0 = 2ull * k;
        if (s0 < limit) atomicAdd(&cnt[rows[k]], 1);
        // a diagonal entry is placed once -- unless the source's SpMV scatters whatever it is given (COO / CSR, diag_twice)
        if (BOTH && s0 + 1 < limit && (cols[k] != rows[k] || diag_twice)) atomicAdd(&cnt[cols[k]], 1);
    }
}

__global__ __launch_bounds__(256) void max_count(int64_t n, const int *__restrict__ cnt, int *result)
{
    int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = cnt[i] > m ? cnt[i] : m;
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(result, m);
}

template <bool BOTH>
__global__ __launch_bounds__(256) void bucket_placements(int64_t U, unsigned long long limit, const int *__restrict__ rows,
                                                        const int *__restrict__ cols, const int64_t *__restrict__ row_ptr,
                                                        int *__restrict__ cursor, unsigned *__restrict__ seqs, int diag_twice)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < U; k += (int64_t)gridDim.x * 256) {
        const unsigned long long s0 = 2ull * k;
        if (s0 < limit) {
            const int r = rows[k];
            seqs[row_ptr[r] + atomicAdd(&cursor[r], 1)] = (unsigned)s0;
        }
        if (BOTH && s0 + 1 < limit && (cols[k] != rows[k] || diag_twice)) {
            const int r = cols[k];
            seqs[row_ptr[r] + atomicAdd(&cursor[r], 1)] = (unsigned)(s0 + 1);
        }
    }
}

struct out_desc {
    int      mode;
    int64_t  n;
    int      K;              // ELL width (slot bound); unused for CSR
    double  *vals;
    void    *cols;           // int* or uint64_t*
    const int64_t *row_ptr;  // CSR destination offsets (= bucket offsets)
    const int64_t *slice_ptr;  // SELL
    const int     *row_pos = nullptr;  // sigma-sorted SELL: position (slice*64 + lane) of each row; NULL = the row itself
};

template <typename ColT>
__device__ __forceinline__ void place_one(const out_desc &o, int r, int slot, double v, int c, unsigned long long seq,
                                          unsigned long long *overflow_stop)
{
    int64_t at;
    switch (o.mode) {
    case OUT_ELL_SLOT_MAJOR:
        if (slot >= o.K) { atomicMin(overflow_stop, seq); return; }
        at = (int64_t)slot * o.n + r;
        break;
    case OUT_ELL_ROW_MAJOR:
        if (slot >= o.K) { atomicMin(overflow_stop, seq); return; }
        at = (int64_t)r * o.K + slot;
        break;
    case OUT_CSR: at = o.row_ptr[r] + slot; break;
    default: {
        const int p = o.row_pos ? o.row_pos[r] : r;
        at = o.slice_ptr[p >> 6] + (int64_t)slot * 64 + (p & 63);
        break;
    }
    }
    o.vals[at] = v;
    reinterpret_cast<ColT *>(o.cols)[at] = (ColT)c;
}

template <bool BOTH, typename ColT>
__global__ __launch_bounds__(256) void place_entries(int64_t U, unsigned long long limit, const int *__restrict__ rows,
                                                    const int *__restrict__ cols, const double *__restrict__ vals,
                                                    const int64_t *__restrict__ row_ptr, const unsigned *__restrict__ seqs,
                                                    out_desc o, unsigned long long *overflow_stop, int diag_twice = 0)
{
    const int sides = BOTH ? 2 : 1;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < U * sides; w += (int64_t)gridDim.x * 256) {
        const int64_t            k = BOTH ? (w >> 1) : w;
        const int                side = BOTH ? (int)(w & 1) : 0;
        const unsigned long long seq = 2ull * k + side;
        if (seq >= limit) continue;
        if (side && cols[k] == rows[k] && !diag_twice) continue;
        const int     r = side ? cols[k] : rows[k];
        const int     c = side ? rows[k] : cols[k];
        const int64_t b0 = row_ptr[r], b1 = row_ptr[r + 1];
        int           slot = 0;
        for (int64_t q = b0; q < b1; ++q) slot += seqs[q] < (unsigned)seq;
        place_one<ColT>(o, r, slot, vals[k], c, seq, overflow_stop);
    }
}

template <typename ColT>
__global__ __launch_bounds__(256) void fill_pad(int64_t total, double *__restrict__ vals, ColT *__restrict__ cols)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        vals[i] = 0.0;
        cols[i] = (ColT)-1;
    }
}

// SELL: width of slice s = max count over its 64 rows; stored entries of the slice = 64*width
__global__ __launch_bounds__(256) void slice_sizes(int64_t n, const int *__restrict__ cnt, int64_t nslices, int64_t *__restrict__ sizes,
                                                  const int *__restrict__ perm)
{
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nslices) return;
    int64_t r = s * 64 + (threadIdx.x & 63);
    if (perm != nullptr) r = perm[r] < 0 ? n : perm[r];
    int           m = r < n ? cnt[r] : 0;
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) sizes[s] = (int64_t)m * 64;
}

// coo_to_csr (reference src/csr.c:233-272): row_ptr[r+1] = first entry index at which the running maximum of
// the row ids exceeds r; trailing rows get U.  emax = exclusive running max (identity 0).
__global__ __launch_bounds__(256) void row_ptr_from_running_max(int64_t U, int64_t n, const int *__restrict__ rows,
                                                               const int64_t *__restrict__ emax, int *__restrict__ row_ptr)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= U; i += (int64_t)gridDim.x * 256) {
        if (i == U) {
            // tail: rows from the final running max to n-1
            int64_t m = 0;
            if (U > 0) {
                m = emax[U - 1];
                if (rows[U - 1] > m) m = rows[U - 1];
            }
            for (int64_t r = m; r < n; ++r) row_ptr[r + 1] = (int)U;
            if (m <= n) row_ptr[0] = 0;
            continue;
        }
        const int64_t lo = emax[i];
        int64_t       hi = rows[i] > lo ? rows[i] : lo;
        if (hi > n) hi = n;
        for (int64_t r = lo; r < hi; ++r) row_ptr[r + 1] = (int)i;
    }
}

__global__ __launch_bounds__(256) void count_nonpad(int64_t total, const int *__restrict__ cols, unsigned long long *out)
{
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) c += cols[i] != -1;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// any layout -> slot-major ELL of width K (download / debugging)
__global__ __launch_bounds__(256) void to_ell_rows(int64_t n, int K, int layout, const double *__restrict__ vals,
                                                  const int *__restrict__ cols, const int64_t *__restrict__ row_ptr,
                                                  const int64_t *__restrict__ slice_ptr, double *__restrict__ ev,
                                                  int *__restrict__ ec, const int *__restrict__ row_pos)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        int slot = 0;
        if (layout == CGH_LAYOUT_CSR) {
            for (int64_t q = row_ptr[r]; q < row_ptr[r + 1]; ++q, ++slot) {
                ev[(int64_t)slot * n + r] = vals[q];
                ec[(int64_t)slot * n + r] = cols[q];
            }
        } else {
            const int64_t p = row_pos ? row_pos[r] : r;
            const int64_t b = slice_ptr[p >> 6];
            const int     w = (int)((slice_ptr[(p >> 6) + 1] - b) >> 6);
            for (int t = 0; t < w; ++t) {
                const int c = cols[b + (int64_t)t * 64 + (p & 63)];
                if (c == -1) continue;
                ev[(int64_t)slot * n + r] = vals[b + (int64_t)t * 64 + (p & 63)];
                ec[(int64_t)slot * n + r] = c;
                ++slot;
            }
        }
        for (; slot < K; ++slot) {
            ev[(int64_t)slot * n + r] = 0.0;
            ec[(int64_t)slot * n + r] = -1;
        }
    }
}

__global__ __launch_bounds__(256) void min_row_length(int64_t n, const int64_t *__restrict__ row_ptr, int *result)
{
    int m = 0x7fffffff;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int len = (int)(row_ptr[i + 1] - row_ptr[i]);
        m = len < m ? len : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_down(m, off, 64);
        m = o < m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMin(result, m);
}

// sigma-sorted SELL: key = (window, descending length), value = row; a stable sort leaves equal lengths in row order
__global__ __launch_bounds__(256) void sigma_keys(int64_t n, int sigma, const int *__restrict__ cnt, unsigned long long *__restrict__ keys,
                                                 int *__restrict__ rows)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        keys[r] = ((unsigned long long)(r / sigma) << 32) | (unsigned long long)(0xffffffffu - (unsigned)cnt[r]);
        rows[r] = (int)r;
    }
}

// perm[p] = row at position p (padded with -1 up to a whole slice); pos[row] = p
__global__ __launch_bounds__(256) void sigma_positions(int64_t n, int64_t padded, const int *__restrict__ sorted_rows, int *__restrict__ perm,
                                                      int *__restrict__ pos)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < padded; p += (int64_t)gridDim.x * 256) {
        const int r = p < n ? sorted_rows[p] : -1;
        perm[p] = r;
        if (r >= 0 && pos != nullptr) pos[r] = (int)p;
    }
}

// slices by descending stored size (stable: equal sizes stay in window order)
__global__ __launch_bounds__(256) void slice_keys(int64_t ns, const int64_t *__restrict__ sizes, unsigned long long *__restrict__ keys,
                                                 int *__restrict__ ids)
{
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < ns; t += (int64_t)gridDim.x * 256) {
        keys[t] = ~(unsigned long long)sizes[t];
        ids[t] = (int)t;
    }
}

// slice t of the final order is slice order[t] of the window order; pos[row] = its final position
__global__ __launch_bounds__(256) void permute_slices(int64_t ns, const int *__restrict__ order, const int *__restrict__ perm_in,
                                                     int *__restrict__ perm_out, int *__restrict__ pos)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < ns * 64; p += (int64_t)gridDim.x * 256) {
        const int r = perm_in[(int64_t)order[p >> 6] * 64 + (p & 63)];
        perm_out[p] = r;
        if (r >= 0) pos[r] = (int)p;
    }
}

struct dev_buf {
    void *p = nullptr;
    explicit dev_buf(size_t bytes) { p = cgh_alloc(bytes); }
    ~dev_buf() { cgh_free(p); }
    dev_buf(const dev_buf &) = delete;
    dev_buf &operator=(const dev_buf &) = delete;
    template <typename T> T *as() { return (T *)p; }
};

inline int grid_for(int64_t work) { return cgh_stream_grid((work + 255) / 256, 8); }

} // namespace

// Result of the bucket engine's analysis phase.
struct cgh_buckets {
    int64_t            n = 0, U = 0;
    bool               both = true;
    bool               diag_twice = false;  // an entry (i, i) is placed twice in row i (COO / CSR scatter semantics)
    unsigned long long limit = NO_STOP;  // placements with seq >= limit are not made
    unsigned long long bad_seq = NO_STOP;  // first bad entry (seq = 2k), NO_STOP if none
    int                max_count = 0;
    int               *cnt = nullptr;      // [n]   per-row placement count
    int64_t           *row_ptr = nullptr;  // [n+1] bucket offsets
    unsigned          *seqs = nullptr;     // [placements]
    int64_t            placements = 0;
    ~cgh_buckets()
    {
        cgh_free(cnt);
        cgh_free(row_ptr);
        cgh_free(seqs);
    }
};

// Phase 1+2: validate, count, scan, bucket.  d_rows/d_cols are device arrays.
int cgh_buckets_build(cgh_buckets *B, int64_t n, int64_t U, const int *d_rows, const int *d_cols, bool both, int need_upper,
                      hipStream_t s)
{
    if (2 * (uint64_t)U + 1 >= 0xffffffffull) {
        cgh_set_error("conversion: %lld upper entries exceed the 32-bit placement ids", (long long)U);
        return CGH_EINVAL;
    }
    B->n = n;
    B->U = U;
    B->both = both;
    B->diag_twice = both && need_upper == 0;  // any (i, j): the symmetric scatter of the COO / CSR SpMVs, (i, i) included
    dev_buf stop(sizeof(unsigned long long)), mx(sizeof(int));
    B->cnt = (int *)cgh_alloc(sizeof(int) * (size_t)(n + 1));
    B->row_ptr = (int64_t *)cgh_alloc(sizeof(int64_t) * (size_t)(n + 2));
    if (!stop.p || !mx.p || !B->cnt || !B->row_ptr) return CGH_ENOMEM;
    CGH_HIP(hipMemsetAsync(stop.p, 0xff, sizeof(unsigned long long), s));
    CGH_HIP(hipMemsetAsync(mx.p, 0, sizeof(int), s));
    CGH_HIP(hipMemsetAsync(B->cnt, 0, sizeof(int) * (size_t)(n + 1), s));
    const int g = grid_for(U);
    if (U > 0) hipLaunchKernelGGL(find_bad_entry, dim3(g), dim3(256), 0, s, U, n, d_rows, d_cols, need_upper, stop.as<unsigned long long>());
    CGH_HIP(hipMemcpyAsync(&B->bad_seq, stop.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    CGH_HIP(hipStreamSynchronize(s));
    B->limit = B->bad_seq;
    if (U > 0) {
        if (both)
            hipLaunchKernelGGL(count_placements<true>, dim3(g), dim3(256), 0, s, U, B->limit, d_rows, d_cols, B->cnt, B->diag_twice ? 1 : 0);
        else
            hipLaunchKernelGGL(count_placements<false>, dim3(g), dim3(256), 0, s, U, B->limit, d_rows, d_cols, B->cnt, 0);
    }
    hipLaunchKernelGGL(max_count, dim3(grid_for(n)), dim3(256), 0, s, n, (const int *)B->cnt, mx.as<int>());
    int rc = cgh_exclusive_scan_i32_to_i64(B->cnt, B->row_ptr, n + 1, s);  // cnt[n] = 0 -> row_ptr[n] = total
    if (rc) return rc;
    CGH_HIP(hipMemcpyAsync(&B->max_count, mx.p, sizeof(int), hipMemcpyDeviceToHost, s));
    CGH_HIP(hipMemcpyAsync(&B->placements, B->row_ptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    CGH_HIP(hipStreamSynchronize(s));
    B->seqs = (unsigned *)cgh_alloc(sizeof(unsigned) * (size_t)(B->placements ? B->placements : 1));
    dev_buf cursor(sizeof(int) * (size_t)(n + 1));
    if (!B->seqs || !cursor.p) return CGH_ENOMEM;
    CGH_HIP(hipMemsetAsync(cursor.p, 0, sizeof(int) * (size_t)(n + 1), s));
    if (U > 0) {
        if (both)
            hipLaunchKernelGGL(bucket_placements<true>, dim3(g), dim3(256), 0, s, U, B->limit, d_rows, d_cols,
                               (const int64_t *)B->row_ptr, cursor.as<int>(), B->seqs, B->diag_twice ? 1 : 0);
        else
            hipLaunchKernelGGL(bucket_placements<false>, dim3(g), dim3(256), 0, s, U, B->limit, d_rows, d_cols,
                               (const int64_t *)B->row_ptr, cursor.as<int>(), B->seqs, 0);
    }
    CGH_HIP(hipGetLastError());
    CGH_HIP(hipStreamSynchronize(s));  // cursor is freed on return
    return CGH_OK;
}

// Phase 3 into ELL arrays of width K (slot- or row-major; int or uint64_t columns), pre-filled with pads.
// *stop_seq_out = seq at which the reference would have stopped (bad entry or slot overflow), NO_STOP if none.
template <typename ColT>
static int place_ell(cgh_buckets *B, const int *d_rows, const int *d_cols, const double *d_vals, int K, bool row_major,
                     double *d_ev, ColT *d_ec, unsigned long long *stop_seq_out, hipStream_t s)
{
    const int64_t total = B->n * (int64_t)K;
    dev_buf       ov(sizeof(unsigned long long));
    if (!ov.p) return CGH_ENOMEM;
    out_desc o;
    o.mode = row_major ? OUT_ELL_ROW_MAJOR : OUT_ELL_SLOT_MAJOR;
    o.n = B->n;
    o.K = K;
    o.vals = d_ev;
    o.cols = d_ec;
    o.row_ptr = B->row_ptr;
    o.slice_ptr = nullptr;
    unsigned long long limit = B->limit;
    for (int pass = 0; pass < 2; ++pass) {
        if (total > 0) hipLaunchKernelGGL(fill_pad<ColT>, dim3(grid_for(total)), dim3(256), 0, s, total, d_ev, d_ec);
        CGH_HIP(hipMemsetAsync(ov.p, 0xff, sizeof(unsigned long long), s));
        const int64_t work = B->U * (B->both ? 2 : 1);
        if (work > 0) {
            if (B->both)
                hipLaunchKernelGGL((place_entries<true, ColT>), dim3(grid_for(work)), dim3(256), 0, s, B->U, limit, d_rows,
                                   d_cols, d_vals, (const int64_t *)B->row_ptr, (const unsigned *)B->seqs, o,
                                   ov.as<unsigned long long>());
            else
                hipLaunchKernelGGL((place_entries<false, ColT>), dim3(grid_for(work)), dim3(256), 0, s, B->U, limit, d_rows,
                                   d_cols, d_vals, (const int64_t *)B->row_ptr, (const unsigned *)B->seqs, o,
                                   ov.as<unsigned long long>());
        }
        CGH_HIP(hipGetLastError());
        unsigned long long ovs = NO_STOP;
        CGH_HIP(hipMemcpyAsync(&ovs, ov.p, sizeof ovs, hipMemcpyDeviceToHost, s));
        CGH_HIP(hipStreamSynchronize(s));
        if (ovs == NO_STOP) break;
        limit = ovs;  // a slot overflowed: redo, making only the placements before it (ranks below it are unaffected)
    }
    if (stop_seq_out) *stop_seq_out = limit;
    return CGH_OK;
}

int cgh_convert_coo_to_ell_device(int64_t n, int64_t U, const double *d_vals, const int *d_rows, const int *d_cols, int K,
                                  bool both, bool row_major, bool wide_cols, double *d_ev, void *d_ec,
                                  unsigned long long *stop_seq, hipStream_t s)
{
    cgh_buckets B;
    int         rc = cgh_buckets_build(&B, n, U, d_rows, d_cols, both, 1, s);
    if (rc) return rc;
    if (wide_cols) return place_ell<uint64_t>(&B, d_rows, d_cols, d_vals, K, row_major, d_ev, (uint64_t *)d_ec, stop_seq, s);
    return place_ell<int>(&B, d_rows, d_cols, d_vals, K, row_major, d_ev, (int *)d_ec, stop_seq, s);
}

// compute_max_nnz_row_{full,upper}: -1 on a bad entry (reference ell.c:333-340, :377-382), else the max count.
int cgh_max_nnz_row_device(int64_t n, int64_t U, const int *d_rows, const int *d_cols, bool both, int64_t *bad_entry, hipStream_t s)
{
    cgh_buckets B;
    int         rc = cgh_buckets_build(&B, n, U, d_rows, d_cols, both, 1, s);
    if (rc) return rc;
    if (bad_entry) *bad_entry = B.bad_seq == NO_STOP ? -1 : (int64_t)(B.bad_seq >> 1);
    return B.bad_seq == NO_STOP ? B.max_count : -1;
}

// Stable counting sort of an arbitrary COO by row (reference csr.c:287-336): int32 row_ptr, cols, vals.
int cgh_sort_coo_to_csr_device(int64_t n, int64_t nnz, const double *d_vals, const int *d_rows, const int *d_cols,
                               int *d_row_ptr32, int *d_csr_cols, double *d_csr_vals, hipStream_t s);

int cgh_coo_row_ptr_device(int64_t n, int64_t U, const int *d_rows, int *d_row_ptr, hipStream_t s)
{
    dev_buf emax(sizeof(int64_t) * (size_t)(U + 1));
    if (!emax.p) return CGH_ENOMEM;
    int rc = exclusive_scan<int, OpMax>(d_rows, emax.as<int64_t>(), U, s);
    if (rc) return rc;
    hipLaunchKernelGGL(row_ptr_from_running_max, dim3(grid_for(U + 1)), dim3(256), 0, s, U, n, d_rows, (const int64_t *)emax.p,
                       d_row_ptr);
    CGH_HIP(hipGetLastError());
    CGH_HIP(hipStreamSynchronize(s));
    return CGH_OK;
}

namespace {
__global__ __launch_bounds__(256) void narrow_row_ptr(int64_t n1, const int64_t *__restrict__ in, int *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += (int64_t)gridDim.x * 256) out[i] = (int)in[i];
}
} // namespace

int cgh_sort_coo_to_csr_device(int64_t n, int64_t nnz, const double *d_vals, const int *d_rows, const int *d_cols,
                               int *d_row_ptr32, int *d_csr_cols, double *d_csr_vals, hipStream_t s)
{
    cgh_buckets B;
    int         rc = cgh_buckets_build(&B, n, nnz, d_rows, d_cols, false, 0, s);
    if (rc) return rc;
    if (B.bad_seq != NO_STOP) {
        cgh_set_error("sorting_coo_to_csr: entry %llu has an index outside [0,%lld)", B.bad_seq >> 1, (long long)n);
        return CGH_EINPUT;
    }
    out_desc o;
    o.mode = OUT_CSR;
    o.n = n;
    o.K = 0;
    o.vals = d_csr_vals;
    o.cols = d_csr_cols;
    o.row_ptr = B.row_ptr;
    o.slice_ptr = nullptr;
    dev_buf ov(sizeof(unsigned long long));
    if (!ov.p) return CGH_ENOMEM;
    if (nnz > 0)
        hipLaunchKernelGGL((place_entries<false, int>), dim3(grid_for(nnz)), dim3(256), 0, s, nnz, NO_STOP, d_rows, d_cols, d_vals,
                           (const int64_t *)B.row_ptr, (const unsigned *)B.seqs, o, ov.as<unsigned long long>());
    hipLaunchKernelGGL(narrow_row_ptr, dim3(grid_for(n + 1)), dim3(256), 0, s, n + 1, (const int64_t *)B.row_ptr, d_row_ptr32);
    CGH_HIP(hipGetLastError());
    CGH_HIP(hipStreamSynchronize(s));
    return CGH_OK;
}

int cgh_csr_build_row_blocks(cgh_matrix *m, hipStream_t s);  // spmv_csr.hip
cgh_matrix *cgh_csrb_from_csr(cgh_matrix *csr);                // spmv_csrb.hip
cgh_matrix *cgh_pb_from_csr(cgh_matrix *csr);                  // spmv_pb.hip

int cgh_csr_to_ell_download(const cgh_matrix *m, double *ell_vals, int *ell_cols)
{
    const int64_t total = m->n * (int64_t)m->K;
    dev_buf       ev(sizeof(double) * (size_t)(total ? total : 1)), ec(sizeof(int) * (size_t)(total ? total : 1));
    if (!ev.p || !ec.p) return CGH_ENOMEM;
    if (m->layout == CGH_LAYOUT_CSRB || m->layout == CGH_LAYOUT_PB) {
        cgh_set_error("cgh_matrix_download_ell: the column-blocked layouts do not keep the rows' original entry order");
        return CGH_EINVAL;
    }
    dev_buf pos(sizeof(int) * (size_t)m->n);
    if (!pos.p) return CGH_ENOMEM;
    if (m->perm != nullptr) {
        dev_buf scratch(sizeof(int) * (size_t)(m->nslices * 64));
        if (!scratch.p) return CGH_ENOMEM;
        hipLaunchKernelGGL(sigma_positions, dim3(grid_for(m->nslices * 64)), dim3(256), 0, g_cgh.stream, m->nslices * 64, m->nslices * 64,
                           (const int *)m->perm, scratch.as<int>(), pos.as<int>());
        CGH_HIP(hipStreamSynchronize(g_cgh.stream));
    }
    hipLaunchKernelGGL(to_ell_rows, dim3(grid_for(m->n)), dim3(256), 0, g_cgh.stream, m->n, m->K,
                       m->layout == CGH_LAYOUT_CSR ? CGH_LAYOUT_CSR : CGH_LAYOUT_SELL, (const double *)m->vals, (const int *)m->cols,
                       (const int64_t *)m->row_ptr, (const int64_t *)m->slice_ptr, ev.as<double>(), ec.as<int>(),
                       m->perm ? (const int *)pos.p : (const int *)nullptr);
    CGH_HIP(hipGetLastError());
    int rc = cgh_d2h(ell_vals, ev.p, sizeof(double) * (size_t)total);
    if (rc) return rc;
    return cgh_d2h(ell_cols, ec.p, sizeof(int) * (size_t)total);
}

extern "C" {

// Shortest row (stored off-diagonals) of a CSR-layout matrix; -1 for other layouts.
int cgh_matrix_set_exact_rows(cgh_matrix *m, int exact)
{
    if (!m) return CGH_EINVAL;
    m->exact_rows = exact != 0;
    return CGH_OK;
}

int cgh_matrix_min_row(const cgh_matrix *m)
{
    CGH_REQUIRE_READY(-1);
    if (!m || m->layout != CGH_LAYOUT_CSR) return -1;
    dev_buf r(sizeof(int));
    if (!r.p) return -1;
    int v = 0x7fffffff;
    if (cgh_h2d(r.p, &v, sizeof v)) return -1;
    hipLaunchKernelGGL(min_row_length, dim3(grid_for(m->n)), dim3(256), 0, g_cgh.stream, m->n, (const int64_t *)m->row_ptr, r.as<int>());
    if (cgh_d2h(&v, r.p, sizeof v)) return -1;
    return v;
}

cgh_matrix *cgh_matrix_from_coo_device(int64_t n, int64_t U, const double *d_diag, const double *d_upper, const int *d_rows,
                                       const int *d_cols, int layout)
{
    return cgh_matrix_from_entries_device(n, U, d_diag, d_upper, d_rows, d_cols, layout);
}

} // extern "C"

// d_diag == NULL: the entry list carries the diagonal itself as (i, i, a_ii) entries placed where the source
// format's SpMV adds them (row-major upper ELL, reference ell.c:182-210, adds diag AFTER the contributions of
// earlier rows); the matrix then gets an all-zero diag array, and 0.0 + 0.0*x = +0.0 starts each row's sum.
// any_entries: the list is what the reference's COO / CSR SpMVs are given (coo.c:202-233, csr.c:190-229 there): entry k
// adds v x[j] to y[i] and v x[i] to y[j] whatever (i, j) is -- lower entries and (i, i), which then counts twice, included.
cgh_matrix *cgh_matrix_from_entries_device(int64_t n, int64_t U, const double *d_diag, const double *d_upper,
                                           const int *d_rows, const int *d_cols, int layout, bool any_entries)
{
    CGH_REQUIRE_READY(nullptr);
    const bool inline_diag = d_diag == nullptr;
    if (layout == CGH_LAYOUT_CSRB) {
        cgh_matrix *csr = cgh_matrix_from_entries_device(n, U, d_diag, d_upper, d_rows, d_cols, CGH_LAYOUT_CSR, any_entries);
        return csr ? cgh_csrb_from_csr(csr) : nullptr;
    }
    if (layout == CGH_LAYOUT_PB) {
        cgh_matrix *csr = cgh_matrix_from_entries_device(n, U, d_diag, d_upper, d_rows, d_cols, CGH_LAYOUT_CSR, any_entries);
        return csr ? cgh_pb_from_csr(csr) : nullptr;
    }
    if (n <= 0 || U < 0 || (U > 0 && (!d_upper || !d_rows || !d_cols)) || layout < 0 || layout > CGH_LAYOUT_SELLS) {
        cgh_set_error("cgh_matrix_from_coo: bad arguments (n=%lld U=%lld layout=%d)", (long long)n, (long long)U, layout);
        return nullptr;
    }
    hipStream_t s = g_cgh.stream;
    cgh_buckets B;
    if (cgh_buckets_build(&B, n, U, d_rows, d_cols, true, any_entries ? 0 : (inline_diag ? 2 : 1), s) != CGH_OK) return nullptr;
    if (B.bad_seq != NO_STOP) {
        cgh_set_error(any_entries ? "cgh_matrix_from_coo: entry %llu has an index outside [0, n)"
                                  : "cgh_matrix_from_coo: entry %llu is not strictly upper-triangular / in range (the reference's "
                                    "compute_max_nnz_row_full returns -1 here)",
                      B.bad_seq >> 1);
        return nullptr;
    }
    cgh_matrix *m = new cgh_matrix();
    m->n = n;
    m->K = B.max_count;
    m->layout = layout;
    m->nnz_off = B.placements;
    m->diag = (double *)cgh_alloc(sizeof(double) * (size_t)n);
    bool ok = m->diag != nullptr;
    if (ok && inline_diag) ok = hipMemsetAsync(m->diag, 0, sizeof(double) * (size_t)n, s) == hipSuccess;
    if (ok && !inline_diag) ok = hipMemcpyAsync(m->diag, d_diag, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s) == hipSuccess;
    out_desc o;
    o.n = n;
    o.K = m->K;
    o.row_ptr = B.row_ptr;
    o.slice_ptr = nullptr;
    dev_buf row_pos(0);  // sigma-sorted SELL: position of each row (construction only)
    if (ok && layout == CGH_LAYOUT_ELL) {
        m->stored = n * (int64_t)m->K;
        o.mode = OUT_ELL_SLOT_MAJOR;
    } else if (ok && layout == CGH_LAYOUT_CSR) {
        m->stored = B.placements;
        o.mode = OUT_CSR;
        m->row_ptr = B.row_ptr;  // adopt the bucket offsets as the CSR row pointer
        B.row_ptr = nullptr;
        o.row_ptr = m->row_ptr;
    } else if (ok) {
        m->nslices = (n + 63) / 64;
        m->slice_ptr = (int64_t *)cgh_alloc(sizeof(int64_t) * (size_t)(m->nslices + 2));
        ok = m->slice_ptr != nullptr;
        if (ok && layout == CGH_LAYOUT_SELLS) {
            // rows sorted by descending length inside windows of sigma rows (stable: equal lengths stay in row order)
            m->sigma = g_cfg.sell_sigma < 64 ? 64 : g_cfg.sell_sigma;
            m->perm = (int *)cgh_alloc(sizeof(int) * (size_t)(m->nslices * 64));
            cgh_free(row_pos.p);
            row_pos.p = cgh_alloc(sizeof(int) * (size_t)n);
            dev_buf k0(sizeof(unsigned long long) * (size_t)n), k1(sizeof(unsigned long long) * (size_t)n), r0(sizeof(int) * (size_t)n),
                r1(sizeof(int) * (size_t)n);
            ok = m->perm && row_pos.p && k0.p && k1.p && r0.p && r1.p;
            if (ok) {
                hipLaunchKernelGGL(sigma_keys, dim3(grid_for(n)), dim3(256), 0, s, n, m->sigma, (const int *)B.cnt,
                                   k0.as<unsigned long long>(), r0.as<int>());
                size_t     tmp_bytes = 0;
                hipError_t he = rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0.as<unsigned long long>(), k1.as<unsigned long long>(),
                                                          r0.as<int>(), r1.as<int>(), (size_t)n, 0u, 64u, s);
                dev_buf    tmp(tmp_bytes ? tmp_bytes : 16);
                ok = he == hipSuccess && tmp.p;
                if (ok)
                    he = rocprim::radix_sort_pairs(tmp.p, tmp_bytes, k0.as<unsigned long long>(), k1.as<unsigned long long>(), r0.as<int>(),
                                                   r1.as<int>(), (size_t)n, 0u, 64u, s);
                ok = ok && he == hipSuccess;
                // slices (64 consecutive sorted rows) ordered by descending width over the whole matrix, so that the
                // SpMV's round-robin deal of slices to waves gives every wave the same mix of wide and narrow ones
                // (in window order the few very wide slices of all windows met on the same 32 waves: 23 ms per SpMV)
                const int64_t ns = m->nslices;
                dev_buf       ptmp(sizeof(int) * (size_t)(ns * 64)), sz(sizeof(int64_t) * (size_t)(ns + 1)),
                    sk0(sizeof(unsigned long long) * (size_t)ns), sk1(sizeof(unsigned long long) * (size_t)ns), so0(sizeof(int) * (size_t)ns),
                    so1(sizeof(int) * (size_t)ns);
                ok = ok && ptmp.p && sz.p && sk0.p && sk1.p && so0.p && so1.p;
                if (ok) {
                    hipLaunchKernelGGL(sigma_positions, dim3(grid_for(ns * 64)), dim3(256), 0, s, n, ns * 64, (const int *)r1.p,
                                       ptmp.as<int>(), (int *)nullptr);
                    hipLaunchKernelGGL(slice_sizes, dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, s, n, (const int *)B.cnt, ns,
                                       sz.as<int64_t>(), (const int *)ptmp.p);
                    hipLaunchKernelGGL(slice_keys, dim3(grid_for(ns)), dim3(256), 0, s, ns, (const int64_t *)sz.p,
                                       sk0.as<unsigned long long>(), so0.as<int>());
                    size_t stmp = 0;
                    he = rocprim::radix_sort_pairs(nullptr, stmp, sk0.as<unsigned long long>(), sk1.as<unsigned long long>(), so0.as<int>(),
                                                   so1.as<int>(), (size_t)ns, 0u, 64u, s);
                    dev_buf tmp2(stmp ? stmp : 16);
                    ok = he == hipSuccess && tmp2.p;
                    if (ok)
                        he = rocprim::radix_sort_pairs(tmp2.p, stmp, sk0.as<unsigned long long>(), sk1.as<unsigned long long>(), so0.as<int>(),
                                                       so1.as<int>(), (size_t)ns, 0u, 64u, s);
                    ok = ok && he == hipSuccess;
                    if (ok)
                        hipLaunchKernelGGL(permute_slices, dim3(grid_for(ns * 64)), dim3(256), 0, s, ns, (const int *)so1.p,
                                           (const int *)ptmp.p, m->perm, row_pos.as<int>());
                    ok = ok && hipStreamSynchronize(s) == hipSuccess;
                }
            }
            o.row_pos = row_pos.as<int>();
        }
        if (ok) {
            hipLaunchKernelGGL(slice_sizes, dim3((unsigned)((m->nslices + 3) / 4)), dim3(256), 0, s, n, (const int *)B.cnt, m->nslices,
                               m->slice_ptr, (const int *)m->perm);
            int64_t total = 0;
            ok = cgh_exclusive_scan_i64(m->slice_ptr, m->slice_ptr, m->nslices, &total, s) == CGH_OK;
            m->stored = total;
        }
        o.mode = OUT_SELL;
        o.slice_ptr = m->slice_ptr;
    }
    if (ok) {
        const int64_t st = m->stored ? m->stored : 1;
        m->vals = (double *)cgh_alloc(sizeof(double) * (size_t)st);
        m->cols = (int *)cgh_alloc(sizeof(int) * (size_t)st);
        ok = m->vals && m->cols;
    }
    if (ok) {
        if (layout != CGH_LAYOUT_CSR && m->stored > 0)
            hipLaunchKernelGGL(fill_pad<int>, dim3(grid_for(m->stored)), dim3(256), 0, s, m->stored, m->vals, m->cols);
        o.vals = m->vals;
        o.cols = m->cols;
        dev_buf ov(sizeof(unsigned long long));
        ok = ov.p != nullptr;
        if (ok && U > 0)
            hipLaunchKernelGGL((place_entries<true, int>), dim3(grid_for(2 * U)), dim3(256), 0, s, U, NO_STOP, d_rows, d_cols, d_upper,
                               (const int64_t *)o.row_ptr, (const unsigned *)B.seqs, o, ov.as<unsigned long long>(), B.diag_twice ? 1 : 0);
        ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    if (ok && layout == CGH_LAYOUT_CSR) ok = cgh_csr_build_row_blocks(m, s) == CGH_OK;
    if (!ok) {
        if (!g_cgh.err[0]) cgh_set_error("cgh_matrix_from_coo: device conversion failed");
        cgh_matrix_free(m);
        return nullptr;
    }
    return m;
}

extern "C" {

cgh_matrix *cgh_matrix_from_coo_host(int64_t n, int64_t U, const double *diag, const double *upper, const int *rows,
                                     const int *cols, int layout)
{
    CGH_REQUIRE_READY(nullptr);
    if (n <= 0 || U < 0 || !diag || (U > 0 && (!upper || !rows || !cols))) {
        cgh_set_error("cgh_matrix_from_coo_host: bad arguments");
        return nullptr;
    }
    const size_t u = (size_t)(U ? U : 1);
    dev_buf      dd(sizeof(double) * (size_t)n), dv(sizeof(double) * u), dr(sizeof(int) * u), dc(sizeof(int) * u);
    if (!dd.p || !dv.p || !dr.p || !dc.p) return nullptr;
    if (cgh_h2d(dd.p, diag, sizeof(double) * (size_t)n) || cgh_h2d(dv.p, upper, sizeof(double) * (size_t)U) ||
        cgh_h2d(dr.p, rows, sizeof(int) * (size_t)U) || cgh_h2d(dc.p, cols, sizeof(int) * (size_t)U))
        return nullptr;
    return cgh_matrix_from_coo_device(n, U, dd.as<double>(), dv.as<double>(), dr.as<int>(), dc.as<int>(), layout);
}

} // extern "C"
