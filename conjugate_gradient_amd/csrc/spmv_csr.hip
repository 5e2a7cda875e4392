// spmv_csr.hip -- y = A x for the full (both triangles) CSR and the sliced-ELL layouts.
//
// Replaces reference src/csr.c:190-229 (mv_csr_symmetric) and src/coo.c:202-233 (mv_coo_symmetric).
// The reference stores only the upper triangle and scatters (out[j] += a*v[i]), which is serial by
// construction.  Here the matrix is expanded once (convert.hip) to full rows whose entries sit in the
// order in which the reference's loop adds them into out[row] (diag, then contributions in entry
// order), so a gather-only kernel that sums each row left to right reproduces the reference bit for
// bit -- and needs no atomics.
//
// CSR "stream/adaptive" kernel (skewed row lengths, BASELINE config 5): rows are grouped into row blocks
// whose entries start inside one 1024-entry window (CGH_CSR_WINDOW; measured 256^3 7-point: 3.18 / 3.98 / 1.92 TB/s at
// 512 / 1024 / 2048).  A workgroup streams the block's values/columns
// with coalesced non-temporal loads, forms the rounded products v*x[col] into LDS, then
//   * rows of <= 64 entries: one lane adds the row's products left to right (bit-exact order);
//   * longer rows: a wave adds strided partial sums and combines them with a shuffle tree (NOT the
//     reference's order; differs in the last bits.  cgh_matrix_set_exact_rows() / CGH_CSR_EXACT=1 force the exact path for all rows,
//     which is what the reference-signature entry points of dropin.hip do);
//   * a row longer than the window is a block of its own: all 256 lanes stride over it.
// Sliced ELL (slice = 64 rows = one wave, slot-major inside the slice): exact order, coalesced.
#include "cgh_internal.h"
#include "device_utils.h"

int cgh_exclusive_scan_i32_to_i64(const int *d_in, int64_t *d_out, int64_t count, hipStream_t s);

namespace {

// entries per row-block window (a block's rows start in one window and the last may run one window over, so the LDS
// stage holds 2*window products): 1024 -> 16 KB of LDS per workgroup, 8 workgroups per CU
constexpr int CSR_WAVE_ROW = 64;             // rows longer than this are summed by a wave

int csr_window() { return g_cfg.csr_window; }

__global__ __launch_bounds__(256) void flag_row_blocks(int64_t n, const int64_t *__restrict__ row_ptr, int *__restrict__ flag, int CSR_WINDOW)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        int f = 0;
        if (r == 0) {
            f = 1;
        } else {
            const int64_t a = row_ptr[r - 1], b = row_ptr[r], c = row_ptr[r + 1];
            f = (a / CSR_WINDOW != b / CSR_WINDOW) || (b - a > CSR_WINDOW) || (c - b > CSR_WINDOW);
        }
        flag[r] = f;
    }
}

__global__ __launch_bounds__(256) void compact_row_blocks(int64_t n, const int *__restrict__ flag, const int64_t *__restrict__ pos,
                                                         int *__restrict__ blocks, int64_t nblocks)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= n; r += (int64_t)gridDim.x * 256) {
        if (r == n)
            blocks[nblocks] = (int)n;
        else if (flag[r])
            blocks[pos[r]] = (int)r;
    }
}

template <bool DOT, int CSR_WINDOW>
__global__ __launch_bounds__(CGH_BLOCK) void spmv_csr_stream(int64_t nblocks, const int *__restrict__ blocks,
                                                            const int64_t *__restrict__ row_ptr, const double *__restrict__ diag,
                                                            const double *__restrict__ vals, const int *__restrict__ cols,
                                                            const double *__restrict__ x, double *__restrict__ y,
                                                            double *__restrict__ partials, const int *__restrict__ done, int exact)
{
    constexpr int CSR_LDS = 2 * CSR_WINDOW;
    __shared__ double prod[CSR_LDS];
    __shared__ double red[4];
    if (done != nullptr && *done != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double    local = 0.0;
    for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int     r0 = blocks[b], r1 = blocks[b + 1];
        const int64_t e0 = row_ptr[r0], e1 = row_ptr[r1];
        if (r1 - r0 == 1 && e1 - e0 > CSR_LDS) {
            // one very long row: every lane strides over it
            double acc = 0.0;
            for (int64_t e = e0 + threadIdx.x; e < e1; e += CGH_BLOCK)
                acc += __builtin_nontemporal_load(vals + e) * x[__builtin_nontemporal_load(cols + e)];
            const double tot = block_sum_256(acc, red);
            if (threadIdx.x == 0) {
                const double xr = x[r0];
                const double yv = add_rn(0.0, mul_rn(diag[r0], xr)) + tot;
                y[r0] = yv;
                if (DOT) local += xr * yv;
            }
            continue;
        }
        const int cnt = (int)(e1 - e0);
        for (int q = threadIdx.x; q < cnt; q += CGH_BLOCK)
            prod[q] = mul_rn(__builtin_nontemporal_load(vals + e0 + q), x[__builtin_nontemporal_load(cols + e0 + q)]);
        __syncthreads();
        bool any_long = false;
        for (int r = r0 + threadIdx.x; r < r1; r += CGH_BLOCK) {
            const int q0 = (int)(row_ptr[r] - e0), q1 = (int)(row_ptr[r + 1] - e0);
            if (!exact && q1 - q0 > CSR_WAVE_ROW) {
                any_long = true;
                continue;
            }
            const double xr = x[r];
            double       acc = add_rn(0.0, mul_rn(diag[r], xr));
            for (int q = q0; q < q1; ++q) acc = add_rn(acc, prod[q]);
            __builtin_nontemporal_store(acc, y + r);
            if (DOT) local += xr * acc;
        }
        if (!exact && __syncthreads_or(any_long)) {
            for (int r = r0 + wave; r < r1; r += 4) {
                const int q0 = (int)(row_ptr[r] - e0), q1 = (int)(row_ptr[r + 1] - e0);
                if (q1 - q0 <= CSR_WAVE_ROW) continue;
                double acc = 0.0;
                for (int q = q0 + lane; q < q1; q += 64) acc += prod[q];
                acc = wave_sum(acc);
                if (lane == 0) {
                    const double xr = x[r];
                    const double yv = add_rn(0.0, mul_rn(diag[r], xr)) + acc;
                    y[r] = yv;
                    if (DOT) local += xr * yv;
                }
            }
        }
        __syncthreads();
    }
    if constexpr (DOT) {
        const double tot = block_sum_256(local, red);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

template <bool DOT>
__global__ __launch_bounds__(CGH_BLOCK) void spmv_sell64(int64_t n, int64_t nslices, const int64_t *__restrict__ slice_ptr,
                                                        const double *__restrict__ diag, const double *__restrict__ vals,
                                                        const int *__restrict__ cols, const double *__restrict__ x,
                                                        double *__restrict__ y, double *__restrict__ partials,
                                                        const int *__restrict__ done, const int *__restrict__ perm)
{
    __shared__ double red[4];
    if (done != nullptr && *done != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double    local = 0.0;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < nslices; s += (int64_t)gridDim.x * 4) {
        const int64_t b = slice_ptr[s];
        const int     w = (int)((slice_ptr[s + 1] - b) >> 6);
        // sigma-sorted slices: the lane's row comes from the permutation (-1 = no row)
        const int64_t r = perm != nullptr ? (int64_t)perm[s * 64 + lane] : s * 64 + lane;
        if (r < 0 || r >= n) continue;
        const double xr = x[r];
        double       acc = add_rn(0.0, mul_rn(diag[r], xr));
#pragma unroll 4
        for (int t = 0; t < w; ++t) {
            const int    c = __builtin_nontemporal_load(cols + b + (int64_t)t * 64 + lane);
            const double v = __builtin_nontemporal_load(vals + b + (int64_t)t * 64 + lane);
            const double g = x[c >= 0 ? (int64_t)c : r];
            const double tt = madd_rn(acc, v, g);
            acc = c >= 0 ? tt : acc;
        }
        __builtin_nontemporal_store(acc, y + r);
        if (DOT) local += xr * acc;
    }
    if constexpr (DOT) {
        const double tot = block_sum_256(local, red);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

} // namespace

// Row blocks for the stream kernel; stored in m->bin_rows (nblocks+1 row ids), count in m->bin_start[0].
int cgh_csr_build_row_blocks(cgh_matrix *m, hipStream_t s)
{
    const int64_t n = m->n;
    int          *flag = (int *)cgh_alloc(sizeof(int) * (size_t)(n + 1));
    int64_t      *pos = (int64_t *)cgh_alloc(sizeof(int64_t) * (size_t)(n + 2));
    int           rc = CGH_OK;
    if (!flag || !pos) rc = CGH_ENOMEM;
    int64_t nblocks = 0;
    if (rc == CGH_OK) {
        const int g = cgh_stream_grid((n + 255) / 256, 8);
        if (hipMemsetAsync(flag + n, 0, sizeof(int), s) != hipSuccess) rc = CGH_EHIP;
        hipLaunchKernelGGL(flag_row_blocks, dim3(g), dim3(256), 0, s, n, (const int64_t *)m->row_ptr, flag, csr_window());
        m->bin_start[1] = csr_window();
        if (rc == CGH_OK) rc = cgh_exclusive_scan_i32_to_i64(flag, pos, n + 1, s);
        if (rc == CGH_OK && (hipMemcpyAsync(&nblocks, pos + n, sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
                             hipStreamSynchronize(s) != hipSuccess))
            rc = CGH_EHIP;
        if (rc == CGH_OK) {
            m->bin_rows = (int *)cgh_alloc(sizeof(int) * (size_t)(nblocks + 1));
            if (!m->bin_rows) rc = CGH_ENOMEM;
        }
        if (rc == CGH_OK) {
            hipLaunchKernelGGL(compact_row_blocks, dim3(g), dim3(256), 0, s, n, (const int *)flag, (const int64_t *)pos, m->bin_rows,
                               nblocks);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = CGH_EHIP;
            m->bin_start[0] = nblocks;
        }
    }
    cgh_free(flag);
    cgh_free(pos);
    if (rc != CGH_OK) cgh_set_error("csr: building row blocks failed (%d)", rc);
    return rc;
}

template <int W>
static void launch_csr(const cgh_matrix *m, const double *x, double *y, double *dot_partials, const int *done, hipStream_t s, int grid,
                       int exact)
{
    const int64_t nb = m->bin_start[0];
    if (dot_partials)
        hipLaunchKernelGGL((spmv_csr_stream<true, W>), dim3(grid), dim3(CGH_BLOCK), 0, s, nb, (const int *)m->bin_rows,
                           (const int64_t *)m->row_ptr, (const double *)m->diag, (const double *)m->vals, (const int *)m->cols, x, y,
                           dot_partials, done, exact);
    else
        hipLaunchKernelGGL((spmv_csr_stream<false, W>), dim3(grid), dim3(CGH_BLOCK), 0, s, nb, (const int *)m->bin_rows,
                           (const int64_t *)m->row_ptr, (const double *)m->diag, (const double *)m->vals, (const int *)m->cols, x, y,
                           (double *)nullptr, done, exact);
}

int cgh_launch_spmv_csr(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                        const int *done, hipStream_t s)
{
    const int     exact = m->exact_rows || g_cfg.csr_exact;
    const int64_t nb = m->bin_start[0];
    const int     w = (int)m->bin_start[1];   // the window the row blocks were built for
    const int     grid = cgh_stream_grid(nb, w <= 1024 ? 8 : 5);
    if (grid_out) *grid_out = grid;
    if (w == 512) launch_csr<512>(m, x, y, dot_partials, done, s, grid, exact);
    else if (w == 2048) launch_csr<2048>(m, x, y, dot_partials, done, s, grid, exact);
    else launch_csr<1024>(m, x, y, dot_partials, done, s, grid, exact);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

int cgh_launch_spmv_sell(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                         const int *done, hipStream_t s)
{
    const int grid = cgh_stream_grid((m->nslices + 3) / 4, 8);
    if (grid_out) *grid_out = grid;
    if (dot_partials)
        hipLaunchKernelGGL(spmv_sell64<true>, dim3(grid), dim3(CGH_BLOCK), 0, s, m->n, m->nslices, (const int64_t *)m->slice_ptr,
                           (const double *)m->diag, (const double *)m->vals, (const int *)m->cols, x, y, dot_partials, done,
                           (const int *)m->perm);
    else
        hipLaunchKernelGGL(spmv_sell64<false>, dim3(grid), dim3(CGH_BLOCK), 0, s, m->n, m->nslices, (const int64_t *)m->slice_ptr,
                           (const double *)m->diag, (const double *)m->vals, (const int *)m->cols, x, y, (double *)nullptr, done,
                           (const int *)m->perm);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}
